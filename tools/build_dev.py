#!/usr/bin/env python3
"""Development build of the kernel library: cl-drd_amd/libcldrd_hip_dev.so = the product sources + -DCLDRD_DEV_BUILD.

Only this build reads the experiments' environment knobs (CLDRD_DEV_INT sites in csrc/: tile / split / schedule overrides) and contains
the timing-only ablation kernels (CLDRD_GEMM_ABLATE, CLDRD_SCAN_ABLATE: WRONG results by design).  The product library
(libcldrd_hip.so, cl-drd_amd/build.py) reads no environment variable and has none of them compiled in (tests/test_capi.py).
Select it per process with CLDRD_LIB=cl-drd_amd/libcldrd_hip_dev.so (tools/*.py, tools/*.sh).  Extra -D flags: pass them as arguments
(python tools/build_dev.py -DCLDRD_TAPE_NT=0)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cl-drd_amd", "csrc")
OUT = os.path.join(ROOT, "cl-drd_amd", "libcldrd_hip_dev.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-fno-slp-vectorize", "-DCLDRD_DEV_BUILD"]


def main(extra):
    objdir = os.path.join(CSRC, "build", "dev")
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(objdir, s[:-4] + ".o")
        objs.append(o)
        sp = os.path.join(CSRC, s)
        if extra or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(sp), hdr_t):
            jobs.append([HIPCC, *FLAGS, *extra, "-c", sp, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(f"hipcc failed: {' '.join(cmd)}\n{r.stderr}")
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs])
    print(OUT)


if __name__ == "__main__":
    main([a for a in sys.argv[1:] if a.startswith("-D")])
