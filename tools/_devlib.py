"""Experiments that flip the library's tuning / ablation knobs need the DEVELOPMENT build (tools/build_dev.py): the product library reads
no environment variable.  `import _devlib` before importing cldrd_amd selects it (and builds it when missing)."""
import os
import subprocess
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV_LIB = os.path.join(_ROOT, "cl-drd_amd", "libcldrd_hip_dev.so")
if "CLDRD_LIB" not in os.environ:
    if not os.path.exists(DEV_LIB):
        subprocess.check_call([sys.executable, os.path.join(_ROOT, "tools", "build_dev.py")])
    os.environ["CLDRD_LIB"] = DEV_LIB
