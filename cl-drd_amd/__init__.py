"""cldrd_amd: MI355X-native (gfx950) implementation of the CL-DRD hot path.

N-way dual-encoder distillation training and the index / retrieve path of HansiZeng/CL-DRD, behind the
reference's own call surface (``models.nway_dual_encoder.NwayDualEncoder``, ``losses.*``,
``retriever.retrieval_utils.*``).  All arithmetic runs in hand-written HIP kernels reached through the
C-ABI library ``libcldrd_hip.so`` (``include/cldrd_hip.h``); there is no CPU fallback: calling an op
without the library or without a GPU raises.

Import as ``import cldrd_amd`` (the directory is ``cl-drd_amd/``; ``cldrd_amd.py`` at the repo root is
the import shim).
"""
import os as _os

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The trainer runs two towers on two streams; once an
# RCCL communicator exists in the process (its own streams) the two collide on one hardware queue and the query tower no longer runs NEXT
# to the passage tower: 13.4 instead of 12.1 ms per cfg2 step, measured with ProcessGroupNCCL initialised on one rank
# (profiles/r03_microbench.txt).  Eight queues restore it.  Only for ranks of a multi-process job, though: the replayed HIP graph of the
# single-process step is SLOWER with eight queues (14.7 against 11.9 ms: its branches spread over more queues and pay for the cross-queue
# dependencies), and data-parallel ranks never replay a graph.  Read by the HIP runtime when it initialises, so this works only before the
# first GPU call of the process - importing the package first is enough; an explicit setting wins.
if int(_os.environ.get("WORLD_SIZE", "1") or 1) > 1 or _os.environ.get("CLDRD_FORCE_DDP", "0") == "1":
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
