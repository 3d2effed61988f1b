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
# RCCL communicator exists in the process (its own streams) the two collide on one hardware queue and an EAGER step loses the overlap of the
# towers: 13.2 instead of 12.6 ms per cfg2 step with ProcessGroupNCCL initialised on one rank; eight queues restore it.  A REPLAYED HIP graph
# is the other way round: 12.3 ms with the default four queues, 15.1 ms with eight (its branches spread over more queues and pay for the
# cross-queue dependencies).  Data-parallel ranks can replay the step as a graph too, RCCL collectives captured inside
# (profiles/r04_microbench.txt) - the default for a ONE-rank process group, opt-in (CLDRD_DDP_GRAPH=1) for real
# multi-rank jobs (trainer/nway_listwise.py: _graph_wanted).  So eight queues for ranks that will NOT replay a graph: CLDRD_GRAPH=0, or a
# multi-rank job without CLDRD_DDP_GRAPH=1, or CLDRD_DDP_GRAPH=0.  Read by the HIP runtime when it initialises: this works only before the first GPU call of the process - importing
# the package first is enough; an explicit setting wins.
_world = int(_os.environ.get("WORLD_SIZE", "1") or 1)
if (_world > 1 or _os.environ.get("CLDRD_FORCE_DDP", "0") == "1") and \
        (_os.environ.get("CLDRD_GRAPH", "1") == "0" or _os.environ.get("CLDRD_DDP_GRAPH", "1" if _world == 1 else "0") != "1"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
