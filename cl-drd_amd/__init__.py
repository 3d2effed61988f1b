"""cldrd_amd: MI355X-native (gfx950) implementation of the CL-DRD hot path.

N-way dual-encoder distillation training and the index / retrieve path of HansiZeng/CL-DRD, behind the
reference's own call surface (``models.nway_dual_encoder.NwayDualEncoder``, ``losses.*``,
``retriever.retrieval_utils.*``).  All arithmetic runs in hand-written HIP kernels reached through the
C-ABI library ``libcldrd_hip.so`` (``include/cldrd_hip.h``); there is no CPU fallback: calling an op
without the library or without a GPU raises.

Import as ``import cldrd_amd`` (the directory is ``cl-drd_amd/``; ``cldrd_amd.py`` at the repo root is
the import shim).
"""
__version__ = "0.1.0"
