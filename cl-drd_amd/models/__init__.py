from .nway_dual_encoder import NwayDualEncoder  # noqa: F401
