"""``losses`` package of the reference (``losses/__init__.py:1-7``) on MI355X: same names and call signatures;
forward value and analytic gradient come from one fused HIP kernel per call (csrc/loss.hip)."""
from .lambda_rank import lambda_mrr_loss  # noqa: F401
from .lambda_rank import bweight_lambda_mrr_loss  # noqa: F401
from .ranknet import ranknet_loss  # noqa: F401
from .weighted_pointwise import weighted_pointwise_loss  # noqa: F401
from .standard_lambda_rank import lambda_loss  # noqa: F401
from .margin_mse import MarginMSE  # noqa: F401
from .kl_div import KLDiv  # noqa: F401
