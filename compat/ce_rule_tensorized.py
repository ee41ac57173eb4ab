"""Alias module: lets the reference's harness (`scripts/validation/validate_all.py:18`) import this build's
estimator under the reference's module name.  See INTEGRATION.md."""
from srsran_ce_pytorch_amd.config import EstimatorConfig, HopConfig  # noqa: F401
from srsran_ce_pytorch_amd.estimator import srs_channel_estimator  # noqa: F401
