"""Alias module for the name the reference's single-case scripts import (`validate_case0.py:12`,
`validate_case4.py:15`, `validate_case8.py:12`, `diagnose_furiosa_backend.py:21`); upstream never shipped it."""
from srsran_ce_pytorch_amd.config import EstimatorConfig, HopConfig  # noqa: F401
from srsran_ce_pytorch_amd.estimator import srs_channel_estimator  # noqa: F401
