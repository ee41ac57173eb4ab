"""Alias module for the reference's `src/ce_dl_cnn.py` (same three public names, same signature, C:802): this build's
estimator with the fixed-weight in-painting interpolation (`interp="cnn"`), `config.CNNSmoothingAlpha` honoured as the
reference reads it (C:864).  See INTEGRATION.md."""
import functools

from srsran_ce_pytorch_amd.config import EstimatorConfig, HopConfig  # noqa: F401
from srsran_ce_pytorch_amd.estimator import srs_channel_estimator as _estimator

srs_channel_estimator = functools.partial(_estimator, interp="cnn")
