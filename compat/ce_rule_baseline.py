"""Alias module for the reference's `src/ce_rule_baseline.py` (loop-style twin of ce_rule_tensorized: same signature,
outputs within 4e-8 of it -- tests/golden/MANIFEST.json records the agreement per fixture).  See INTEGRATION.md."""
from srsran_ce_pytorch_amd.config import EstimatorConfig, HopConfig  # noqa: F401
from srsran_ce_pytorch_amd.estimator import srs_channel_estimator  # noqa: F401
