"""MI355X-native PUSCH DM-RS channel estimator (batched, HIP kernels behind a C ABI)."""
from .config import EstimatorConfig, HopConfig, empty_hop  # noqa: F401
