"""MI355X-native observation-and-rollout hot path of flatland-marl (gfx950 HIP kernels behind a C ABI)."""
from ._lib import FlmError, load  # noqa: F401
from .batch import ObservationBatch, times_per_cell  # noqa: F401
