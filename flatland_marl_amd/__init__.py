"""MI355X-native observation-and-rollout hot path of flatland-marl (gfx950 HIP kernels behind a C ABI)."""
from ._lib import FlmError, load  # noqa: F401
from .batch import ObservationBatch, times_per_cell  # noqa: F401
from .predictions import ShortestPathPredictorForRailEnv  # noqa: F401,E402
from .rail_observations import RailObsForRailEnv  # noqa: F401,E402
from .preprocessing import ObsPreprocessor  # noqa: F401,E402
