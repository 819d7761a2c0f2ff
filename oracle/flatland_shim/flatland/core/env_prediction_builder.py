"""Part of the test-only flatland stand-in; see flatland/_impl.py."""
from flatland._impl import PredictionBuilder  # noqa: F401
