"""Part of the test-only flatland stand-in; see flatland/_impl.py."""
from flatland._impl import DistanceMap  # noqa: F401
