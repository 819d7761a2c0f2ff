"""Part of the test-only flatland stand-in; see flatland/_impl.py."""
from flatland._impl import get_new_position, direction_to_point  # noqa: F401
