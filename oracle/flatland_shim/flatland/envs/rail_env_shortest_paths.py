"""Part of the test-only flatland stand-in; see flatland/_impl.py."""
from flatland._impl import get_valid_move_actions_, get_new_position  # noqa: F401
