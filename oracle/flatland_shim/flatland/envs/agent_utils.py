"""Part of the test-only flatland stand-in; see flatland/_impl.py."""
from flatland._impl import RailAgentStatus, EnvAgent  # noqa: F401
