"""Part of the test-only flatland stand-in; see flatland/_impl.py."""
