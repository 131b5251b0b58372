"""tests-only stand-in package (see tests/stubs/README.md)."""
