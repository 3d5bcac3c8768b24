"""Symbol-only stand-in (see tests/refshim/README.md)."""
