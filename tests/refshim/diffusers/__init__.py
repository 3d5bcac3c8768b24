"""Symbol-only stand-in for diffusers==0.16.0 (see tests/refshim/README.md)."""
