"""Minimal stand-in for the `fvcore` package, written for this repo.

TEST INFRASTRUCTURE ONLY.  It exists so that `tests/golden/make_golden.py` can import the
reference (`/root/reference/vidgen`, which hard-depends on fvcore/termcolor that are not installed
in this image) and capture golden input/output vectors.  Nothing in `lvt_amd/` imports it.
"""
