"""Minimal stand-in for ``torchrl`` (tests only; see tests/stubs/tensordict)."""
