"""CPU oracle for the blub fluid step.  TEST INFRASTRUCTURE ONLY (see oracle/blub_oracle.cpp header)."""
from .oracle import Oracle, build_oracle  # noqa: F401
