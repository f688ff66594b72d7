"""CPU oracle — test infrastructure only (see sdrm_oracle.py header)."""
