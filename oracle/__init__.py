"""CPU oracle package -- test infrastructure only (see oracle/lq_oracle.py header)."""
