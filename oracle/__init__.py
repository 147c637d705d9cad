"""CPU oracle for the spectral-mix hot path — TEST INFRASTRUCTURE ONLY (see spectral_mix_oracle.py)."""
