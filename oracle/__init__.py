"""CPU oracle package (test infrastructure only -- see bls12_381.py / msm_oracle.c headers)."""
