"""CPU oracle of the LBM hot path -- test infrastructure only (see lettuce_oracle.py)."""
