"""CPU oracles (test infrastructure only; never imported by the product path)."""
