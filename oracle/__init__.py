"""CPU oracle package (test infrastructure only; never imported by needle_amd)."""
