"""CPU oracle for the hot path — test infrastructure only (see ref_cpu.py header)."""
