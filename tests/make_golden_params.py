"""Seeds / shapes shared by tests/golden/make_golden.py (which needs /root/reference) and the tests (which must not)."""
GEN_B4 = dict(size=1024, batch=4, z_seed=21, noise_seed=2100)
