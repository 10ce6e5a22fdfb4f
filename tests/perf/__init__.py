"""Measurement scripts that time the CPU oracle beside the HIP path (kept under tests/: only tests/,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import `oracle/`). Not collected by pytest."""
