"""CPU oracle for the TriForce hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package,
and only as the checker / reported CPU baseline.  triforce_amd/ never imports it and has no CPU
fallback: without the HIP library its ops raise.

Parity status: the reference has no golden vectors for this path.  The oracle is
pinned against the reference's own Python, run in the build container by oracle/gen_golden.py;
the resulting fixtures live in tests/golden/ and are re-checked on every test run.
"""
