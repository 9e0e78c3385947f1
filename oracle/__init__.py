"""CPU restatement of the reference's hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (usdm_amd/) never does: it has no CPU fallback.
Pinning status of each oracle is stated in its module header and in DESIGN.md.
"""
