"""CPU oracle for the Eagle hot path -- TEST INFRASTRUCTURE ONLY (parity unpinned, see eagle_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
