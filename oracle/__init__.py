"""ORACLE -- test infrastructure only (CPU restatement of the reference's hot path).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
