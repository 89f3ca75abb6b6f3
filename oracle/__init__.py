"""ORACLE package — test infrastructure only: CPU restatements of the reference's hot-path algorithm.
Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the product."""
