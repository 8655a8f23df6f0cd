"""ORACLE package -- test infrastructure only.

CPU restatements of the reference's hot path (each function cites the
reference file:line it follows).  Nothing under activezero_amd/ may import from
here; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
"""
