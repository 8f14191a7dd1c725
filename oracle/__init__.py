"""ORACLE — CPU restatement of the reference's RA / GP / BA algorithms.

Test infrastructure only.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this package; the product path (glomap_amd/) must never do so.

parity unpinned: the reference (colmap/glomap v1.1.0) stores no golden numeric vectors for this
path and cannot be built here (Eigen/Ceres/COLMAP/CHOLMOD absent) — see DESIGN.md §Oracle.
"""
