"""CPU oracle of the SpMM hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; nothing under matrix-multiplication_amd/ does.  See
spmm_oracle.c for the algorithms (each cites the reference file:line it
restates) and for how the oracle is pinned.
"""
from .oracle import *  # noqa: F401,F403
