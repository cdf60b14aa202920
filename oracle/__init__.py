"""CPU oracle for the BP+OSD decode path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  See oracle/bposd_oracle.h for the parity status ("parity unpinned").
"""
from .oracle import OracleDecoder, build_oracle, portable_math, portable_tanh_half, portable_log_quot  # noqa: F401
