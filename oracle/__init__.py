"""CPU oracle for the SAD stereo block matcher -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The shipped path (opencv_amd/) never does.  See oracle/sad_oracle.c for the parity statement
("parity unpinned by the reference": the reference has no tests for this path).
"""
from .oracle import (  # noqa: F401
    RAW_LOC, build, lib, literal, literal_rows, literal_mt, band, box, rds, splitmix64, numpy_restatement,
)
