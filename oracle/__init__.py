"""CPU parity oracle — TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (pygraphblas_amd) never does."""
from .oracle import *  # noqa: F401,F403
