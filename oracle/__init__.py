"""CPU oracle (TEST INFRASTRUCTURE ONLY): ctypes loader for oracle/liborb_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from .orb_oracle import *  # noqa: F401,F403
