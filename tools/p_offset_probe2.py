#!/usr/bin/env python3
"""Same physical allocation of P (over-allocated by 3 GB), different start offsets: is the E-step's
speed a function of the address?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enstop_amd.engine import Engine
from bench import init_factors
eng = Engine(0)
n, m, k = 1_000_000, 100_000, 64
eng.generate_synthetic(n, m, 100_000_000, seed=0)
U0, V0 = init_factors(n, m, k, 42)
eng.set_factors(U0, V0)
eng.timing(True)
os.environ["PLSA_P_SLACK_MB"] = "3072"
def measure(tag):
    eng.e_step(1e-32, want_host_copy=False)
    eng.timing_reset()
    for _ in range(4):
        eng.e_step(1e-32, want_host_copy=False)
    ms, cnt = eng.timing_get("k_e_step")
    print("%-26s %.3f ms  frac %.3f" % (tag, ms / cnt, 26.3989e9 / (ms / cnt / 1e3) / 8e12), flush=True)
for alloc in range(2):
    eng.release_scratch()
    print("-- allocation", alloc)
    for off_kb in (0, 1, 4, 16, 64, 256, 1024, 2048, 4096, 8192, 16384, 65536, 262144, 1048576, 2097152, 0):
        os.environ["PLSA_P_OFFSET_KB"] = str(off_kb)
        measure("offset %d KB" % off_kb)
