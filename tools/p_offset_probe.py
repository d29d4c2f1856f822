#!/usr/bin/env python3
"""Does the E-step's speed depend on where the 25.7 GB P array lands?  Re-allocates P after
perturbing allocations inside ONE process and times k_e_step each time; PLSA_CONTIG=1 asks for
physically contiguous HBM."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enstop_amd.engine import Engine
from bench import init_factors

eng = Engine(0)
n, m, k = 1_000_000, 100_000, 64
nnz = eng.generate_synthetic(n, m, 100_000_000, seed=0)
U0, V0 = init_factors(n, m, k, 42)
eng.set_factors(U0, V0)
eng.timing(True)

def measure(tag):
    eng.e_step(1e-32, want_host_copy=False)          # first touch
    eng.timing_reset()
    for _ in range(5):
        eng.e_step(1e-32, want_host_copy=False)
    ms, cnt = eng.timing_get("k_e_step")
    print("%-34s %.3f ms  frac %.3f" % (tag, ms / cnt, 26.3989e9 / (ms / cnt / 1e3) / 8e12), flush=True)

for rep in range(3):
    for off_kb in (0, 4, 2048):
        os.environ["PLSA_P_OFFSET_KB"] = str(off_kb)
        eng.release_scratch()
        measure("contig=%s offset %d KB" % (os.environ.get("PLSA_CONTIG", "0"), off_kb))
    eng.release_scratch()
    eng.stream_bandwidth((7 + 13 * rep) << 30, 0, 1)  # perturb the allocator
    os.environ["PLSA_P_OFFSET_KB"] = "0"
    measure("contig=%s after %d GB alloc/free" % (os.environ.get("PLSA_CONTIG", "0"), 7 + 13 * rep))
