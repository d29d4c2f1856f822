"""Per-kernel time of one NumPy-identical initialisation at the 20NG shape (HIP events around every launch family)."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from enstop_amd.engine import Engine
with Engine() as eng:
    eng.generate_synthetic(18846, 173762, 2950000, seed=0)
    for _ in range(3): eng.init_factors_numpy_stream(20, np.random.RandomState(42))
    best = 1e9
    for rep in range(10):
        rs = np.random.RandomState(42)
        eng.synchronize(); t = time.perf_counter(); eng.init_factors_numpy_stream(20, rs); best = min(best, time.perf_counter() - t)
    print("init wall ms", round(best * 1e3, 3))
