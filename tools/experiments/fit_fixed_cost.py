"""Fixed cost of one plsa_fit call (config 2 and the 20NG shape): wall time of fits of 0, 1, 11, 50, 200 iterations."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import bench
from enstop_amd.engine import Engine, PLSA_FUSED
for cfg_id in (2, 1):
    cfg = bench.CONFIGS[cfg_id]
    eng = Engine(0)
    eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], seed=0)
    U0, V0 = bench.init_factors(cfg["n"], cfg["m"], cfg["k"], 42)
    eng.set_factors(U0, V0)
    eng.fit(None, n_iter=20, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
    for n_iter in (0, 1, 2, 11, 50, 200, 50, 11, 1):
        best = 1e9
        for rep in range(5):
            eng.synchronize(); t = time.perf_counter()
            eng.fit(None, n_iter=n_iter, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
            eng.synchronize(); best = min(best, time.perf_counter() - t)
        print(json.dumps({"config": cfg_id, "n_iter": n_iter, "ms": round(best * 1e3, 4)}), flush=True)
    eng.close()
