"""Kernel-level cost of rebuilding the per-resample structure (CSC copy, column items, visiting order, row order / items) at
the 20NG shape: 40 x (bootstrap + a 1-iteration fit) under rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from enstop_amd.engine import Engine, PLSA_FUSED
N, M, NNZ, K = 18_846, 173_762, 2_950_000, 20
eng = Engine(0)
eng.generate_synthetic(N, M, NNZ, seed=0)
eng.init_factors_device(K, 1)
for r in range(3):
    eng.bootstrap(np.random.RandomState(r).randint(0, N, size=N)); eng.init_factors_device(K, r)
    eng.fit(None, n_iter=1, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
eng.synchronize(); t = time.perf_counter()
for r in range(40):
    eng.bootstrap(np.random.RandomState(10 + r).randint(0, N, size=N)); eng.init_factors_device(K, r)
    eng.fit(None, n_iter=1, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
eng.synchronize(); print("ms per (bootstrap + device init + 1-iteration fit):", (time.perf_counter() - t) / 40 * 1e3)
