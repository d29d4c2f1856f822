import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from enstop_amd.engine import Engine
eng = Engine()
eng.generate_synthetic(18846, 173762, 2950000, seed=0)
for r in range(21):
    eng.init_factors_numpy_stream(20, np.random.RandomState(42 + r))
eng.synchronize()
t = time.perf_counter()
for r in range(20):
    eng.init_factors_numpy_stream(20, np.random.RandomState(42 + r))
print("avg init ms", (time.perf_counter() - t) / 20 * 1e3)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for r in range(20):
    eng.init_factors_numpy_stream(20, np.random.RandomState(42 + r))
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
