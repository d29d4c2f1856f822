"""MT19937 factor initialisation (plsa_init_factors_mt19937) at the 20NG shape (config 1 / 4: 7.7 M words) for
different numbers of jump-ahead streams: total time and the two kernel families.  python tools/mt_time_small.py"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from enstop_amd.engine import Engine

for streams in (1, 8, 16, 32, 64, 128, 256, 512, 1024):
    os.environ["PLSA_MT_STREAMS"] = str(streams)
    os.environ["PLSA_MT_MIN_BLOCKS"] = "1"
    with Engine() as eng:
        eng.generate_synthetic(18846, 173762, 2950000, seed=0)
        eng.init_factors_numpy_stream(20, np.random.RandomState(42))     # warm-up: jump polynomials, buffers
        best = 1e9
        for rep in range(5):
            rs = np.random.RandomState(42)
            eng.synchronize(); t = time.perf_counter(); eng.init_factors_numpy_stream(20, rs); best = min(best, time.perf_counter() - t)
        eng.timing(True); eng.timing_reset()
        eng.init_factors_numpy_stream(20, np.random.RandomState(42))
        rep_ = {k: round(v[1] / v[0], 4) for k, v in eng.timing_report().items()}
        U, V = eng.get_factors()
        print(json.dumps({"mt_streams": streams, "init_ms": round(best * 1e3, 3), "kernels_ms": rep_, "check": [float(U[123, 7]), float(V[3, 4567]), rs.rand()]}), flush=True)
