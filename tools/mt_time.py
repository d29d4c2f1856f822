"""Time the device-side NumPy-stream factor initialisation (plsa_init_factors_mt19937) at config-3
scale for different numbers of jump-ahead streams.  Run on the GPU box: python tools/mt_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from enstop_amd.engine import Engine

for streams in (1, 32, 128, 256, 512):
    os.environ["PLSA_MT_STREAMS"] = str(streams)
    with Engine() as eng:
        eng.generate_synthetic(1000000, 100000, 100000000, seed=0)
        rs = np.random.RandomState(42)
        eng.init_factors_numpy_stream(64, rs)          # warm-up: builds the jump polynomials once
        rs = np.random.RandomState(42)
        eng.timing(True); eng.timing_reset()
        t = time.time(); eng.init_factors_numpy_stream(64, rs); dt = time.time() - t
        U, V = eng.get_factors()
        print(streams, "init %.1f ms" % (dt * 1e3), "fill", eng.timing_get("k_mt19937_fill"), "jump", eng.timing_get("k_mt_jump"),
              float(U[12345, 7]), float(V[3, 4567]), rs.rand(), flush=True)
