"""Member initialisation at the 20NG shape and at config 3's shape: chunked marginals against the plain chain (PLSA_MT_CHAIN=1)."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from enstop_amd.engine import Engine
for shape in ((18846, 173762, 2950000, 20), (100000, 50000, 10000000, 32), (1000000, 100000, 100000000, 64)):
    n, m, nnz, k = shape
    for chain in ("0", "1"):
        os.environ["PLSA_MT_CHAIN"] = chain
        with Engine() as eng:
            eng.generate_synthetic(n, m, nnz, seed=0)
            eng.init_factors_numpy_stream(k, np.random.RandomState(42))
            best = 1e9
            for rep in range(5):
                rs = np.random.RandomState(42)
                eng.synchronize(); t = time.perf_counter(); eng.init_factors_numpy_stream(k, rs); best = min(best, time.perf_counter() - t)
            marg = eng.mt_marginals()
            print(json.dumps({"shape": shape, "chain": chain, "init_ms": round(best * 1e3, 3), "marg0_bits": int(marg[:1].view(np.int64)[0])}), flush=True)
