#!/usr/bin/env python3
"""Per-kernel HIP-event times of one EM iteration in the REFERENCE arithmetic (PLSA_REFERENCE_SUMS) at BASELINE config 1, config 2
and the first 150 000 documents of config 3: which of the chains binds the parity mode."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                          # noqa: E402
from enstop_amd.engine import Engine, PLSA_REFERENCE_SUMS, PLSA_REFERENCE_LL   # noqa: E402

# (--whole: the whole of config 3 instead -- 100 M non-zeros, 26 GB of P(z|w,d))
for cfg_id, rows in (((3, 0),) if "--whole" in sys.argv else ((1, 0), (2, 0), (3, 150_000))):
    cfg = bench.CONFIGS[cfg_id]
    with Engine(0) as eng:
        eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], seed=0)
        if rows:
            eng.bootstrap(np.arange(rows, dtype=np.int64))
        n, m, nnz = eng.shape
        eng.init_factors_numpy_stream(cfg["k"], np.random.RandomState(42))
        kw = dict(n_iter_per_test=2, tolerance=0.0, e_step_thresh=1e-32, flags=PLSA_REFERENCE_SUMS | PLSA_REFERENCE_LL)
        eng.fit(None, n_iter=2, **kw)
        wall = {}
        for name, fl in (("reference", PLSA_REFERENCE_SUMS), ("reference_source", PLSA_REFERENCE_SUMS | PLSA_REFERENCE_LL)):
            kw_wall = dict(kw, flags=fl, n_iter_per_test=10)          # the reference's default: one likelihood per ten iterations
            reps = 4 if "--whole" in sys.argv else 20
            eng.fit(None, n_iter=reps // 2, **kw_wall)
            t0 = time.perf_counter()
            eng.fit(None, n_iter=reps, **kw_wall)
            wall[name] = round((time.perf_counter() - t0) / reps * 1e3, 2)
        eng.timing(True)
        eng.timing_reset()
        eng.fit(None, n_iter=4, **kw)
        rep = eng.timing_report()
        eng.timing(False)
        print(json.dumps({"config": cfg_id, "ms_per_iteration": wall, "norm_chain": eng.reference_chain_info(), "rows": n, "nnz": nnz, "k": cfg["k"],
                          "avg_ms": {k: round(v[1] / v[0], 3) for k, v in sorted(rep.items())},
                          "launches": {k: v[0] for k, v in sorted(rep.items())}}), flush=True)
