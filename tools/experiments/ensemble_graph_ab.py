#!/usr/bin/env python3
"""Does hipGraph replay of the EM iterations (PLSA_GRAPH=1; neutral for a single fit) pay when FOUR members are in flight
on one GPU and their threads compete for the runtime's launch path?  20NG-shaped corpus, 32 members x 50 iterations through
enstop_amd.ensemble_of_topics, n_jobs in {1, 2, 4}.   for g in 0 1; do PLSA_GRAPH=$g python tools/experiments/ensemble_graph_ab.py; done"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import enstop_amd  # noqa: E402
from enstop_amd.engine import get_engine  # noqa: E402

cfg = bench.CONFIGS[4]
eng = get_engine(0)
eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], seed=0)
X = eng.download_active_csr()
for jobs in (1, 2, 4):
    kw = dict(n_iter=50, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, random_state=7, n_jobs=jobs)
    enstop_amd.ensemble_of_topics(X, cfg["k"], n_runs=jobs, **kw)
    walls = []
    for _ in range(5):
        t0 = time.perf_counter()
        S = enstop_amd.ensemble_of_topics(X, cfg["k"], n_runs=32, **kw)
        walls.append(time.perf_counter() - t0)
    w = sorted(walls)[2]
    print(json.dumps({"PLSA_GRAPH": os.environ.get("PLSA_GRAPH", "0"), "n_jobs": jobs, "fits_per_min": round(32 / w * 60, 1),
                      "ms_per_fit": round(w / 32 * 1e3, 3), "checksum": float(S.astype(np.float64).sum())}), flush=True)
