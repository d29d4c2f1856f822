#!/usr/bin/env python3
"""Round 6, host-boundary share of an ensemble member at config 3: `ensemble_of_topics(X_host, 64, n_runs=4, n_iter=50)` with one
member at a time (the default above 2e9 cells) and with two members in flight on two contexts (the second member's upload-free
setup -- bootstrap gather, CSC / item build, MT19937 initialisation -- then overlaps the first member's iterations)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1:
    import numpy as np
    import bench
    import enstop_amd
    from enstop_amd.engine import Engine
    cfg = bench.CONFIGS[3]
    with Engine(0) as eng:
        eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], seed=0)
        X = eng.download_active_csr()
    kw = dict(n_runs=4, n_iter=50, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, random_state=7, n_jobs=int(sys.argv[1]))
    enstop_amd.ensemble_of_topics(X, cfg["k"], **dict(kw, n_runs=2))         # warm-up: contexts, buffers, page-locked slots
    walls = []
    for _ in range(3):
        t0 = time.perf_counter()
        T = enstop_amd.ensemble_of_topics(X, cfg["k"], **kw)
        walls.append(time.perf_counter() - t0)
    w = sorted(walls)[1]
    print(json.dumps({"config": 3, "members": 4, "members_in_flight": int(sys.argv[1]), "wall_s": round(w, 4),
                      "walls_s": [round(x, 4) for x in walls], "ms_per_member": round(w / 4 * 1e3, 1),
                      "fits_per_min": round(4 / w * 60, 1), "checksum": float(T.astype(np.float64).sum())}), flush=True)
else:
    for jobs, cells in ((1, "2e9"), (2, "1e12")):
        env = dict(os.environ, ENSTOP_AMD_CONCURRENT_MEMBERS_CELLS=cells)
        subprocess.run([sys.executable, os.path.abspath(__file__), str(jobs)], env=env, check=False)
