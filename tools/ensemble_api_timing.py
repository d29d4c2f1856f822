#!/usr/bin/env python3
"""Throughput of the reference-shaped API call itself -- enstop_amd.ensemble_of_topics(X, k, n_runs=...)
on a host scipy matrix (upload, per-member bootstrap draw + device gather, NumPy-identical device
initialisation, fit, topic download, vstack) -- on one GPU.  Prints one JSON object per configuration."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import enstop_amd                                              # noqa: E402
from enstop_amd.engine import Engine                           # noqa: E402

CONFIGS = {"cfg4(20NG-shaped,k=20,n_starts=32)": (18_846, 173_762, 2_950_000, 20, 32),
           "cfg3-shape(1Mx100k,k=64,n_starts=4)": (1_000_000, 100_000, 100_000_000, 64, 4)}


def main():
    for name, (n, m, nnz_t, k, runs) in CONFIGS.items():
        with Engine(0) as eng:
            eng.generate_synthetic(n, m, nnz_t, seed=0)
            X = eng.download_active_csr()
        enstop_amd.ensemble_of_topics(X, k, n_runs=1, parallelism="none", n_iter=2, tolerance=0.0, random_state=1)   # warm-up
        enstop_amd.ensemble_of_topics(X, k, n_runs=4, n_jobs=4, n_iter=2, tolerance=0.0, random_state=1)             # extra contexts
        plan = (("none", 1), ("dask", 1), ("dask", 2), ("dask", 4))
        if os.environ.get("JOBS"):                       # e.g. JOBS=4,6,8: thread counts of the threaded ("dask") mode only
            plan = tuple(("dask", int(j)) for j in os.environ["JOBS"].split(","))
        if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
            continue
        for par, jobs in plan:
            t0 = time.perf_counter()
            T = enstop_amd.ensemble_of_topics(X, k, n_runs=runs, parallelism=par, n_jobs=jobs, n_iter=50,
                                              n_iter_per_test=10, tolerance=0.0, random_state=7)
            dt = time.perf_counter() - t0
            assert T.shape == (runs * k, m) and np.isfinite(T).all()
            print(json.dumps({"config": name, "nnz": int(X.nnz), "parallelism": par, "n_jobs": jobs, "n_iter": 50,
                              "seconds": round(dt, 3), "fits_per_min_1gpu": round(runs / dt * 60, 1)}), flush=True)


if __name__ == "__main__":
    main()
