#!/usr/bin/env python3
"""Members of a bootstrap ensemble fitted CONCURRENTLY on one GPU: J engines (contexts, each with its own
HIP streams and buffers) driven by J host threads -- the reference's own model (a pool of nogil fits,
enstop_.py:209-217).  On a small corpus one fit leaves most of the chip idle (0.15 ms per EM iteration at
the 20NG shape); this measures how far concurrency fills it.
    python tools/ensemble_jobs.py [--config 1] [--members 32] [--jobs 1 2 4 8]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from enstop_amd.engine import Engine, PLSA_FUSED  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--members", type=int, default=32)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--jobs", type=int, nargs="+", default=[1, 2, 4, 8])
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
n, m, k = cfg["n"], cfg["m"], cfg["k"]
seed_eng = Engine(0)
seed_eng.generate_synthetic(n, m, cfg["nnz"], seed=0)
X = seed_eng.download_active_csr()
seed_eng.close()


def member(eng, r):
    rng = np.random.RandomState(100 + r)
    eng.bootstrap(rng.randint(0, n, size=n))
    eng.init_factors_numpy_stream(k, rng)
    eng.fit(None, n_iter=a.iters, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-16, flags=PLSA_FUSED)
    return eng.get_factors(want_u=False)[1]


ref = None
for J in a.jobs:
    engines = [Engine(0) for _ in range(J)]
    for e in engines:
        e.upload_csr(X)
    out = [None] * a.members

    def work(j):
        for r in range(j, a.members, J):
            out[r] = member(engines[j], r)
    for j in range(J):                      # warm-up: allocations, structure buffers
        member(engines[j], 0)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(j,)) for j in range(J)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    stack = np.vstack(out)
    if ref is None:
        ref = stack
    same = bool(np.array_equal(stack, ref))
    print(json.dumps({"config": a.config, "members": a.members, "jobs": J, "seconds": round(dt, 4),
                      "ms_per_member": round(dt / a.members * 1e3, 3), "fits_per_min": round(a.members / dt * 60, 1),
                      "identical_to_serial": same}), flush=True)
    for e in engines:
        e.close()
