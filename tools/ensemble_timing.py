#!/usr/bin/env python3
"""End-to-end cost of ensemble members on one GPU: bootstrap gather, structure build (row order,
CSC, items), factor initialisation (host MT19937 = parity mode, or device RNG), the fit, the
download of the topics.  Prints one JSON object per configuration."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enstop_amd.engine import Engine, PLSA_FUSED          # noqa: E402
from enstop_amd.plsa import plsa_init                      # noqa: E402

CONFIGS = {"cfg4(20NG-shaped,k=20)": (18_846, 173_762, 2_950_000, 20, 50, 8),
           "cfg3(1Mx100k,k=64)": (1_000_000, 100_000, 100_000_000, 64, 50, 3)}


def main():
    eng = Engine(0)
    for name, (n, m, nnz_t, k, n_iter, members) in CONFIGS.items():
        nnz = eng.generate_synthetic(n, m, nnz_t, seed=0)
        for init_mode in ("host_mt19937", "device_mt19937", "device_random"):
            t = dict(bootstrap=0.0, init=0.0, fit=0.0, download=0.0)
            for r in range(members):
                rng = np.random.RandomState(100 + r)
                t0 = time.perf_counter()
                eng.bootstrap(rng.randint(0, n, size=n)); eng.synchronize()
                t1 = time.perf_counter()
                if init_mode == "host_mt19937":
                    class S: shape = (n, m)
                    U, V = plsa_init(S, k, rng=rng)
                    eng.set_factors(U.astype(np.float32), V.astype(np.float32))
                elif init_mode == "device_mt19937":
                    eng.init_factors_numpy_stream(k, rng)      # same stream as the host path, bit-identical
                else:
                    eng.init_factors_device(k, 100 + r)
                eng.synchronize()
                t2 = time.perf_counter()
                it, _ = eng.fit(None, n_iter=n_iter, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-16, flags=PLSA_FUSED)
                t3 = time.perf_counter()
                eng.get_factors(want_u=False)
                t4 = time.perf_counter()
                if r == 0:
                    continue            # first member pays allocations
                t["bootstrap"] += t1 - t0; t["init"] += t2 - t1; t["fit"] += t3 - t2; t["download"] += t4 - t3
            cnt = members - 1
            per = {k_: round(v / cnt * 1e3, 2) for k_, v in t.items()}
            total = sum(per.values())
            print(json.dumps({"config": name, "nnz": nnz, "init": init_mode, "n_iter": n_iter, "ms_per_member": per,
                              "total_ms": round(total, 2), "fits_per_min_1gpu": round(60000.0 / total, 2)}), flush=True)


if __name__ == "__main__":
    main()
