#!/usr/bin/env python3
"""Where the wall time of the estimator-level calls goes (host preparation vs device work):
PLSA.fit / transform and plsa_fit on host scipy matrices of config 1, 2 and 3 size."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import enstop_amd                                              # noqa: E402
from enstop_amd.engine import Engine                           # noqa: E402


def main():
    for name, (n, m, nnz_t, k) in {"cfg1": (18_846, 173_762, 2_950_000, 20), "cfg2": (100_000, 50_000, 10_000_000, 32),
                                   "cfg3": (1_000_000, 100_000, 100_000_000, 64)}.items():
        with Engine(0) as eng:
            eng.generate_synthetic(n, m, nnz_t, seed=0)
            X = eng.download_active_csr()
        model = enstop_amd.PLSA(n_components=k, n_iter=50, tolerance=0.0, random_state=3)
        model.fit(X[:1000])                                     # warm-up of the library
        for what in ("fit", "transform", "plsa_fit"):
            pr = cProfile.Profile()
            t0 = time.perf_counter()
            pr.enable()
            if what == "fit":
                model.fit(X)
            elif what == "transform":
                model.transform(X)
            else:
                enstop_amd.plsa_fit(X, k, np.ones(n, np.float32), n_iter=50, tolerance=0.0, random_state=3)
            pr.disable()
            dt = time.perf_counter() - t0
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14)
            print("==== %s %s: %.3f s" % (name, what, dt))
            print("\n".join(l for l in s.getvalue().splitlines() if l.strip() and "function calls" not in l)[:2600], flush=True)


if __name__ == "__main__":
    main()
