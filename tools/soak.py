#!/usr/bin/env python3
"""Soak test: many ensemble members and estimator calls in one process; device memory must not creep
and results must stay bit-reproducible.  Prints free HBM (torch.cuda.mem_get_info) along the way."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                   # noqa: E402
import enstop_amd                                              # noqa: E402
from enstop_amd.engine import Engine                           # noqa: E402


def free_gb():
    return torch.cuda.mem_get_info(0)[0] / 1e9


def main():
    with Engine(0) as eng:
        eng.generate_synthetic(18_846, 173_762, 2_950_000, seed=0)
        X = eng.download_active_csr()
    first = enstop_amd.ensemble_of_topics(X, 20, n_runs=4, n_iter=30, tolerance=0.0, random_state=5)
    marks = []
    t0 = time.perf_counter()
    for rep in range(12):
        T = enstop_amd.ensemble_of_topics(X, 20, n_runs=32, n_iter=30, tolerance=0.0, random_state=5)
        np.testing.assert_array_equal(T[:80], first)
        model = enstop_amd.PLSA(n_components=16, n_iter=20, random_state=1).fit(X)
        model.transform(X[:5000])
        U, V = enstop_amd.plsa_fit(X, 33, np.ones(X.shape[0], np.float32), n_iter=10, random_state=2, flags=0)   # materialised
        marks.append(free_gb())
        print("round %2d  free HBM %.3f GB  elapsed %.1f s" % (rep, marks[-1], time.perf_counter() - t0), flush=True)
    drift = marks[1] - marks[-1]
    print("drift between round 1 and round %d: %.3f GB" % (len(marks) - 1, drift))
    assert abs(drift) < 0.05, "device memory is creeping"
    print("soak ok")


if __name__ == "__main__":
    main()
