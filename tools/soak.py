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
    with Engine(0) as eng:                                     # config 2: 80 MB of CSR, the staged (page-locked, multi-threaded) copies
        eng.generate_synthetic(100_000, 50_000, 10_000_000, seed=0)
        X2 = eng.download_active_csr()
    ones = np.ones(X.shape[0], np.float32)
    first = enstop_amd.ensemble_of_topics(X, 20, n_runs=4, n_iter=30, tolerance=0.0, random_state=5)
    ref_first = enstop_amd.plsa_fit(X, 20, ones, n_iter=6, tolerance=0.0, random_state=3, arithmetic="reference_source")
    big_first = enstop_amd.plsa_fit(X2, 32, np.ones(X2.shape[0], np.float32), n_iter=5, tolerance=0.0, random_state=4)
    marks = []
    t0 = time.perf_counter()
    for rep in range(12):
        T = enstop_amd.ensemble_of_topics(X, 20, n_runs=32, n_iter=30, tolerance=0.0, random_state=5)
        np.testing.assert_array_equal(T[:80], first)
        model = enstop_amd.PLSA(n_components=16, n_iter=20, random_state=1).fit(X)
        model.transform(X[:5000])
        U, V = enstop_amd.plsa_fit(X, 33, np.ones(X.shape[0], np.float32), n_iter=10, random_state=2, flags=0)   # materialised
        # round 6: the reference arithmetic (its chains, its own buffers) and the staged copies, bit-reproducible from call to call
        Ur, Vr = enstop_amd.plsa_fit(X, 20, ones, n_iter=6, tolerance=0.0, random_state=3, arithmetic="reference_source")
        np.testing.assert_array_equal(Ur, ref_first[0]); np.testing.assert_array_equal(Vr, ref_first[1])
        Ub, Vb = enstop_amd.plsa_fit(X2, 32, np.ones(X2.shape[0], np.float32), n_iter=5, tolerance=0.0, random_state=4)
        np.testing.assert_array_equal(Ub, big_first[0]); np.testing.assert_array_equal(Vb, big_first[1])
        marks.append(free_gb())
        print("round %2d  free HBM %.3f GB  elapsed %.1f s" % (rep, marks[-1], time.perf_counter() - t0), flush=True)
    drift = marks[1] - marks[-1]
    print("drift between round 1 and round %d: %.3f GB" % (len(marks) - 1, drift))
    assert abs(drift) < 0.05, "device memory is creeping"
    print("soak ok")


if __name__ == "__main__":
    main()
