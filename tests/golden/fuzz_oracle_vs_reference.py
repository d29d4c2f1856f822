#!/usr/bin/env python3
"""Randomised pin of the CPU oracle against THE REFERENCE ITSELF (build container only: imports /root/reference
under the numba shim of make_golden.py).  The 34 committed fixtures pin the oracle on hand-picked cases; this
script draws hundreds of small random problems -- shapes, k, thresholds, sample weights, tolerances, test
intervals, empty words -- runs enstop/plsa.py (fit, refit), enstop/streamed_plsa.py (fit, refit) and
enstop/block_parallel_plsa.py (fit) on each and demands from oracle/plsa_oracle.c

    bit-identical factors, identical iteration counts, log-likelihood traces equal to float32 rounding.

Nothing is stored: the script prints one summary line per family; a copy of its output is kept under profiles/.
usage:  python tests/golden/fuzz_oracle_vs_reference.py [cases=300] [seed=1]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg                                      # noqa: E402  (installs the shims, imports the reference)
import enstop.streamed_plsa as st                             # noqa: E402
import enstop.block_parallel_plsa as bp                       # noqa: E402
from oracle.plsa_oracle import Oracle                         # noqa: E402

ref = mg.ref


def corpus(rs):
    n, m = int(rs.randint(3, 40)), int(rs.randint(3, 50))
    X = sp.random(n, m, density=float(rs.uniform(0.05, 0.5)), random_state=rs, format="csr", dtype=np.float64)
    X.data = np.ceil(X.data * rs.randint(1, 9))
    X = X.tolil()
    for r in range(n):                                        # the reference's fit needs non-empty documents
        if X[r].nnz == 0:
            X[r, rs.randint(m)] = 1.0
    X = X.tocsr()
    X.sort_indices()
    return X


def same_trace(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    if a.shape != b.shape:
        return False
    fin = np.isfinite(b)
    return np.array_equal(np.isfinite(a), fin) and np.array_equal(a[~fin], b[~fin]) and \
        np.allclose(a[fin], b[fin], rtol=3e-6, atol=0)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rs = np.random.RandomState(seed)
    o = Oracle()
    o.set_threads(1)
    tally = {f: [0, 0] for f in ("plsa.plsa_fit", "plsa.plsa_refit", "streamed.plsa_fit", "streamed.plsa_refit",
                                 "block_parallel.plsa_fit")}
    t0 = time.time()
    for case in range(cases):
        X = corpus(rs)
        n, m = X.shape
        k = int(rs.randint(1, 9))
        weighted = rs.rand() < 0.4
        sw = (0.25 + 2 * rs.rand(n)).astype(np.float32) if weighted else np.ones(n, np.float32)
        thresh = float(rs.choice([1e-32, 1e-32, 1e-16, 1e-3, 0.0]))
        kw = dict(n_iter=int(rs.randint(1, 16)), n_iter_per_test=int(rs.randint(1, 7)),
                  tolerance=float(rs.choice([0.0, 1e-3, 1e-2])), e_step_thresh=thresh)
        fit_seed = int(rs.randint(1 << 30))
        with np.errstate(divide="ignore", invalid="ignore"):
            # ---- enstop/plsa.py plsa_fit
            with mg.Recorder() as rec:
                U, V = ref.plsa_fit(X, k, sw, random_state=fit_seed, **kw)
            Uo, Vo, tr, it = o.plsa_fit(X, k, sw, random_state=fit_seed, return_trace=True, **kw)
            ok = np.array_equal(U, Uo) and np.array_equal(V, Vo) and it == rec.n_e and same_trace(tr, rec.ll)
            tally["plsa.plsa_fit"][0] += 1; tally["plsa.plsa_fit"][1] += int(not ok)
            # ---- enstop/plsa.py plsa_refit against the topics just fitted
            with mg.Recorder() as rec:
                R = ref.plsa_refit(X, V, sw, random_state=np.random.RandomState(fit_seed), **kw)
            Ro, tr, it = o.plsa_refit(X, V, sw, random_state=np.random.RandomState(fit_seed), return_trace=True, **kw)
            ok = np.array_equal(R, Ro) and it == rec.n_e and same_trace(tr, rec.ll)
            tally["plsa.plsa_refit"][0] += 1; tally["plsa.plsa_refit"][1] += int(not ok)
            # ---- enstop/streamed_plsa.py
            block = int(rs.choice([7, 16, 64, 65536]))
            with mg.StreamRecorder(st) as rec:
                U, V2 = st.plsa_fit(X, k, sw, block_size=block, random_state=fit_seed, **kw)
            Uo, Vo, tr, it = o.streamed_plsa_fit(X, k, sw, block_size=block, random_state=fit_seed, return_trace=True, **kw)
            ok = np.array_equal(U, Uo) and np.array_equal(V2, Vo) and it == rec.n_em and same_trace(tr, rec.ll)
            tally["streamed.plsa_fit"][0] += 1; tally["streamed.plsa_fit"][1] += int(not ok)
            with mg.StreamRecorder(st) as rec:
                R = st.plsa_refit(X, V, sw, block_size=block, random_state=np.random.RandomState(fit_seed), **kw)
            Ro, tr, it = o.streamed_plsa_refit(X, V, sw, block_size=block, random_state=np.random.RandomState(fit_seed),
                                               return_trace=True, **kw)
            ok = np.array_equal(R, Ro) and it == rec.n_em and same_trace(tr, rec.ll)
            tally["streamed.plsa_refit"][0] += 1; tally["streamed.plsa_refit"][1] += int(not ok)
            # ---- enstop/block_parallel_plsa.py: same maths, tile-wise summation order -> compared to rounding, and the
            #      iteration count only where no stop test is live (tolerance 0: its loop has no `change == 0` arm)
            if case % 4 == 0 and n >= 4 and m >= 4:
                kwb = dict(kw, tolerance=0.0)
                Ub, Vb = bp.plsa_fit(X, k, n_row_blocks=2, n_col_blocks=2, random_state=fit_seed, **kwb)
                ones = np.ones(n, np.float32)
                Uo, Vo = o.plsa_fit(X, k, ones, random_state=fit_seed, **dict(kwb, tolerance=-1.0))   # never stops early
                tol = 5e-3 if thresh >= 1e-6 else 2e-5
                ok = np.abs(Ub - Uo).max() <= tol * max(Uo.max(), 1e-30) and np.abs(Vb - Vo).max() <= tol * max(Vo.max(), 1e-30)
                tally["block_parallel.plsa_fit"][0] += 1; tally["block_parallel.plsa_fit"][1] += int(not ok)
    print("oracle vs the reference itself: %d random problems, seed %d, %.0f s" % (cases, seed, time.time() - t0))
    bad = 0
    for fam, (cnt, miss) in tally.items():
        how = "factors to rounding (tile-wise sums)" if fam.startswith("block") else "factors bit-identical, iterations, LL trace"
        print("  %-26s %4d cases  %d mismatches   (%s)" % (fam, cnt, miss, how))
        bad += miss
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
