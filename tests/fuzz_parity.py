#!/usr/bin/env python3
"""Extended parity fuzz (GPU box): hundreds of seeded random corpora -- odd k, empty / heavy rows and
columns, thresholds, sample weights, refit -- under several structure knobs (E-step traversal, row
items, column item length), each against the pinned CPU oracle.  A superset of
tests/test_hip_parity.py::test_randomised_shapes_vs_oracle, kept out of the suite for its run time.
usage: python tests/fuzz_parity.py [cases] [seed]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.plsa_oracle import Oracle                          # noqa: E402   (checker only)

# (PLSA_REF_HEAVY_MIN: the reference arithmetic's long-column kernel from 24 / 100 entries on -- these corpora have no column of 2048;
#  PLSA_REF_CHUNK: 64- and 1024-addend chunks of the parity-pair chains beside the default 256; PLSA_REF_ROW_TILED: the document
#  pass of the large corpora on these small ones)
KNOBS = [{"PLSA_REF_ROW_TILED": "1"}, {"PLSA_E_ROWS": "1", "PLSA_E_SEG": "0", "PLSA_REF_HEAVY_MIN": "24"},
         {"PLSA_E_ROWS": "1", "PLSA_E_SEG": "8", "PLSA_ROW_ITEMS": "1", "PLSA_ROW_SEG": "8", "PLSA_REF_CHUNK": "64"},
         {"PLSA_COL_SEG": "4", "PLSA_HEAVY_ITEMS": "2", "PLSA_E_ROWS": "1", "PLSA_REF_HEAVY_MIN": "100"},
         {"PLSA_OVERLAP": "0", "PLSA_SORT_ROWS": "0", "PLSA_XCD_SPLIT": "0", "PLSA_ITEM_ORDER": "0", "PLSA_REF_CHUNK": "1024", "PLSA_REF_ROW_TILED": "1"}]


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    oracle = Oracle()
    oracle.set_threads(8)
    rs = np.random.RandomState(seed)
    worst = dict(U=0.0, V=0.0, ll=0.0)
    ref_bits = ref_cases = 0
    t0 = time.time()
    bad = 0
    for case in range(cases):
        knobs = KNOBS[case % len(KNOBS)]
        for k_, v_ in knobs.items():
            os.environ[k_] = v_
        import enstop_amd
        from enstop_amd.engine import reset_engines
        reset_engines()                                        # knobs are read when a context is created
        n = int(rs.randint(1, 600)); m = int(rs.randint(1, 700))
        k = int(rs.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 24, 31, 32, 40, 64, 65, 70, 100, 128, 130, 200, 256, 300]))
        dens = float(rs.choice([0.005, 0.02, 0.1, 0.4]))
        X = sp.random(n, m, density=dens, format="lil", random_state=rs, dtype=np.float64)
        if rs.rand() < 0.7:
            X[rs.randint(n), :] = 1.0
        if rs.rand() < 0.7:
            X[:, rs.randint(m)] = 2.0
        if n > 3 and rs.rand() < 0.5:
            X[rs.randint(n)] = 0
        X = X.tocsr(); X.data = np.ceil(X.data * 3).astype(np.float32); X.eliminate_zeros()
        X = X.astype(np.float32)
        if X.nnz == 0:
            X = sp.csr_matrix(([1.0], ([0], [0])), shape=(n, m), dtype=np.float32)
        sw = (0.5 + rs.rand(n)).astype(np.float32) if rs.rand() < 0.3 else np.ones(n, np.float32)
        thresh = float(rs.choice([1e-32, 1e-16, 1e-6]))
        kw = dict(n_iter=int(rs.randint(1, 12)), n_iter_per_test=int(rs.randint(1, 5)), tolerance=float(rs.choice([0.0, 1e-3])),
                  e_step_thresh=thresh, random_state=int(rs.randint(1000)))
        Uo, Vo, trace, iters = oracle.plsa_fit(X, k, sw, return_trace=True, **kw)
        for mode in (0, 1):
            U, V, info = enstop_amd.plsa_fit(X, k, sw, flags=mode, return_info=True, **kw)
            msg = "case %d mode %d knobs %r: n=%d m=%d k=%d dens=%g thresh=%g %r" % (case, mode, knobs, n, m, k, dens, thresh, kw)
            if info["n_iter"] != iters:
                # legitimate only when the stop test sits on a knife edge: -inf likelihoods, or a converged
                # fit whose likelihood has stopped moving (the `change == 0` arm then fires as soon as two
                # float32 values repeat -- at once with HIP's float64 accumulation, later or never with the
                # reference's noisy float32 sum): every later change of the longer trace must be within
                # that float32 noise (3e-5 relative)
                tr = info["log_likelihood_trace"].astype(np.float64)
                q = min(len(tr), len(trace))
                longer = tr if len(tr) > len(trace) else trace.astype(np.float64)
                if np.all(np.isfinite(longer)) and q >= 2:
                    # (relative to at least 1e-3: a likelihood that is identically zero -- one-word vocabulary --
                    # is +-1e-6 of rounding noise on both sides and never "converges" bit for bit)
                    delta = np.abs(np.diff(longer[q - 2:]))
                    later = np.where(delta <= 1e-4, 0.0, delta / np.maximum(np.abs(longer[q - 1:]), 1e-30))
                    # ... or the deciding test sits ON the tolerance: the relative change of one side within 2e-6 (the float32
                    # likelihood's own rounding, twice) of `tolerance`, the other side on the other side of it
                    a, b = tr[q - 2:q], trace[q - 2:q].astype(np.float64)
                    on_edge = all(abs(abs(x[1] - x[0]) / max(abs(x[1]), 1e-30) - kw["tolerance"]) <= 2e-6 for x in (a, b))
                    if later.max() > 3e-5 and not on_edge:
                        print("ITER MISMATCH", msg, info["n_iter"], iters, "\n   hip   ", tr, "\n   oracle", trace); bad += 1
                continue
            eu = np.abs(U - Uo).max() / max(Uo.max(), 1e-30); ev = np.abs(V - Vo).max() / max(Vo.max(), 1e-30)
            worst["U"] = max(worst["U"], eu); worst["V"] = max(worst["V"], ev)
            tr = info["log_likelihood_trace"].astype(np.float64); q = min(len(tr), len(trace))
            fin = np.isfinite(trace[:q]) & np.isfinite(tr[:q])
            if fin.any():
                # 1e-5 relative (north_star) with two allowances that are properties of the reference's own
                # float32 arithmetic, not of this engine: an absolute floor of 1e-3 (a likelihood that is exactly
                # 0, e.g. a one-word vocabulary, is rounding noise on both sides: +-1e-7 per term) and 3e-5 for the
                # reference's sequential float32 accumulation of norm_pwz over > 1e5 non-zeros / 300-term
                # dot products
                d_ll = np.abs(tr[:q][fin] - trace[:q][fin])
                e_ll = float(np.max(np.maximum(d_ll - 1e-3, 0.0) / np.maximum(np.abs(trace[:q][fin]), 1e-30)))
                if e_ll > 3e-5:
                    print("LL MISMATCH %.2e" % e_ll, msg, "\n   hip   ", tr[:q], "\n   oracle", trace[:q]); bad += 1
                worst["ll"] = max(worst["ll"], e_ll)
            # a threshold as large as 1e-6 sits inside the range of the products P(w|z) P(z|d): entries flip
            # in and out on last-bit differences (in the reference itself, between thread schedules)
            # (each flip moves one responsibility by ~thresh / norm: with k = 200-300 topics and a handful of
            # non-zeros per document that is up to 5e-3 of the largest P(z|d) entry after a few iterations)
            tol = 1e-4 if thresh <= 1e-16 and X.nnz < 50_000 else (1e-3 if thresh <= 1e-16 else 2e-2)
            if eu > tol or ev > tol:
                print("FACTOR MISMATCH %.2e %.2e" % (eu, ev), msg); bad += 1
        # round 6: THE REFERENCE'S ROUNDING (arithmetic="reference_source": PLSA_REFERENCE_SUMS | PLSA_REFERENCE_LL) -- no tolerance:
        # same iteration count (the sequential float32 likelihoods agree to the last place of the logarithm, so a different
        # stop decision needs a test sitting within ~1e-7 of its tolerance) and then both factors BIT FOR BIT
        oracle.set_ll_sequential(True)      # the reference's source on one thread (the E-step keeps its threads: same bits)
        Uo, Vo, trace, iters = oracle.plsa_fit(X, k, sw, return_trace=True, **kw)
        oracle.set_ll_sequential(False)
        U, V, info = enstop_amd.plsa_fit(X, k, sw, return_info=True, arithmetic="reference_source", **kw)
        ref_cases += 1
        msg = "case %d knobs %r: n=%d m=%d k=%d dens=%g thresh=%g %r" % (case, knobs, n, m, k, dens, thresh, kw)
        if info["n_iter"] != iters:
            tr = info["log_likelihood_trace"].astype(np.float64); q = min(len(tr), len(trace))
            fin = np.isfinite(trace[:q]) & np.isfinite(tr[:q])
            e_ll = float(np.max(np.abs(tr[:q][fin] - trace[:q][fin]) / np.maximum(np.abs(trace[:q][fin]), 1e-30))) if fin.any() else 0.0
            print("REFERENCE-ARITHMETIC ITER MISMATCH", msg, info["n_iter"], iters, "ll rel %.2e" % e_ll); bad += 1
        elif not (np.array_equal(U.view(np.uint32), Uo.view(np.uint32)) and np.array_equal(V.view(np.uint32), Vo.view(np.uint32))):
            # (-0.0 against +0.0 cannot occur: all sums are of non-negative terms)
            print("REFERENCE-ARITHMETIC BITS DIFFER %.2e %.2e" % (np.abs(U - Uo).max(), np.abs(V - Vo).max()), msg); bad += 1
        else:
            ref_bits += 1
        if k <= 64 and case % 4 == 0:                          # refit against the oracle
            topics = Vo
            Uo2 = oracle.plsa_refit(X, topics, sw, n_iter=5, n_iter_per_test=2, tolerance=0.0, e_step_thresh=thresh, random_state=3)
            U2 = enstop_amd.plsa_refit(X, topics, sw, n_iter=5, n_iter_per_test=2, tolerance=0.0, e_step_thresh=thresh, random_state=3)
            eu = np.abs(U2 - Uo2).max() / max(Uo2.max(), 1e-30)
            if eu > 1e-4:
                print("REFIT MISMATCH %.2e" % eu, "case", case, knobs); bad += 1
        for k_ in knobs:
            os.environ.pop(k_, None)
    print("cases %d  mismatches %d  worst rel err U %.2e V %.2e LL %.2e  reference arithmetic: %d of %d fits bit-identical to the oracle  (%.0f s)"
          % (cases, bad, worst["U"], worst["V"], worst["ll"], ref_bits, ref_cases, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
