"""HIP path against the CPU oracle AT THE BASELINE.json CONFIGURATIONS (needs a real MI355X: -m gpu).

  config 1  20NG-shaped synthetic CSR 18 846 x 173 762, 2.95 M nnz, k = 20   fit, both schedules, 5 iterations
  config 2  synthetic CSR 100 k x 50 k, 10 M nnz, k = 32                      fit, both schedules, 3 iterations
  config 4  ensemble_of_topics(n_runs = 32) on the config-1 corpus            stack == serial members (bitwise),
                                                                              one member against the oracle
  config 5  5 M x 200 k, 500 M nnz, k = 128                                   size-independent properties
  config 3  its shape on the first 150 000 documents (15 M nnz, k = 64)              fit, both schedules, 2 iterations
            (the full corpus: test_hip_parity.py::test_full_size_properties and bench.py)

The corpora are produced by the engine's deterministic generator (plsa_generate_synthetic) and
downloaded for the oracle.  Every comparison is made against three builds of the one oracle source:

  strict  the reference's arithmetic: every accumulator float32, M-step scatter sequential
          (plsa.py:182-194).  At these sizes the float32 running sum norm_pwz[z] += s over ALL nnz
          (plsa.py:193) loses 3-4 digits -- the reference is the inaccurate side.
  n64     the same, norm_pwz and the log-likelihood accumulator in float64
  wide    every accumulator float64: the exact-arithmetic limit of the algorithm

Asserted: HIP == wide and HIP == n64 inside the north-star tolerances (factors 1e-4 of the largest entry,
log-likelihood 1e-5 relative; measured ~1e-6), and HIP's distance to strict is no larger than strict's
own distance to wide (i.e. the gap IS the reference's rounding, not a defect of the port).
Every figure is written to gpurun_out/r03_parity_at_scale.json (copied to profiles/ by hand).
"""
import json
import os
import time

import numpy as np
import pytest

from conftest import ROOT, coo_arrays

pytestmark = pytest.mark.gpu

REPORT = {}
REPORT_DIR = os.environ.get("PARITY_REPORT_DIR", os.path.join(ROOT, "gpurun_out"))

CONFIG1 = dict(n=18_846, m=173_762, nnz=2_950_000, k=20)
CONFIG2 = dict(n=100_000, m=50_000, nnz=10_000_000, k=32)


def _flush_report():
    try:
        os.makedirs(REPORT_DIR, exist_ok=True)
        with open(os.path.join(REPORT_DIR, "r03_parity_at_scale.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


def errs(a, b):
    """a: HIP (or other) result, b: the side it is compared with.  peak_rel = max |a-b| / max |b|;
    elem_rel_* = elementwise |a-b| / |b| over the entries with |b| >= 1e-3 max |b| (relative error of
    a probability that is itself negligible is not meaningful), maximum and 99.9th percentile."""
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    d = np.abs(a - b)
    peak = max(np.abs(b).max(), 1e-300)
    big = np.abs(b) >= 1e-3 * peak
    rel = d[big] / np.abs(b[big])
    return dict(peak_rel=float(d.max() / peak), elem_rel_max=float(rel.max()) if rel.size else 0.0,
                elem_rel_p999=float(np.percentile(rel, 99.9)) if rel.size else 0.0,
                mean_abs_over_peak=float(d.mean() / peak))


def ll_rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b) / np.abs(b)))


def host_init(n, m, k, seed):
    from enstop_amd.plsa import plsa_init

    class S:
        shape = (n, m)
    U, V = plsa_init(S, k, rng=np.random.RandomState(seed))
    return U.astype(np.float32), V.astype(np.float32)


@pytest.fixture(scope="module")
def amd():
    import enstop_amd
    return enstop_amd


@pytest.fixture(scope="module")
def oracles():
    from oracle.plsa_oracle import Oracle
    out = {}
    threads = max(1, min(os.cpu_count() or 1, 64))
    for v in ("strict", "n64", "wide"):
        o = Oracle(variant=v)
        # the E-step is independent per non-zero (thread count cannot change it); the M-step scatter is
        # serial in every build; only the log-likelihood reduction depends on the thread count (as it
        # does under numba's prange, plsa.py:375)
        o.set_threads(threads)
        out[v] = o
    out["threads"] = threads
    return out


_corpora = {}


def corpus(amd, cfg):
    key = (cfg["n"], cfg["m"], cfg["nnz"])
    if key not in _corpora:
        with amd.Engine() as eng:
            eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], seed=0)
            _corpora[key] = eng.download_active_csr()
    return _corpora[key]


def _fit_case(amd, oracles, cfg, name, n_iter, n_iter_per_test):
    X = corpus(amd, cfg)
    n, m = X.shape
    k = cfg["k"]
    r, c, v = coo_arrays(X)
    U0, V0 = host_init(n, m, k, 42)
    ones = np.ones(n, np.float32)
    rec = REPORT.setdefault(name, {"shape": [n, m], "nnz": int(X.nnz), "k": k, "n_iter": n_iter,
                                   "n_iter_per_test": n_iter_per_test, "oracle_threads": oracles["threads"]})
    ref = {}
    for variant in ("strict", "n64", "wide"):
        t0 = time.time()
        U, V = U0.copy(), V0.copy()
        _, _, trace, iters = oracles[variant].plsa_fit_inner(r, c, v, V, U, ones, n_iter=n_iter,
                                                             n_iter_per_test=n_iter_per_test, tolerance=0.0,
                                                             e_step_thresh=1e-32, return_trace=True)
        ref[variant] = (U, V, trace, iters)
        rec.setdefault("oracle_seconds", {})[variant] = round(time.time() - t0, 2)
    # the reference's own rounding: strict against the exact-arithmetic limit
    rec["strict_vs_wide"] = {"U": errs(ref["strict"][0], ref["wide"][0]), "V": errs(ref["strict"][1], ref["wide"][1]),
                             "ll_rel": ll_rel(ref["strict"][2], ref["wide"][2])}
    rec["strict_vs_n64"] = {"U": errs(ref["strict"][0], ref["n64"][0]), "V": errs(ref["strict"][1], ref["n64"][1]),
                            "ll_rel": ll_rel(ref["strict"][2], ref["n64"][2])}
    with amd.Engine() as eng:
        eng.upload_csr(X)
        for sched, flags in (("fused", amd.PLSA_FUSED), ("materialised", 0)):
            eng.set_factors(U0, V0)
            iters, trace = eng.fit(None, n_iter=n_iter, n_iter_per_test=n_iter_per_test, tolerance=0.0,
                                   e_step_thresh=1e-32, flags=flags, trace=True)
            U, V = eng.get_factors()
            out = rec.setdefault(sched, {})
            for variant in ("strict", "n64", "wide"):
                Uo, Vo, tr_o, it_o = ref[variant]
                assert iters == it_o == n_iter
                out["vs_" + variant] = {"U": errs(U, Uo), "V": errs(V, Vo), "ll_rel": ll_rel(trace, tr_o)}
            _flush_report()
            for variant in ("n64", "wide"):
                e = out["vs_" + variant]
                assert e["U"]["peak_rel"] <= 1e-4 and e["V"]["peak_rel"] <= 1e-4, (sched, variant, e)
                assert e["ll_rel"] <= 1e-5, (sched, variant, e)
            # distance to the float32 reference arithmetic: bounded by that arithmetic's own error
            s, w = out["vs_strict"], rec["strict_vs_wide"]
            for f in ("U", "V"):
                assert s[f]["peak_rel"] <= 1.5 * w[f]["peak_rel"] + 2e-5, (sched, f, s[f], w[f])
            assert s["ll_rel"] <= 1.5 * w["ll_rel"] + 1e-5, (sched, s["ll_rel"], w["ll_rel"])
        # kernel level, one step from the initial factors: norm_pwz and the un-normalised P(w|z)
        eng.set_factors(U0, V0)
        eng.e_step(1e-32, want_host_copy=False)
        npwz, npdz = eng.m_step()
        Um, Vm = eng.get_factors()
        kl = rec.setdefault("one_m_step", {})
        for variant in ("strict", "n64", "wide"):
            o = oracles[variant]
            P = np.zeros((X.nnz, k), np.float32)
            o.plsa_e_step(r, c, v, V0, U0, P, 1e-32)
            Vo, Uo = V0.copy(), U0.copy()
            nw = np.zeros(k, np.float32); nd = np.zeros(n, np.float32)
            o.plsa_m_step(r, c, v, Vo, Uo, P, nw, nd)
            kl["vs_" + variant] = {"norm_pwz": errs(npwz, nw), "norm_pdz": errs(npdz, nd),
                                   "V_unnormalised": errs(Vm.astype(np.float64) * npwz[:, None].astype(np.float64),
                                                          Vo.astype(np.float64) * nw[:, None].astype(np.float64)),
                                   "U": errs(Um, Uo), "V": errs(Vm, Vo)}
            del P
        _flush_report()
        for variant in ("n64", "wide"):
            e = kl["vs_" + variant]
            assert e["norm_pwz"]["elem_rel_max"] <= 1e-5, (variant, e["norm_pwz"])
            assert e["V_unnormalised"]["peak_rel"] <= 1e-5 and e["V"]["peak_rel"] <= 1e-5, (variant, e)
            assert e["U"]["peak_rel"] <= 1e-5, (variant, e)


def test_config1_fit_vs_oracle(amd, oracles):
    _fit_case(amd, oracles, CONFIG1, "config1", n_iter=5, n_iter_per_test=2)


def test_config2_fit_vs_oracle(amd, oracles):
    _fit_case(amd, oracles, CONFIG2, "config2", n_iter=3, n_iter_per_test=1)


def test_config3_shape_row_sample_vs_oracle(amd, oracles):
    """Config 3's shape (100 k vocabulary, k = 64, ~100 entries per document) on the first 150 000 documents of
    the config-3 corpus (15 M nnz: what the serial M-step of the oracle finishes in seconds); the full
    corpus is covered by test_full_size_properties and bench.py."""
    with amd.Engine() as eng:
        eng.generate_synthetic(1_000_000, 100_000, 100_000_000, seed=0)
        eng.bootstrap(np.arange(150_000, dtype=np.int64))
        X = eng.download_active_csr()
    n, m = X.shape
    k = 64
    r, c, v = coo_arrays(X)
    U0, V0 = host_init(n, m, k, 42)
    ones = np.ones(n, np.float32)
    rec = REPORT.setdefault("config3_first_150k_docs", {"shape": [n, m], "nnz": int(X.nnz), "k": k, "n_iter": 2})
    ref = {}
    for variant in ("strict", "wide"):
        U, V = U0.copy(), V0.copy()
        _, _, trace, iters = oracles[variant].plsa_fit_inner(r, c, v, V, U, ones, n_iter=2, n_iter_per_test=1,
                                                             tolerance=0.0, e_step_thresh=1e-32, return_trace=True)
        ref[variant] = (U, V, trace)
    rec["strict_vs_wide"] = {"U": errs(ref["strict"][0], ref["wide"][0]), "V": errs(ref["strict"][1], ref["wide"][1]),
                             "ll_rel": ll_rel(ref["strict"][2], ref["wide"][2])}
    with amd.Engine() as eng:
        eng.upload_csr(X)
        for sched, flags in (("fused", amd.PLSA_FUSED), ("materialised", 0)):
            eng.set_factors(U0, V0)
            iters, trace = eng.fit(None, n_iter=2, n_iter_per_test=1, tolerance=0.0, e_step_thresh=1e-32, flags=flags,
                                   trace=True)
            U, V = eng.get_factors()
            out = rec.setdefault(sched, {})
            for variant in ("strict", "wide"):
                out["vs_" + variant] = {"U": errs(U, ref[variant][0]), "V": errs(V, ref[variant][1]),
                                        "ll_rel": ll_rel(trace, ref[variant][2])}
            _flush_report()
            e = out["vs_wide"]
            assert e["U"]["peak_rel"] <= 1e-4 and e["V"]["peak_rel"] <= 1e-4 and e["ll_rel"] <= 1e-5, (sched, e)
            s_, w_ = out["vs_strict"], rec["strict_vs_wide"]
            for f in ("U", "V"):
                assert s_[f]["peak_rel"] <= 1.5 * w_[f]["peak_rel"] + 2e-5, (sched, f, s_[f], w_[f])


def test_config4_ensemble_on_20ng_shaped_corpus(amd, oracles):
    """EnsembleTopics' member fan-out at its BASELINE size: 32 bootstrapped fits, k = 20."""
    X = corpus(amd, CONFIG1)
    n, m = X.shape
    k = CONFIG1["k"]
    kw = dict(n_iter=20, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-16)
    amd.ensemble_of_topics(X, k, n_runs=4, random_state=7, n_jobs=4, **kw)       # warm-up: contexts, buffers
    t0 = time.time()
    stack = amd.ensemble_of_topics(X, k, n_runs=32, random_state=7, n_jobs=4, **kw)
    dt = time.time() - t0
    t0 = time.time()
    serial = amd.ensemble_of_topics(X, k, n_runs=32, random_state=7, n_jobs=1, **kw)
    dt1 = time.time() - t0
    np.testing.assert_array_equal(stack, serial)          # concurrent members: same stack, bit for bit
    assert stack.shape == (32 * k, m) and stack.dtype == np.float32
    np.testing.assert_allclose(stack.sum(axis=1, dtype=np.float64), 1.0, atol=2e-4)
    rec = REPORT.setdefault("config4", {"n_runs": 32, "k": k, "shape": [n, m], "nnz": int(X.nnz),
                                        "ensemble_seconds_20_iters_4_concurrent_members": round(dt, 3),
                                        "ensemble_seconds_20_iters_one_member_at_a_time": round(dt1, 3)})
    # run r of the ensemble == the standalone member with the r-th derived stream, bit for bit
    for run in (0, 13, 31):
        V = amd.plsa_topics(X, k, random_state=np.random.RandomState(7 + run), **kw)
        np.testing.assert_array_equal(V, stack[run * k:(run + 1) * k])
    # the members are different fits
    assert not np.array_equal(stack[:k], stack[k:2 * k])
    # one member against the oracle: same bootstrap indices, the stream continues into plsa_init
    run = 5
    rng = np.random.RandomState(7 + run)
    idx = rng.randint(0, n, size=n)                                  # enstop_.py:87
    B = X[idx]
    ones = np.ones(n, np.float32)
    short = dict(kw, n_iter=6, n_iter_per_test=2)
    V_hip = amd.plsa_topics(X, k, random_state=np.random.RandomState(7 + run), **short)
    out = {}
    for variant in ("strict", "wide"):
        rs = np.random.RandomState(7 + run)
        rs.randint(0, n, size=n)                                     # consume the bootstrap draw
        Uo, Vo, trace, iters = oracles[variant].plsa_fit(B, k, ones, random_state=rs, return_trace=True, **short)
        out[variant] = Vo
        rec["member_vs_" + variant] = {"V": errs(V_hip, Vo)}
    rec["member_strict_vs_wide"] = {"V": errs(out["strict"], out["wide"])}
    _flush_report()
    assert rec["member_vs_wide"]["V"]["peak_rel"] <= 1e-4
    assert rec["member_vs_strict"]["V"]["peak_rel"] <= 1.5 * rec["member_strict_vs_wide"]["V"]["peak_rel"] + 2e-5


def test_config5_full_size_properties(amd):
    """BASELINE.json configs[4]: 5 M x 200 k, 500 M nnz, k = 128 -- both schedules (the materialised one
    through the 256 GB P(z|w,d) array), same properties as test_full_size_properties."""
    from test_hip_parity import test_full_size_properties
    from enstop_amd.engine import reset_engines
    reset_engines()                      # ~270 of the 288 GB are needed: drop the cached engine's buffers
    _corpora.clear()
    t0 = time.time()
    test_full_size_properties(amd, (5_000_000, 200_000, 500_000_000, 128))
    REPORT["config5"] = {"properties": "schedules agree, likelihood increases, rows sum to one, bit-identical "
                                       "re-run, M-step(E-step) == fused iteration", "seconds": round(time.time() - t0, 1)}
    _flush_report()
