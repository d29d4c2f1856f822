"""HIP path against the CPU oracle AT THE BASELINE.json CONFIGURATIONS, at the sizes AND iteration counts BASELINE.json
states (needs a real MI355X: -m gpu).  What is compared with what (round 5):

  config 1  20NG-shaped synthetic CSR 18 846 x 173 762, 2.95 M nnz, k = 20
            * 50 iterations (BASELINE: "50 EM iters"), tolerance 0, both schedules vs strict / n64 / wide      [asserted]
            * the DEFAULT tolerance 1e-3 (PLSA(): n_iter 100, test every 10): the iteration the fit stops at, HIP (both
              schedules) == n64 == wide [asserted]; strict on 1 thread and on N threads [recorded: the reference's float32
              log-likelihood carries an error larger than the tolerance there]
            * the reference's OWN output, 2 iterations (tests/golden/fit_cfg1_shape.npz)                        [asserted]
            * the reference COMPILED BY NUMBA (tests/golden/numba_cfg1.npz): 50 iterations, and its default-tolerance
              stop iteration on 1 / 2 / 4 / 8 threads (61 on all) == HIP's                                      [asserted]
            * plsa_refit of the reference COMPILED BY NUMBA (tests/golden/numba_cfg1_refit.npz): P(z|d) within the literal
              north-star tolerance 1e-4                                                                          [asserted]
            * refit (PLSA.transform's loop: 50 iterations, test every 5, tolerance 0.005, never stops early) of
              P(z|d) against fixed topics, both schedules vs strict / n64 / wide                                [asserted]
            * a fit with document weights (plsa_m_step_w_sample_weight), 10 iterations, both schedules             [asserted]
            * the estimator as a user types it: PLSA(n_components=20, random_state=7).fit(X) with every default, then
              transform of 2 000 documents: stop iteration, components_, embedding_, transformed rows vs the oracle's
              restatement of the drivers                                                                         [asserted]
  config 2  synthetic CSR 100 k x 50 k, 10 M nnz, k = 32       3 iterations, both schedules vs strict / n64 / wide
            * the document-sharded fit (4 row shards as 4 engines, accumulate / sum / finish per iteration), 6 iterations,
              vs the oracle's plsa_fit and vs the unsharded fit                                                  [asserted]
            * one E-step over all 10 M non-zeros with a threshold inside the products' range: zero pattern of
              P(z|w,d) identical to the oracle's, entry for entry                                                [asserted]
  config 3  1 M x 100 k, 100 M nnz, k = 64: the WHOLE corpus, 2 iterations, both schedules vs n64               [asserted]
            (the first 150 000 documents vs strict / wide stay as the quick check)
  config 4  ensemble_of_topics(n_runs = 32) on the config-1 corpus: stack == serial members (bitwise), one member vs oracle
  config 5  5 M x 200 k, 500 M nnz, k = 128: the first 500 000 documents (50 M nnz, full vocabulary) vs the
            block-streamed oracle (enstop/streamed_plsa.py's loop: no nnz x k array) in n64 / wide arithmetic  [asserted];
            the full size through size-independent properties
  long run  a topical corpus (plsa_generate_synthetic_topics), k = 20, 150 iterations: P(z|d) entries really fall below
            e_step_thresh and become exact zeros; the zero PATTERN of P(z|d) and P(w|z) vs the oracle           [asserted]

The corpora are produced by the engine's deterministic generator (plsa_generate_synthetic[_topics]) and
downloaded for the oracle.  Every comparison is made against three builds of the one oracle source:

  strict  the reference's arithmetic: every accumulator float32, M-step scatter sequential
          (plsa.py:182-194).  At these sizes the float32 running sum norm_pwz[z] += s over ALL nnz
          (plsa.py:193) loses 3-4 digits -- the reference is the inaccurate side.
  n64     the same, norm_pwz and the log-likelihood accumulator in float64
  wide    every accumulator float64: the exact-arithmetic limit of the algorithm

Asserted: HIP == wide and HIP == n64 inside the north-star tolerances (factors 1e-4 of the largest entry,
log-likelihood 1e-5 relative; measured ~1e-6), and HIP's distance to strict is no larger than strict's
own distance to wide (i.e. the gap IS the reference's rounding, not a defect of the port).
Round 6: every BASELINE-sized leg also runs in THE REFERENCE'S ROUNDING (PLSA_REFERENCE_SUMS, enstop_amd/csrc/plsa_ref_kernels.hpp)
and asserts the literal north-star tolerances there: bit-identical factors against the strict oracle (configs 1, 2, 3-sample)
and against the reference's own run (fit_cfg1_shape.npz), 1e-4 / 1e-5 against the numba-compiled reference after 50 iterations.
Every figure is written to gpurun_out/r06_parity_at_scale.json (copied to profiles/ by hand).
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
        with open(os.path.join(REPORT_DIR, "r06_parity_at_scale.json"), "w") as f:
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


def _numba_fixture_leg(X, fixture, results, rec):
    """Round 6: RESULT fixtures of the reference COMPILED BY NUMBA above config 1 (tests/golden/numba_cfg2.npz,
    numba_cfg3_sample.npz; tests/golden/numba_reference.py at_scale).  `results` = {label: (U, V, trace)} of fits of this very
    corpus from RandomState(42) with the fixture's iteration count and a likelihood test after every iteration.  Asserted:
    (i) the engine's generator still produces the corpus the compiled reference was run on (sha256); (ii) in THE REFERENCE'S
    ROUNDING ("reference_arithmetic") north_star's literal tolerances against the compiled reference: 1e-4 on the sampled
    P(z|d) rows and P(w|z) columns, 1e-5 on every tested log-likelihood; (iii) in the DEFAULT arithmetic (float64 norm_pwz /
    likelihood) the relaxed bound -- no further from the compiled reference than it is from exact arithmetic."""
    import hashlib
    from conftest import load_golden, peak_rel
    g = load_golden(fixture)
    h = hashlib.sha256()
    for a in (X.indptr.astype(np.int32), X.indices.astype(np.int32), X.data.astype(np.float32)):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == str(g["corpus_sha256"]), "the synthetic corpus of %s changed" % fixture
    rows, cols = g["U_rows"], g["V_cols"]
    out = rec.setdefault("vs_numba_compiled_reference", {
        "fixture": fixture, "numba_version": str(g["numba_version"]),
        "compiled_reference_vs_exact": {"U": float(g["vs_wide_U"]), "V": float(g["vs_wide_V"]), "ll_rel": float(g["vs_wide_ll"])},
        "compiled_reference_vs_strict_oracle": {"U": float(g["vs_strict_U"]), "V": float(g["vs_strict_V"])}})
    for label, (U, V, trace) in results.items():
        out[label] = {"U": peak_rel(U[rows], g["U_sample"]), "V": peak_rel(V[:, cols], g["V_sample"]),
                      "V_rowsum": float(np.abs(V.astype(np.float64).sum(axis=1) - g["V_rowsum64"]).max()),
                      "ll_rel": ll_rel(np.asarray(trace)[:len(g["ll_trace"])], g["ll_trace"])}
    _flush_report()
    for label, e in out.items():
        if not isinstance(e, dict) or "ll_rel" not in e or label.startswith("compiled_"):
            continue
        if label == "reference_arithmetic":
            assert e["U"] <= 1e-4 and e["V"] <= 1e-4 and e["ll_rel"] <= 1e-5, (fixture, label, e)
        else:
            w = out["compiled_reference_vs_exact"]
            assert e["U"] <= 1.5 * w["U"] + 2e-5 and e["V"] <= 1.5 * w["V"] + 2e-5, (fixture, label, e, w)
            assert e["ll_rel"] <= 1.5 * w["ll_rel"] + 1e-5, (fixture, label, e, w)


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return bool(a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)))


def _reference_arithmetic_leg(amd, oracles, eng, coo, U0, V0, strict, rec, n_iter, n_iter_per_test):
    """PLSA_REFERENCE_SUMS at a BASELINE size: the fit must return the strict oracle's factors BIT FOR BIT (the strict oracle =
    the reference's source semantics, pinned bit for bit by the reference-generated fixtures), i.e. the literal north-star
    tolerance against the reference's own arithmetic with nothing to spare and nothing missing.  Then the two likelihoods on
    those factors: the sequential float32 one (PLSA_REFERENCE_LL) against the oracle on ONE thread, the float64-accumulated
    one against the oracle's float64 accumulator.  `strict` = (U, V, trace, iters) of the strict oracle from (U0, V0)."""
    r, c, v = coo
    ones = np.ones(U0.shape[0], np.float32)
    eng.set_factors(U0, V0)
    t0 = time.time()
    iters, trace = eng.fit(None, n_iter=n_iter, n_iter_per_test=n_iter_per_test, tolerance=0.0, e_step_thresh=1e-32,
                           flags=amd.PLSA_FUSED | amd.PLSA_REFERENCE_SUMS, trace=True)
    sec = time.time() - t0
    U, V = eng.get_factors()
    out = rec.setdefault("reference_arithmetic", {})
    out.update({"seconds": round(sec, 3), "ms_per_iteration": round(1e3 * sec / max(iters, 1), 2),
                "U_bits_equal_strict_oracle": bits_equal(U, strict[0]), "V_bits_equal_strict_oracle": bits_equal(V, strict[1]),
                "vs_strict": {"U": errs(U, strict[0]), "V": errs(V, strict[1])}})
    o = oracles["strict"]
    o.set_ll_sequential(True)
    ll_seq_o = float(o.log_likelihood(r, c, v, strict[1], strict[0], ones))
    o.set_ll_sequential(False)
    ll64_o = float(oracles["n64"].log_likelihood(r, c, v, strict[1], strict[0], ones))
    ll64_h = float(eng.log_likelihood())                       # the engine holds the fit's final factors
    eng.set_arithmetic("reference_source")
    ll_seq_h = float(np.float32(eng.log_likelihood()))
    eng.set_arithmetic(None)
    out["final_log_likelihood"] = {"sequential_float32": {"hip": ll_seq_h, "oracle_one_thread": ll_seq_o,
                                                          "rel": abs(ll_seq_h - ll_seq_o) / abs(ll_seq_o)},
                                   "float64_accumulator": {"hip": ll64_h, "oracle": ll64_o, "rel": abs(ll64_h - ll64_o) / abs(ll64_o)},
                                   "sequential_vs_float64": abs(ll_seq_o - ll64_o) / abs(ll64_o)}
    _flush_report()
    assert iters == strict[3] == n_iter
    assert out["U_bits_equal_strict_oracle"] and out["V_bits_equal_strict_oracle"], out["vs_strict"]
    assert out["final_log_likelihood"]["sequential_float32"]["rel"] <= 1e-5, out["final_log_likelihood"]
    assert out["final_log_likelihood"]["float64_accumulator"]["rel"] <= 1e-6, out["final_log_likelihood"]
    return U, V, trace


def _fit_case(amd, oracles, cfg, name, n_iter, n_iter_per_test, numba_fixture=None):
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
    # n64 still adds the P(w|z) / P(z|d) sums themselves in float32, sequentially (plsa.py:190-191): over 50 iterations
    # at config 1 that alone carries it 1.3e-4 away from exact arithmetic
    rec["n64_vs_wide"] = {"U": errs(ref["n64"][0], ref["wide"][0]), "V": errs(ref["n64"][1], ref["wide"][1]),
                          "ll_rel": ll_rel(ref["n64"][2], ref["wide"][2])}
    results = {}
    with amd.Engine() as eng:
        eng.upload_csr(X)
        for sched, flags in (("fused", amd.PLSA_FUSED), ("materialised", 0)):
            eng.set_factors(U0, V0)
            iters, trace = eng.fit(None, n_iter=n_iter, n_iter_per_test=n_iter_per_test, tolerance=0.0,
                                   e_step_thresh=1e-32, flags=flags, trace=True)
            U, V = eng.get_factors()
            results[sched] = (U, V, trace)
            out = rec.setdefault(sched, {})
            for variant in ("strict", "n64", "wide"):
                Uo, Vo, tr_o, it_o = ref[variant]
                assert iters == it_o == n_iter
                out["vs_" + variant] = {"U": errs(U, Uo), "V": errs(V, Vo), "ll_rel": ll_rel(trace, tr_o)}
            _flush_report()
            e = out["vs_wide"]                   # the north-star tolerances against exact arithmetic
            assert e["U"]["peak_rel"] <= 1e-4 and e["V"]["peak_rel"] <= 1e-4, (sched, e)
            assert e["ll_rel"] <= 1e-5, (sched, e)
            # distance to the float32 arithmetics (the reference's, and n64's float32 factor sums): bounded by each
            # arithmetic's OWN distance to exact -- at short horizons that bound is itself below 1e-4 for n64
            for variant in ("strict", "n64"):
                s_, w = out["vs_" + variant], rec[variant + "_vs_wide"]
                for f in ("U", "V"):
                    assert s_[f]["peak_rel"] <= 1.5 * w[f]["peak_rel"] + 2e-5, (sched, variant, f, s_[f], w[f])
                assert s_["ll_rel"] <= 1.5 * w["ll_rel"] + 1e-5, (sched, variant, s_["ll_rel"], w["ll_rel"])
        # the reference's rounding as an option (PLSA_REFERENCE_SUMS): the strict oracle's bits at this size
        results["reference_arithmetic"] = _reference_arithmetic_leg(amd, oracles, eng, (r, c, v), U0, V0, ref["strict"], rec,
                                                                    n_iter, n_iter_per_test)
        if numba_fixture:
            _numba_fixture_leg(X, numba_fixture, results, rec)
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
    """BASELINE configs[0]: "PLSA(n_components=20), 50 EM iters" -- all fifty, likelihood test every 10 (the default)."""
    _fit_case(amd, oracles, CONFIG1, "config1_50_iterations", n_iter=50, n_iter_per_test=10)


def test_config1_default_tolerance_stops_where_the_oracle_stops(amd, oracles):
    """The user-visible behaviour of a default `PLSA(n_components=20).fit(X)` at config 1: n_iter = 100, a likelihood
    test every 10 iterations, tolerance 1e-3 (plsa.py:1074-1084, 630-638).  The iteration the fit stops at must be the
    one the algorithm stops at in accurate arithmetic (n64, wide); the float32 reference arithmetic is RECORDED on one
    thread (numba without parallel reductions) and on N threads (prange partial sums): its log-likelihood error at this
    size (3e-3 relative on one thread) exceeds the tolerance, so its own stop iteration depends on the thread count."""
    X = corpus(amd, CONFIG1)
    n, m = X.shape
    k = CONFIG1["k"]
    r, c, v = coo_arrays(X)
    U0, V0 = host_init(n, m, k, 42)
    ones = np.ones(n, np.float32)
    kw = dict(n_iter=100, n_iter_per_test=10, tolerance=1e-3, e_step_thresh=1e-32)
    rec = REPORT.setdefault("config1_default_tolerance", {"shape": [n, m], "nnz": int(X.nnz), "k": k, **kw})
    stops, traces = {}, {}
    # (rounds 4-5 also RECORDED the strict build with its likelihood reduced on N threads -- it stops at 61, like numba's compiled
    #  reduction, profiles/r05_parity_at_scale.json; nothing was asserted on it and the run cost the suite 8 s)
    for name, variant, threads in (("n64", "n64", oracles["threads"]), ("wide", "wide", oracles["threads"]),
                                   ("strict_1_thread", "strict", 1)):
        o = oracles[variant]
        # "one thread": only the log-likelihood reduction depends on the thread count (module docstring) -- it alone runs
        # sequentially, the E-step keeps its threads: the same bits as set_threads(1) in a third of the time
        o.set_ll_sequential(threads == 1)
        t0 = time.time()
        _, _, trace, iters = o.plsa_fit_inner(r, c, v, V0.copy(), U0.copy(), ones, return_trace=True, **kw)
        o.set_ll_sequential(False)
        stops[name], traces[name] = int(iters), [float(x) for x in trace]
        rec.setdefault("oracle_seconds", {})[name] = round(time.time() - t0, 1)
    with amd.Engine() as eng:
        eng.upload_csr(X)
        for sched, flags in (("hip_fused", amd.PLSA_FUSED), ("hip_materialised", 0)):
            eng.set_factors(U0, V0)
            iters, trace = eng.fit(None, flags=flags, trace=True, **kw)
            stops[sched], traces[sched] = int(iters), [float(x) for x in trace]
        # the reference's rounding: with the accurate likelihood, and with the source's sequential float32 one
        for sched, flags in (("hip_reference_sums", amd.PLSA_REFERENCE_SUMS),
                             ("hip_reference_sums_and_sequential_ll", amd.PLSA_REFERENCE_SUMS | amd.PLSA_REFERENCE_LL)):
            eng.set_factors(U0, V0)
            iters, trace = eng.fit(None, flags=flags, trace=True, **kw)
            stops[sched], traces[sched] = int(iters), [float(x) for x in trace]
    rec["stop_iteration"] = stops
    rec["log_likelihood_traces"] = traces
    _flush_report()
    assert stops["hip_fused"] == stops["hip_materialised"] == stops["n64"] == stops["wide"], stops
    nl = len(traces["n64"])
    assert len(traces["hip_fused"]) == nl and ll_rel(traces["hip_fused"], traces["n64"]) <= 1e-5
    assert ll_rel(traces["hip_materialised"][:nl], traces["n64"]) <= 1e-5
    # PLSA_REFERENCE_SUMS | PLSA_REFERENCE_LL is the reference's source on one thread: same stop iteration, same trace
    assert stops["hip_reference_sums_and_sequential_ll"] == stops["strict_1_thread"], stops
    assert ll_rel(traces["hip_reference_sums_and_sequential_ll"], traces["strict_1_thread"]) <= 1e-5


def test_cfg1_reference_run(amd, oracles):
    """BASELINE config 1's exact corpus, fitted by the reference itself (fit_cfg1_shape.npz: plsa_fit, k = 20, seed 7,
    two iterations).  (i) the engine's generator still produces that corpus (sha256 of the CSR arrays); (ii) through
    the reference-level API -- enstop_amd.plsa_fit(X, 20, ones, random_state=7), device MT19937 init included -- both
    schedules land no further from the reference's output than 1.5 x the reference's own distance to exact
    arithmetic + 2e-5, and within 1e-4 / 1e-5 of the exact-arithmetic build."""
    import hashlib
    from conftest import load_golden, golden_csr, peak_rel
    g = load_golden("fit_cfg1_shape")
    Xg = golden_csr(g)
    X = corpus(amd, CONFIG1)
    h = hashlib.sha256()
    for a in (X.indptr.astype(np.int32), X.indices.astype(np.int32), X.data.astype(np.float32)):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == str(g["corpus_sha256"]), "plsa_generate_synthetic(config 1, seed 0) changed"
    assert (Xg != X).nnz == 0
    k, n = int(g["k"]), X.shape[0]
    cols = g["V_cols"]
    ones = np.ones(n, np.float32)
    kw = dict(n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]), tolerance=0.0,
              e_step_thresh=float(g["thresh"]), random_state=int(g["fit_seed"]))
    Uw, Vw, tw, _ = oracles["wide"].plsa_fit(X, k, ones, return_trace=True, **kw)
    ref_vs_exact = {"U": peak_rel(g["U"], Uw), "V": peak_rel(g["V_sample"], Vw[:, cols]),
                    "ll_rel": ll_rel(g["ll_trace"], tw)}
    rec = REPORT.setdefault("config1_reference_run", {"nnz": int(X.nnz), "k": k, "n_iter": int(g["n_iter"]),
                                                      "reference_vs_exact": ref_vs_exact})
    for sched, flags in (("fused", amd.PLSA_FUSED), ("materialised", 0)):
        U, V, info = amd.plsa_fit(X, k, ones, flags=flags, return_info=True, **kw)
        assert info["n_iter"] == int(g["iters"])
        trace = np.asarray(info["log_likelihood_trace"])[:len(g["ll_trace"])]
        vs_ref = {"U": peak_rel(U, g["U"]), "V": peak_rel(V[:, cols], g["V_sample"]),
                  "V_rowsum": float(np.abs(V.astype(np.float64).sum(axis=1) - g["V_rowsum64"]).max()),
                  "ll_rel": ll_rel(trace, g["ll_trace"])}
        vs_exact = {"U": peak_rel(U, Uw), "V": peak_rel(V, Vw), "ll_rel": ll_rel(trace, tw)}
        rec[sched] = {"vs_reference": vs_ref, "vs_exact": vs_exact}
        _flush_report()
        assert vs_exact["U"] <= 1e-4 and vs_exact["V"] <= 1e-4 and vs_exact["ll_rel"] <= 1e-5, (sched, vs_exact)
        for f in ("U", "V"):
            assert vs_ref[f] <= 1.5 * ref_vs_exact[f] + 2e-5, (sched, f, vs_ref, ref_vs_exact)
        assert vs_ref["ll_rel"] <= 1.5 * ref_vs_exact["ll_rel"] + 1e-5, (sched, vs_ref, ref_vs_exact)
    # arithmetic="reference_source": the reference's OWN output at this BASELINE size, bit for bit (factors) and to the last
    # place of the float32 logarithm (its sequential float32 log-likelihood, 3.4e-3 from the exact value)
    U, V, info = amd.plsa_fit(X, k, ones, return_info=True, arithmetic="reference_source", **kw)
    trace = np.asarray(info["log_likelihood_trace"])[:len(g["ll_trace"])]
    rec["reference_arithmetic"] = {"U_bits_equal_reference": bits_equal(U, g["U"]),
                                   "V_sample_bits_equal_reference": bits_equal(V[:, cols], g["V_sample"]),
                                   "vs_reference": {"U": peak_rel(U, g["U"]), "V": peak_rel(V[:, cols], g["V_sample"]),
                                                    "ll_rel": ll_rel(trace, g["ll_trace"])}}
    _flush_report()
    assert info["n_iter"] == int(g["iters"])
    assert rec["reference_arithmetic"]["U_bits_equal_reference"] and rec["reference_arithmetic"]["V_sample_bits_equal_reference"], rec["reference_arithmetic"]
    assert rec["reference_arithmetic"]["vs_reference"]["ll_rel"] <= 1e-5, rec["reference_arithmetic"]


def test_cfg1_numba_compiled_reference(amd, oracles):
    """The reference AS COMPILED BY NUMBA (fastmath, parallel prange; tests/golden/numba_reference.py runs it in the build
    container) at BASELINE config 1's exact corpus: 50 EM iterations from RandomState(42), and the iteration a
    default-tolerance fit stops at on 1 / 2 / 4 / 8 threads.  Asserted: (i) HIP stops at the iteration the compiled
    reference stops at, whatever its thread count (61: numba's SIMD-vectorised float32 likelihood is accurate enough there;
    the strictly sequential float32 sum of the reference's SOURCE semantics stops at 71, see
    test_config1_default_tolerance_stops_where_the_oracle_stops); (ii) after 50 iterations HIP is no further from the
    compiled reference than the compiled reference is from exact arithmetic (its float32 norm_pwz running sum), and the
    compiled reference sits where the strict oracle sits (2e-5: compilation moves it that little at this size)."""
    from conftest import load_golden, peak_rel
    g = load_golden("numba_cfg1")
    X = corpus(amd, CONFIG1)
    n, m = X.shape
    k = int(g["k"])
    cols = g["V_cols"]
    r, c, v = coo_arrays(X)
    U0, V0 = host_init(n, m, k, int(g["fit_seed"]))
    ones = np.ones(n, np.float32)
    Uw, Vw = U0.copy(), V0.copy()
    _, _, tw, _ = oracles["wide"].plsa_fit_inner(r, c, v, Vw, Uw, ones, n_iter=50, n_iter_per_test=10, tolerance=0.0,
                                                 e_step_thresh=1e-32, return_trace=True)
    ref_vs_exact = {"U": peak_rel(g["U50"], Uw), "V": peak_rel(g["V50_sample"], Vw[:, cols]), "ll_rel": ll_rel(g["ll50"], tw)}
    stops_ref = {int(t): int(s) for t, s in zip(g["stop_iteration_threads"], g["stop_iteration"])}
    rec = REPORT.setdefault("config1_numba_compiled_reference", {
        "numba_version": str(g["numba_version"]), "reference_vs_exact_after_50_iterations": ref_vs_exact,
        "reference_default_tolerance_stop_iteration_by_threads": stops_ref})
    assert len(set(stops_ref.values())) == 1
    with amd.Engine() as eng:
        eng.upload_csr(X)
        for sched, flags in (("fused", amd.PLSA_FUSED), ("materialised", 0)):
            eng.set_factors(U0, V0)
            iters, trace = eng.fit(None, n_iter=50, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=flags, trace=True)
            U, V = eng.get_factors()
            assert iters == 50
            vs_ref = {"U": peak_rel(U, g["U50"]), "V": peak_rel(V[:, cols], g["V50_sample"]),
                      "V_rowsum": float(np.abs(V.astype(np.float64).sum(axis=1) - g["V50_rowsum64"]).max()),
                      "ll_rel": ll_rel(trace, g["ll50"])}
            eng.set_factors(U0, V0)
            stop, _ = eng.fit(None, n_iter=100, n_iter_per_test=10, tolerance=1e-3, e_step_thresh=1e-32, flags=flags)
            rec[sched] = {"vs_compiled_reference_after_50_iterations": vs_ref, "default_tolerance_stop_iteration": int(stop)}
            _flush_report()
            assert stop == stops_ref[1], (sched, stop, stops_ref)
            # DEFAULT arithmetic (float64 norm_pwz / likelihood): more accurate than the reference, hence bounded by the
            # reference's own distance to exact arithmetic, not by the north-star figures
            for f in ("U", "V"):
                assert vs_ref[f] <= 1.5 * ref_vs_exact[f] + 2e-5, (sched, f, vs_ref, ref_vs_exact)
            assert vs_ref["ll_rel"] <= 1.5 * ref_vs_exact["ll_rel"] + 1e-5, (sched, vs_ref, ref_vs_exact)
        # THE REFERENCE'S ROUNDING (arithmetic="reference", PLSA_REFERENCE_SUMS): north_star's literal tolerances against the
        # reference as its users run it -- 1e-4 on both factor matrices, 1e-5 on every tested log-likelihood -- after all 50
        # iterations, and the same default-tolerance stop iteration.  (Either schedule flag: the mode has one kernel sequence.)
        for sched, flags in (("reference_arithmetic", amd.PLSA_FUSED | amd.PLSA_REFERENCE_SUMS),
                             ("reference_arithmetic_materialised_flag", amd.PLSA_REFERENCE_SUMS)):
            eng.set_factors(U0, V0)
            iters, trace = eng.fit(None, n_iter=50, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=flags, trace=True)
            U, V = eng.get_factors()
            vs_ref = {"U": peak_rel(U, g["U50"]), "V": peak_rel(V[:, cols], g["V50_sample"]),
                      "V_rowsum": float(np.abs(V.astype(np.float64).sum(axis=1) - g["V50_rowsum64"]).max()),
                      "ll_rel": ll_rel(trace, g["ll50"])}
            rec[sched] = {"vs_compiled_reference_after_50_iterations": vs_ref}
            if flags & amd.PLSA_FUSED:       # (the stop iterations once: the other flag runs the same kernels)
                eng.set_factors(U0, V0)
                stop, _ = eng.fit(None, n_iter=100, n_iter_per_test=10, tolerance=1e-3, e_step_thresh=1e-32, flags=flags)
                rec[sched]["default_tolerance_stop_iteration"] = int(stop)
                assert stop == stops_ref[1], (sched, stop, stops_ref)
            _flush_report()
            assert iters == 50
            assert vs_ref["U"] <= 1e-4 and vs_ref["V"] <= 1e-4, (sched, vs_ref)
            assert vs_ref["ll_rel"] <= 1e-5, (sched, vs_ref)


def test_cfg1_numba_compiled_refit(amd, oracles):
    """`plsa_refit` of the reference COMPILED BY NUMBA at config 1's exact corpus (tests/golden/numba_cfg1_refit.npz: the call
    PLSA.transform makes, plsa.py:1210-1218 -- 50 iterations, a test every 5, tolerance 0.001, RandomState(42) -- against a
    fixed topic matrix both sides rebuild bit for bit).  The refit has no corpus-long float32 sum, so here the north-star
    tolerance holds LITERALLY against the reference as its users run it: P(z|d) within 1e-4 (measured ~5e-6)."""
    from conftest import load_golden, peak_rel
    g = load_golden("numba_cfg1_refit")
    X = corpus(amd, CONFIG1)
    n, m = X.shape
    k = int(g["k"])
    w = np.arange(m, dtype=np.int64)[None, :]
    z = np.arange(k, dtype=np.int64)[:, None]
    T = ((w * 7 + z * 131) % 97 + 1).astype(np.float64)
    topics = (T / T.sum(axis=1, keepdims=True)).astype(np.float32)
    assert float(topics.astype(np.float64).sum()) == float(g["topics_checksum"])
    rec = REPORT.setdefault("config1_numba_compiled_refit", {"shape": [n, m], "k": k})
    for sched, flags in (("fused", amd.PLSA_FUSED), ("materialised", 0)):
        U = amd.plsa_refit(X, topics, np.ones(n, np.float32), n_iter=50, n_iter_per_test=5, tolerance=0.001,
                           e_step_thresh=1e-32, random_state=42, flags=flags)
        rec[sched] = {"vs_compiled_reference": peak_rel(U[::2], g["U_every_second_row"])}
        _flush_report()
        assert rec[sched]["vs_compiled_reference"] <= 1e-4, rec
    rec["compiled_reference_vs_strict_oracle"] = float(g["compiled_vs_strict"])
    _flush_report()


def test_config1_refit_vs_oracle(amd, oracles):
    """SURVEY.md 8f-1 at a BASELINE size: `PLSA.transform` / the refit step of `ensemble_fit` (plsa.py:923-997,
    1184-1220; enstop_.py:567) on config 1's corpus -- topics fixed (taken from a 20-iteration fit), P(z|d) re-estimated from
    the transform's own start (RandomState(42), plsa.py:1214) with ITS loop parameters: 50 iterations, a likelihood test
    every 5, tolerance 0.005 -- and the loop that never stops early (plsa.py:913).  Both schedules vs strict / n64 / wide."""
    X = corpus(amd, CONFIG1)
    n, m = X.shape
    k = CONFIG1["k"]
    r, c, v = coo_arrays(X)
    U0, V0 = host_init(n, m, k, 42)
    ones = np.ones(n, np.float32)
    kw = dict(n_iter=50, n_iter_per_test=5, tolerance=0.005, e_step_thresh=1e-32)
    rec = REPORT.setdefault("config1_refit", {"shape": [n, m], "nnz": int(X.nnz), "k": k, **kw})
    with amd.Engine() as eng:
        eng.upload_csr(X)
        eng.set_factors(U0, V0)
        eng.fit(None, n_iter=20, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=amd.PLSA_FUSED)
        _, topics = eng.get_factors()
        topics = np.ascontiguousarray(topics, np.float32)
        rs = np.random.RandomState(42)                                   # plsa.py:978-981: rand(n, k), rows normalised
        Ut = rs.rand(n, k)
        Ut /= Ut.sum(axis=1, keepdims=True)
        Ut = Ut.astype(np.float32)
        ref = {}
        for variant in ("strict", "n64", "wide"):
            t0 = time.time()
            U = Ut.copy()
            _, trace, iters = oracles[variant].plsa_refit_inner(r, c, v, topics, U, ones, return_trace=True, **kw)
            ref[variant] = (U, trace, iters)
            rec.setdefault("oracle_seconds", {})[variant] = round(time.time() - t0, 2)
        rec["strict_vs_wide"] = {"U": errs(ref["strict"][0], ref["wide"][0]), "ll_rel": ll_rel(ref["strict"][1], ref["wide"][1])}
        for sched, flags in (("fused", amd.PLSA_FUSED), ("materialised", 0)):
            eng.set_factors(Ut, topics)
            iters, trace = eng.refit(None, flags=flags, trace=True, **kw)
            U, V = eng.get_factors()
            np.testing.assert_array_equal(V, topics)                     # the topics are not touched
            out = rec.setdefault(sched, {})
            for variant in ("strict", "n64", "wide"):
                Uo, tr_o, it_o = ref[variant]
                assert iters == it_o == 50, (iters, it_o)                # never stops early (plsa.py:913)
                out["vs_" + variant] = {"U": errs(U, Uo), "ll_rel": ll_rel(trace, tr_o)}
            _flush_report()
            e = out["vs_wide"]
            assert e["U"]["peak_rel"] <= 1e-4 and e["ll_rel"] <= 1e-5, (sched, e)
            s_, w = out["vs_strict"], rec["strict_vs_wide"]
            assert s_["U"]["peak_rel"] <= 1.5 * w["U"]["peak_rel"] + 2e-5, (sched, s_, w)
            assert s_["ll_rel"] <= 1.5 * w["ll_rel"] + 1e-5, (sched, s_, w)


def test_config1_weighted_fit_vs_oracle(amd, oracles):
    """SURVEY.md 8a-3 at a BASELINE size: `plsa_m_step_w_sample_weight` (plsa.py:207-310) through the fit loop on config 1's
    corpus -- document weights in [0.25, 4], a tenth of them exactly 1 and fifty exactly 0 (a weight of zero removes the document
    from P(w|z) and from the likelihood but not from its own P(z|d), plsa.py:292-300; words that occur in such documents only
    then make the reference's likelihood NaN, reproduced) -- 10 iterations, both schedules."""
    X = corpus(amd, CONFIG1)
    n, m = X.shape
    k = CONFIG1["k"]
    r, c, v = coo_arrays(X)
    U0, V0 = host_init(n, m, k, 42)
    rs = np.random.RandomState(9)
    sw = np.exp(rs.uniform(np.log(0.25), np.log(4.0), n)).astype(np.float32)
    sw[rs.rand(n) < 0.1] = 1.0
    sw[rs.choice(n, 50, replace=False)] = 0.0
    kw = dict(n_iter=10, n_iter_per_test=5, tolerance=0.0, e_step_thresh=1e-32)
    rec = REPORT.setdefault("config1_weighted_fit", {"shape": [n, m], "nnz": int(X.nnz), "k": k, **kw})
    ref = {}
    for variant in ("strict", "wide"):
        U, V = U0.copy(), V0.copy()
        _, _, trace, iters = oracles[variant].plsa_fit_inner(r, c, v, V, U, sw, use_sample_weights=True, return_trace=True, **kw)
        ref[variant] = (U, V, trace, iters)
    fin = ~np.isnan(ref["wide"][2])
    rec["strict_vs_wide"] = {"U": errs(ref["strict"][0], ref["wide"][0]), "V": errs(ref["strict"][1], ref["wide"][1]),
                             "ll_rel": ll_rel(ref["strict"][2][fin], ref["wide"][2][fin])}
    with amd.Engine() as eng:
        eng.upload_csr(X)
        for sched, flags in (("fused", amd.PLSA_FUSED), ("materialised", 0)):
            eng.set_factors(U0, V0)
            iters, trace = eng.fit(sw, flags=flags, trace=True, **kw)
            U, V = eng.get_factors()
            out = rec.setdefault(sched, {})
            for variant in ("strict", "wide"):
                Uo, Vo, tr_o, it_o = ref[variant]
                assert iters == it_o == 10
                # a word that occurs in zero-weight documents only loses its whole P(w|z) column in the first M-step; its
                # entries then contribute x * log(0) * 0 = NaN to the likelihood (plsa.py:380-383) -- in the reference, in the
                # oracle and here alike: the tests after the first M-step are NaN on every side, the initial one is finite
                nan_o = np.isnan(tr_o)
                np.testing.assert_array_equal(np.isnan(trace), nan_o)
                assert not nan_o[0] and nan_o[1:].all(), tr_o
                out["vs_" + variant] = {"U": errs(U, Uo), "V": errs(V, Vo), "ll_rel": ll_rel(trace[~nan_o], tr_o[~nan_o]),
                                        "nan_tests": int(nan_o.sum())}
            _flush_report()
            e = out["vs_wide"]
            assert e["U"]["peak_rel"] <= 1e-4 and e["V"]["peak_rel"] <= 1e-4 and e["ll_rel"] <= 1e-5, (sched, e)
            s_, w = out["vs_strict"], rec["strict_vs_wide"]
            for f in ("U", "V"):
                assert s_[f]["peak_rel"] <= 1.5 * w[f]["peak_rel"] + 2e-5, (sched, f, s_[f], w[f])
            assert s_["ll_rel"] <= 1.5 * w["ll_rel"] + 1e-5, (sched, s_["ll_rel"], w["ll_rel"])


def test_config1_estimator_with_its_defaults(amd, oracles):
    """BASELINE configs[0] as a user types it -- `PLSA(n_components=20, random_state=7).fit(X)` on the count matrix (integer
    counts: `standardize_input` leaves them alone, utils.py:276-280), every default in force (100 iterations, test every 10,
    tolerance 1e-3, NumPy's MT19937 stream for the initial factors -- drawn on the DEVICE here, bit-identical) -- and then
    `transform` of 2 000 of the documents with ITS defaults (seed 42, 50 iterations, never stops early).  Against the oracle's
    restatement of the same drivers (plsa.py:643-730, 923-997, 1117-1220) in n64 / wide arithmetic: same stop iteration,
    `components_` / `embedding_` / transformed rows within the north-star tolerances."""
    X = corpus(amd, CONFIG1)
    n, m = X.shape
    k = CONFIG1["k"]
    Xi = X.astype(np.int64)
    model = amd.PLSA(n_components=k, random_state=7)
    emb = model.fit_transform(Xi)
    held = Xi[::9][:2000]
    emb_t = model.transform(held)
    rec = REPORT.setdefault("config1_estimator_defaults", {"shape": [n, m], "nnz": int(X.nnz), "k": k})
    ones = np.ones(n, np.float32)
    for variant in ("n64", "wide"):
        o = oracles[variant]
        U, V, trace, iters = o.plsa_fit(X, k, ones, random_state=7, return_trace=True)
        Ut, tr_t, it_t = o.plsa_refit(held.astype(np.float32), V, np.ones(held.shape[0], np.float32), n_iter=50,
                                      n_iter_per_test=5, tolerance=0.001, e_step_thresh=1e-32, random_state=42, return_trace=True)
        rec["vs_" + variant] = {"stop_iteration": [int(model.n_iter_), int(iters)], "components": errs(model.components_, V),
                                "embedding": errs(emb, U), "transform": errs(emb_t, Ut)}
        _flush_report()
        assert model.n_iter_ == iters, (variant, model.n_iter_, iters)
        if variant == "wide":
            e = rec["vs_wide"]
            assert e["components"]["peak_rel"] <= 1e-4 and e["embedding"]["peak_rel"] <= 1e-4, e
            assert e["transform"]["peak_rel"] <= 1e-4, e
    assert emb.dtype == np.float32 and emb.shape == (n, k) and model.components_.shape == (k, m)


def test_config2_e_step_threshold_pattern(amd, oracles):
    """SURVEY.md 8a-1 at a BASELINE size: one materialising E-step over ALL of config 2 (10 M non-zeros x 32 topics) with a
    threshold INSIDE the range of the products (1e-7: about a sixth of them fall below it) -- the strict `>` of plsa.py:98 decided on the
    float32 product exactly as the reference forms it, so the zero pattern of P(z|w,d) must equal the oracle's entry for entry,
    rows whose products all fail stay all-zero (plsa.py:103-105), and the kept entries agree to rounding."""
    X = corpus(amd, CONFIG2)
    n, m = X.shape
    k = CONFIG2["k"]
    r, c, v = coo_arrays(X)
    U0, V0 = host_init(n, m, k, 42)
    U0[::1000] = 0.0                                         # documents with an all-zero P(z|d): norm == 0
    thresh = np.float32(1e-7)
    Po = np.zeros((X.nnz, k), np.float32)
    oracles["strict"].plsa_e_step(r, c, v, V0, U0, Po, thresh)
    with amd.Engine() as eng:
        eng.upload_csr(X)
        eng.set_factors(U0, V0)
        P = eng.e_step(thresh)
    zero_o = Po == 0.0
    mism = int(np.count_nonzero((P == 0.0) != zero_o))
    rec = REPORT.setdefault("config2_e_step_threshold_pattern", {"shape": [n, m], "nnz": int(X.nnz), "k": k, "thresh": float(thresh)})
    rec.update(zero_fraction=float(zero_o.mean()), pattern_mismatches=mism,
               all_zero_rows=int(np.count_nonzero(~(~zero_o).any(axis=1))),
               max_rel_on_kept=float(np.max(np.abs(P[~zero_o] - Po[~zero_o]) / Po[~zero_o])))
    _flush_report()
    assert 0.05 < rec["zero_fraction"] < 0.95, rec           # the threshold really sits inside the products' range
    assert mism == 0, rec
    assert rec["all_zero_rows"] > 0 and rec["max_rel_on_kept"] <= 3e-6, rec


def test_config2_doc_sharded_fit(amd, oracles):
    """SURVEY.md 8f-4 at a BASELINE size: the document-sharded single fit (the counterpart of distributed_plsa.py:99-131 --
    every shard's un-normalised P(w|z) summed, then normalised identically everywhere) on config 2, four row shards of equal
    nnz as four engines of this process (the exchange goes through the same three ABI calls per iteration the RCCL ranks
    use; a 1-GPU box cannot host four ranks), 6 iterations with a likelihood test every 2.  Against the oracle's `plsa_fit`
    (same seed, same NumPy stream) in wide arithmetic, and against the unsharded fit of the same engine."""
    from enstop_amd.sharded import sharded_plsa_fit
    X = corpus(amd, CONFIG2)
    n, m = X.shape
    k = CONFIG2["k"]
    kw = dict(n_iter=6, n_iter_per_test=2, tolerance=0.0, e_step_thresh=1e-32)
    U, V, info = sharded_plsa_fit(X, k, random_state=3, local_shards=4, return_info=True, **kw)
    Uo, Vo, tr_o, it_o = oracles["wide"].plsa_fit(X, k, np.ones(n, np.float32), random_state=3, return_trace=True, **kw)
    U1, V1 = amd.plsa_fit(X, k, np.ones(n, np.float32), random_state=3, **kw)
    nl = min(len(tr_o), len(info["log_likelihood_trace"]))
    rec = REPORT.setdefault("config2_doc_sharded_fit", {"shape": [n, m], "nnz": int(X.nnz), "k": k, "shards": 4, **kw})
    rec.update(iterations=[int(info["n_iter"]), int(it_o)],
               vs_wide={"U": errs(U, Uo), "V": errs(V, Vo), "ll_rel": ll_rel(info["log_likelihood_trace"][:nl], tr_o[:nl])},
               vs_unsharded={"U": errs(U, U1), "V": errs(V, V1)})
    _flush_report()
    assert info["n_iter"] == it_o == 6
    e = rec["vs_wide"]
    assert e["U"]["peak_rel"] <= 1e-4 and e["V"]["peak_rel"] <= 1e-4 and e["ll_rel"] <= 1e-5, e
    assert rec["vs_unsharded"]["U"]["peak_rel"] <= 1e-5 and rec["vs_unsharded"]["V"]["peak_rel"] <= 1e-5, rec["vs_unsharded"]


def test_config2_fit_vs_oracle(amd, oracles):
    _fit_case(amd, oracles, CONFIG2, "config2", n_iter=3, n_iter_per_test=1, numba_fixture="numba_cfg2")


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
        ref[variant] = (U, V, trace, iters)
    rec["strict_vs_wide"] = {"U": errs(ref["strict"][0], ref["wide"][0]), "V": errs(ref["strict"][1], ref["wide"][1]),
                             "ll_rel": ll_rel(ref["strict"][2], ref["wide"][2])}
    results = {}
    with amd.Engine() as eng:
        eng.upload_csr(X)
        results["reference_arithmetic"] = _reference_arithmetic_leg(amd, oracles, eng, (r, c, v), U0, V0, ref["strict"], rec, 2, 1)
        for sched, flags in (("fused", amd.PLSA_FUSED), ("materialised", 0)):
            eng.set_factors(U0, V0)
            iters, trace = eng.fit(None, n_iter=2, n_iter_per_test=1, tolerance=0.0, e_step_thresh=1e-32, flags=flags,
                                   trace=True)
            U, V = eng.get_factors()
            results[sched] = (U, V, trace)
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
    if os.path.exists(os.path.join(ROOT, "tests", "golden", "numba_cfg3_sample.npz")):
        _numba_fixture_leg(X, "numba_cfg3_sample", results, rec)


def test_config3_full_corpus_vs_oracle(amd, oracles):
    """BASELINE configs[2] WHOLE: 1 M documents x 100 k words, 100 M non-zeros, k = 64 -- two EM iterations of both
    schedules against the oracle in exact (wide) arithmetic; its 25.7 GB P(z|w,d) array lives in host memory, ~20 s per
    iteration on the serial M-step.  This is the asserted config-3 check: a defect above 2^31 bytes / 1e8 entries that both
    schedules share cannot hide here.  Asserted: within 1e-4 / 1e-5 of exact arithmetic.  (Round 5 also ran the n64 build:
    HIP 9.9e-5 from its float32 column sums; since round 6 the reference's float32 arithmetic is compared bit for bit on the
    150 000-document sample instead -- and, ONE iteration, on this whole corpus: PLSA_REFERENCE_SUMS against the strict oracle,
    P(z|d) and P(w|z) bit for bit over all 100 M non-zeros.)"""
    from enstop_amd.engine import reset_engines
    reset_engines()
    with amd.Engine() as eng:
        eng.generate_synthetic(1_000_000, 100_000, 100_000_000, seed=0)
        X = eng.download_active_csr()
    n, m = X.shape
    k = 64
    r, c, v = coo_arrays(X)
    U0, V0 = host_init(n, m, k, 42)
    ones = np.ones(n, np.float32)
    n_iter = 2
    rec = REPORT.setdefault("config3_full_corpus", {"shape": [n, m], "nnz": int(X.nnz), "k": k, "n_iter": n_iter,
                                                    "oracle_threads": oracles["threads"]})
    ref = {}
    # round 6: the exact-arithmetic build only.  (Round 5 also ran the n64 build here -- another 41-second pass -- and found HIP
    # 9.9e-5 from its float32 column sums, profiles/r05_parity_at_scale.json; the reference's float32 arithmetic is now compared
    # BIT FOR BIT on the 150 000-document sample, test_config3_shape_row_sample_vs_oracle, PLSA_REFERENCE_SUMS.)
    for variant in ("wide",):
        t0 = time.time()
        Uo, Vo = U0.copy(), V0.copy()
        _, _, tr_o, it_o = oracles[variant].plsa_fit_inner(r, c, v, Vo, Uo, ones, n_iter=n_iter, n_iter_per_test=1,
                                                           tolerance=0.0, e_step_thresh=1e-32, return_trace=True)
        ref[variant] = (Uo, Vo, tr_o, it_o)
        rec.setdefault("oracle_seconds", {})[variant] = round(time.time() - t0, 1)
    # round 6: THE WHOLE of config 3 in the reference's rounding.  ONE iteration of the strict oracle (another 20 s) and of
    # PLSA_REFERENCE_SUMS: bit for bit -- including the state the reference's float32 norm_pwz is in at this size (recorded: each
    # topic's one running sum over 100 M terms of ~0.02 stops growing once its ulp exceeds twice the term)
    t0 = time.time()
    Us, Vs = U0.copy(), V0.copy()
    _, _, tr_s, it_s = oracles["strict"].plsa_fit_inner(r, c, v, Vs, Us, ones, n_iter=1, n_iter_per_test=1, tolerance=0.0,
                                                        e_step_thresh=1e-32, return_trace=True)
    rec.setdefault("oracle_seconds", {})["strict_one_iteration"] = round(time.time() - t0, 1)
    rec["reference_float32_topics_row_sums_after_one_iteration"] = {
        "min": float(Vs.sum(axis=1, dtype=np.float64).min()), "max": float(Vs.sum(axis=1, dtype=np.float64).max()),
        "exact_arithmetic": float(ref["wide"][1].sum(axis=1, dtype=np.float64).max())}
    with amd.Engine() as eng:
        eng.upload_csr(X)
        _reference_arithmetic_leg(amd, oracles, eng, (r, c, v), U0, V0, (Us, Vs, tr_s, it_s), rec, 1, 1)
        eng.release_scratch()
    del r, c, v, Us, Vs
    with amd.Engine() as eng:
        eng.upload_csr(X)
        for sched, flags in (("fused", amd.PLSA_FUSED), ("materialised", 0)):
            eng.set_factors(U0, V0)
            iters, trace = eng.fit(None, n_iter=n_iter, n_iter_per_test=1, tolerance=0.0, e_step_thresh=1e-32,
                                   flags=flags, trace=True)
            U, V = eng.get_factors()
            out = rec.setdefault(sched, {"U_rowsum_max_dev": float(np.abs(U.sum(axis=1, dtype=np.float64) - 1).max()),
                                         "V_rowsum_max_dev": float(np.abs(V.sum(axis=1, dtype=np.float64) - 1).max())})
            for variant in ("wide",):
                Uo, Vo, tr_o, it_o = ref[variant]
                assert iters == it_o == n_iter
                out["vs_" + variant] = {"U": errs(U, Uo), "V": errs(V, Vo), "ll_rel": ll_rel(trace, tr_o)}
            _flush_report()
            e = out["vs_wide"]
            assert e["U"]["peak_rel"] <= 1e-4 and e["V"]["peak_rel"] <= 1e-4 and e["ll_rel"] <= 1e-5, (sched, e)
            eng.release_scratch()


def test_config5_row_sample_vs_streamed_oracle(amd, oracles):
    """BASELINE configs[4] (5 M x 200 k, 500 M nnz, k = 128): the first 500 000 documents of that corpus -- 50 M
    non-zeros over the full 200 k vocabulary -- against the block-streamed oracle (the loop of
    enstop/streamed_plsa.py:469-603, which never holds an nnz x k array; the config-5 route of the reference's
    block_parallel / streamed classes) in exact (wide) arithmetic; one EM iteration + both likelihoods, both schedules,
    the streamed loop's stop rule (PLSA_STOP_NO_ZERO_ARM)."""
    from enstop_amd.engine import reset_engines
    reset_engines()
    with amd.Engine() as eng:
        eng.generate_synthetic(5_000_000, 200_000, 500_000_000, seed=0)
        eng.bootstrap(np.arange(500_000, dtype=np.int64))
        X = eng.download_active_csr()
    n, m = X.shape
    k = 128
    r, c, v = coo_arrays(X)
    U0, V0 = host_init(n, m, k, 42)
    ones = np.ones(n, np.float32)
    rec = REPORT.setdefault("config5_first_500k_docs", {"shape": [n, m], "nnz": int(X.nnz), "k": k, "n_iter": 1,
                                                        "oracle": "streamed loop, block_size 1048576"})
    ref = {}
    for variant in ("wide",):           # (round 5 also ran the n64 build: 8.3e-6 / 3.5e-5 from exact; one 35-second pass less)
        t0 = time.time()
        U, V = U0.copy(), V0.copy()
        _, _, trace, iters = oracles[variant].streamed_plsa_fit_inner(r, c, v, V, U, ones, block_size=1 << 20, n_iter=1,
                                                                      n_iter_per_test=1, tolerance=0.0,
                                                                      e_step_thresh=1e-32, return_trace=True)
        ref[variant] = (U, V, trace, iters)
        rec.setdefault("oracle_seconds", {})[variant] = round(time.time() - t0, 1)
    del r, c, v
    with amd.Engine() as eng:
        eng.upload_csr(X)
        for sched, flags in (("fused", amd.PLSA_FUSED), ("materialised", 0)):
            eng.set_factors(U0, V0)
            iters, trace = eng.fit(None, n_iter=1, n_iter_per_test=1, tolerance=0.0, e_step_thresh=1e-32,
                                   flags=flags | amd.engine.PLSA_STOP_NO_ZERO_ARM, trace=True)
            U, V = eng.get_factors()
            out = rec.setdefault(sched, {})
            for variant in ("wide",):
                Uo, Vo, tr_o, it_o = ref[variant]
                assert iters == it_o == 1
                out["vs_" + variant] = {"U": errs(U, Uo), "V": errs(V, Vo), "ll_rel": ll_rel(trace, tr_o)}
            _flush_report()
            for variant in ("wide",):
                e = out["vs_" + variant]
                assert e["U"]["peak_rel"] <= 1e-4 and e["V"]["peak_rel"] <= 1e-4 and e["ll_rel"] <= 1e-5, (sched, variant, e)
            eng.release_scratch()


def test_long_run_reaches_the_threshold_regime(amd, oracles):
    """A fit that runs long enough on a corpus WITH topical structure (plsa_generate_synthetic_topics: documents are
    sparse Dirichlet mixtures of 24 latent topics) for P(z|d) entries to decay below e_step_thresh: every product
    P(w|z) P(z|d) of such an entry fails `v > thresh` (plsa.py:97-102) and the entry becomes an exact zero, for good.
    150 iterations, k = 20, default threshold 1e-32.  Asserted: a substantial part of P(z|d) IS exactly zero in the
    oracle (57 % on a host-made corpus of the same model), and the zero pattern of both factors is the oracle's (strict
    arithmetic, one thread: the reference's)."""
    with amd.Engine() as eng:
        nnz = eng.generate_synthetic(6000, 5000, 360_000, seed=5, topics=24, alpha=0.05, background=0.1)
        X = eng.download_active_csr()
    n, m = X.shape
    k = 20
    r, c, v = coo_arrays(X)
    U0, V0 = host_init(n, m, k, 42)
    ones = np.ones(n, np.float32)
    kw = dict(n_iter=150, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32)
    ref = {}
    for variant in ("strict", "wide"):
        o = oracles[variant]
        o.set_threads(1)
        Uo, Vo = U0.copy(), V0.copy()
        _, _, tr_o, it_o = o.plsa_fit_inner(r, c, v, Vo, Uo, ones, return_trace=True, **kw)
        o.set_threads(oracles["threads"])
        ref[variant] = (Uo, Vo, tr_o, it_o)
    Us, Vs = ref["strict"][:2]
    zero_u, zero_v = float((Us == 0).mean()), float((Vs == 0).mean())
    rec = REPORT.setdefault("long_run_threshold_regime", {"shape": [n, m], "nnz": int(nnz), "k": k, **kw,
                                                          "oracle_zero_fraction_U": zero_u, "oracle_zero_fraction_V": zero_v,
                                                          "oracle_smallest_positive_U": float(Us[Us > 0].min())})
    rec["strict_vs_wide"] = {"U": errs(Us, ref["wide"][0]), "V": errs(Vs, ref["wide"][1]),
                             "ll_rel": ll_rel(ref["strict"][2], ref["wide"][2]),
                             "zero_pattern_mismatches_U": int(((Us == 0) != (ref["wide"][0] == 0)).sum()),
                             "zero_pattern_mismatches_V": int(((Vs == 0) != (ref["wide"][1] == 0)).sum())}
    assert zero_u >= 0.05, "the run never reached the threshold regime: %.4f of P(z|d) is zero" % zero_u
    with amd.Engine() as eng:
        eng.upload_csr(X)
        for sched, flags in (("fused", amd.PLSA_FUSED), ("materialised", 0)):
            eng.set_factors(U0, V0)
            iters, trace = eng.fit(None, flags=flags, trace=True, **kw)
            U, V = eng.get_factors()
            out = rec.setdefault(sched, {"zero_fraction_U": float((U == 0).mean()), "zero_fraction_V": float((V == 0).mean())})
            for variant in ("strict", "wide"):
                Uo, Vo, tr_o, it_o = ref[variant]
                assert iters == it_o == 150
                out["vs_" + variant] = {"U": errs(U, Uo), "V": errs(V, Vo), "ll_rel": ll_rel(trace, tr_o),
                                        "zero_pattern_mismatches_U": int(((U == 0) != (Uo == 0)).sum()),
                                        "zero_pattern_mismatches_V": int(((V == 0) != (Vo == 0)).sum())}
            _flush_report()
            # The zero PATTERN against the reference's own arithmetic (strict, one thread).  An entry about to die holds
            # ~1e-30; whether its LAST surviving product passes `> 1e-32` in this iteration or the next can hinge on the
            # 7th digit, which summation order owns (the reference's own prange has the same freedom); such an entry is
            # zero on both sides one iteration later.  Hence: at most 3 disagreeing entries, each negligible (< 1e-25)
            # on the side where it still lives.  Measured: 0 mismatches among 120 000 + 100 000 entries, 59.6 % zeros.
            e = out["vs_strict"]
            assert e["zero_pattern_mismatches_U"] <= 3 and e["zero_pattern_mismatches_V"] <= 3, (sched, e)
            for A, B in ((U, Us), (V, Vs)):
                dis = (A == 0) != (B == 0)
                assert not dis.any() or max(A[dis].max(), B[dis].max()) < 1e-25, (sched, A[dis], B[dis])
            # the VALUES after 150 iterations: the north-star tolerances against exact arithmetic; against the float32
            # reference arithmetic no further than that arithmetic is from exact (its float32 norm_pwz running sum
            # compounds over 150 iterations: 6e-5 on the likelihood)
            e = out["vs_wide"]
            assert e["U"]["peak_rel"] <= 1e-4 and e["V"]["peak_rel"] <= 1e-4 and e["ll_rel"] <= 1e-5, (sched, e)
            e, w = out["vs_strict"], rec["strict_vs_wide"]
            for f in ("U", "V"):
                assert e[f]["peak_rel"] <= 1.5 * w[f]["peak_rel"] + 2e-5, (sched, f, e[f], w[f])
            assert e["ll_rel"] <= 1.5 * w["ll_rel"] + 1e-5, (sched, e, w)


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

