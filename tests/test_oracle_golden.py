"""Pins the CPU oracle (oracle/plsa_oracle.c) to vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU-only; runs in the build container and on the GPU box."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden, golden_csr, coo_arrays, peak_rel

KERNEL_CASES = ["kernels_k6", "kernels_k8_thresh", "kernels_k20", "kernels_k33"]
FIT_CASES = ["fit_k8_tol0", "fit_k5_earlystop", "fit_k4_weighted", "fit_k8_thresh",
             "fit_k6_tupleinit", "fit_k16_mid", "fit_k20_50it"]


@pytest.mark.parametrize("case", KERNEL_CASES)
def test_e_step_bit_exact(oracle, case):
    g = load_golden(case)
    r, c, v = coo_arrays(golden_csr(g))
    P = np.full_like(g["P"], -7.0)
    oracle.plsa_e_step(r, c, v, g["V"].copy(), g["U"].copy(), P, g["thresh"])
    np.testing.assert_array_equal(P, g["P"])


@pytest.mark.parametrize("case", KERNEL_CASES)
def test_m_steps_bit_exact(oracle, case):
    g = load_golden(case)
    r, c, v = coo_arrays(golden_csr(g))
    n, k = g["U"].shape
    V, U = g["V"].copy(), g["U"].copy()
    a, b = np.zeros(k, np.float32), np.zeros(n, np.float32)
    oracle.plsa_m_step(r, c, v, V, U, g["P"], a, b)
    np.testing.assert_array_equal(V, g["V_m"]); np.testing.assert_array_equal(U, g["U_m"])
    np.testing.assert_array_equal(a, g["norm_pwz"]); np.testing.assert_array_equal(b, g["norm_pdz"])

    V, U = g["V"].copy(), g["U"].copy()
    oracle.plsa_m_step_w_sample_weight(r, c, v, V, U, g["P"], g["sw"], a, b)
    np.testing.assert_array_equal(V, g["V_mw"]); np.testing.assert_array_equal(U, g["U_mw"])
    np.testing.assert_array_equal(a, g["norm_pwz_w"]); np.testing.assert_array_equal(b, g["norm_pdz_w"])

    U = g["U"].copy()
    oracle.plsa_refit_m_step(r, c, v, g["V"].copy(), U, g["P"], np.ones(n, np.float32), b)
    np.testing.assert_array_equal(U, g["U_refit"]); np.testing.assert_array_equal(b, g["norm_pdz_refit"])


@pytest.mark.parametrize("case", KERNEL_CASES)
def test_log_likelihood(oracle, case):
    g = load_golden(case)
    r, c, v = coo_arrays(golden_csr(g))
    n = g["U"].shape[0]
    ones = np.ones(n, np.float32)
    for sw, key, VV, UU in ((ones, "ll_ones", g["V"], g["U"]), (g["sw"], "ll_sw", g["V"], g["U"]),
                            (ones, "ll_after_m", g["V_m"], g["U_m"])):
        got = oracle.log_likelihood(r, c, v, VV, UU, sw)
        if np.isinf(g[key]):
            assert got == g[key]
        else:
            # numpy float32 log vs libm logf: last-ulp differences only
            np.testing.assert_allclose(got, g[key], rtol=2e-6)


@pytest.mark.parametrize("case", FIT_CASES)
def test_fit_matches_reference(oracle, case):
    g = load_golden(case)
    X = golden_csr(g)
    init = (g["U_init"], g["V_init"]) if "U_init" in g else "random"
    U, V, trace, iters = oracle.plsa_fit(X, int(g["k"]), g["sw"], init=init, n_iter=int(g["n_iter"]),
                                        n_iter_per_test=int(g["n_iter_per_test"]),
                                        tolerance=float(g["tol"]), e_step_thresh=float(g["thresh"]),
                                        random_state=int(g["fit_seed"]), return_trace=True)
    assert iters == int(g["iters"])
    assert trace.shape == g["ll_trace"].shape
    fin = np.isfinite(g["ll_trace"])
    np.testing.assert_array_equal(np.isfinite(trace), fin)
    np.testing.assert_allclose(trace[fin], g["ll_trace"][fin], rtol=2e-6)
    # factors never depend on the log-likelihood value (only on the stop decision): bit-exact
    np.testing.assert_array_equal(U, g["U"])
    np.testing.assert_array_equal(V, g["V"])


@pytest.mark.parametrize("case", ["fit_k8_tol0", "fit_k6_tupleinit"])
def test_init_matches_reference(oracle, case):
    g = load_golden(case)
    X = golden_csr(g)
    init = (g["U_init"], g["V_init"]) if "U_init" in g else "random"
    U, V = oracle.plsa_fit(X, int(g["k"]), g["sw"], init=init, n_iter=0, random_state=int(g["fit_seed"]))
    np.testing.assert_array_equal(U, g["U0"]); np.testing.assert_array_equal(V, g["V0"])


@pytest.mark.parametrize("case", ["refit_k6", "refit_k8_weighted"])
def test_refit_matches_reference(oracle, case):
    g = load_golden(case)
    X = golden_csr(g)
    U, trace, iters = oracle.plsa_refit(X, g["topics"], g["sw"], n_iter=int(g["n_iter"]),
                                       n_iter_per_test=int(g["n_iter_per_test"]),
                                       tolerance=float(g["tol"]),
                                       random_state=np.random.RandomState(42), return_trace=True)
    assert iters == int(g["iters"]) == int(g["n_iter"])      # refit never early-stops (plsa.py:913)
    np.testing.assert_allclose(trace, g["ll_trace"], rtol=2e-6)
    np.testing.assert_array_equal(U, g["U"])


def test_member_bootstrap_matches_reference(oracle):
    """enstop_.py:84-115: bootstrap rows with rng.randint, then plsa_fit with the SAME stream."""
    g = load_golden("member_k6")
    X = golden_csr(g)
    n, k = X.shape[0], int(g["k"])
    ones = np.ones(n, np.float32)
    kw = dict(n_iter=int(g["n_iter"]), n_iter_per_test=10, tolerance=0.0, e_step_thresh=float(g["thresh"]))
    rs = np.random.RandomState(5)
    idx = rs.randint(0, n, size=n)
    np.testing.assert_array_equal(idx, g["idx_rs5"])
    _, V = oracle.plsa_fit(X[idx], k, ones, random_state=rs, **kw)
    np.testing.assert_array_equal(V, g["V_rs5"])
    idx = np.random.RandomState(9).randint(0, n, size=n)
    _, V = oracle.plsa_fit(X[idx], k, ones, random_state=9, **kw)
    np.testing.assert_array_equal(V, g["V_int9"])
    _, V = oracle.plsa_fit(X, k, ones, random_state=9, **kw)
    np.testing.assert_array_equal(V, g["V_nobootstrap"])
    rs = np.random.RandomState(21)
    stack = []
    for _ in range(3):
        idx = rs.randint(0, n, size=n)
        stack.append(oracle.plsa_fit(X[idx], k, ones, random_state=rs, **kw)[1])
    np.testing.assert_array_equal(np.vstack(stack), g["V_stack_rs21"])


def test_threaded_oracle_agrees(oracle):
    """The multi-threaded run (used as the timed CPU baseline) only reorders the LL reduction."""
    from oracle.plsa_oracle import Oracle
    g = load_golden("fit_k16_mid")
    X = golden_csr(g)
    o = Oracle(fast=True)
    o.set_threads(4)
    U, V, trace, iters = o.plsa_fit(X, int(g["k"]), g["sw"], n_iter=int(g["n_iter"]),
                                    n_iter_per_test=int(g["n_iter_per_test"]), tolerance=0.0,
                                    random_state=int(g["fit_seed"]), return_trace=True)
    oracle.set_threads(1)
    assert iters == int(g["iters"])
    np.testing.assert_allclose(trace, g["ll_trace"], rtol=1e-5)
    np.testing.assert_allclose(U, g["U"], rtol=0, atol=1e-4 * g["U"].max())
    np.testing.assert_allclose(V, g["V"], rtol=0, atol=1e-4 * g["V"].max())


@pytest.mark.parametrize("variant", ["n64", "wide"])
def test_wide_accumulator_variants_track_the_checker(variant):
    """The diagnostic builds (float64 norms / float64 everything) are the same algorithm: on the
    golden-sized problems, where float32 accumulation is still accurate, they agree with the reference's
    outputs to rounding."""
    from oracle.plsa_oracle import Oracle
    o = Oracle(variant=variant)
    o.set_threads(1)
    for case in ("fit_k8_tol0", "fit_k4_weighted", "fit_k16_mid"):
        g = load_golden(case)
        X = golden_csr(g)
        U, V, trace, iters = o.plsa_fit(X, int(g["k"]), g["sw"], n_iter=int(g["n_iter"]),
                                        n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
                                        e_step_thresh=float(g["thresh"]), random_state=int(g["fit_seed"]),
                                        return_trace=True)
        assert iters == int(g["iters"])
        assert np.abs(U - g["U"]).max() <= 2e-5 * g["U"].max()
        assert np.abs(V - g["V"]).max() <= 2e-5 * g["V"].max()
        np.testing.assert_allclose(trace, g["ll_trace"], rtol=3e-6)


def test_big_fit_pins_the_float32_stagnation(oracle):
    """fit_k4_big (1.5 M non-zeros, k = 4, two iterations, generated by the reference itself): large enough for the
    reference's single float32 running sum `norm_pwz[z] += s` (plsa.py:193) to be visibly inexact.
    (i) the strict oracle still reproduces the reference bit for bit, (ii) the all-float64 build of the same
    algorithm is more than 1e-4 of the largest P(w|z) entry away from it: at this size and beyond, "matches the
    reference" and "matches exact arithmetic" are different statements, and the reference is the inexact side."""
    from oracle.plsa_oracle import Oracle
    g = load_golden("fit_k4_big")
    X = golden_csr(g)
    kw = dict(n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
              e_step_thresh=float(g["thresh"]), random_state=int(g["fit_seed"]), return_trace=True)
    U, V, trace, iters = oracle.plsa_fit(X, int(g["k"]), g["sw"], **kw)
    assert iters == int(g["iters"]) == 2
    np.testing.assert_array_equal(U, g["U"])
    np.testing.assert_array_equal(V, g["V"])
    np.testing.assert_allclose(trace, g["ll_trace"], rtol=2e-5)   # float32 sum of 1.5 M terms: NumPy log vs logf
    wide = Oracle(variant="wide")
    wide.set_threads(1)
    Uw, Vw, _, _ = wide.plsa_fit(X, int(g["k"]), g["sw"], **kw)
    assert peak_rel(g["V"], Vw) > 1e-4, peak_rel(g["V"], Vw)
    assert peak_rel(g["V"], Vw) < 1e-3 and peak_rel(g["U"], Uw) < 1e-4


# ------------------------------------------------------------------------------------------------
# enstop/streamed_plsa.py (fixtures generated by running that module)
# ------------------------------------------------------------------------------------------------
STREAMFIT_CASES = ["streamfit_k6", "streamfit_k4_weighted", "streamfit_k5_earlystop", "streamfit_k8_thresh",
                   "streamfit_k1_zero_change"]


@pytest.mark.parametrize("case", STREAMFIT_CASES)
def test_streamed_fit_matches_reference(oracle, case):
    """oracle_streamed_fit_inner (streamed_plsa.py:469-603, blocks of `block_size` non-zeros) reproduces the
    reference's streamed fit bit for bit, iteration count included; where plsa.py's loop ran the same number of
    iterations its restatement gives the SAME bits (one summation order) -- they part on the `change == 0` arm."""
    g = load_golden(case)
    X = golden_csr(g)
    kw = dict(n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
              e_step_thresh=float(g["thresh"]), random_state=int(g["fit_seed"]), return_trace=True)
    U, V, trace, iters = oracle.streamed_plsa_fit(X, int(g["k"]), g["sw"], block_size=int(g["block_size"]), **kw)
    assert iters == int(g["iters"])
    np.testing.assert_array_equal(U, g["U"])
    np.testing.assert_array_equal(V, g["V"])
    assert trace.shape == g["ll_trace"].shape
    np.testing.assert_allclose(trace, g["ll_trace"], rtol=2e-6)
    assert bool(g["plsa_py_same_factors"])
    U2, V2, _, iters2 = oracle.plsa_fit(X, int(g["k"]), g["sw"], **kw)
    assert iters2 == int(g["plsa_py_iters"])
    if case == "streamfit_k1_zero_change":
        assert iters2 < iters                        # plsa.py:635 stops on `change == 0`, :596-597 does not
    else:
        assert iters2 == iters
    np.testing.assert_array_equal(U2, g["U"]); np.testing.assert_array_equal(V2, g["V"])


@pytest.mark.parametrize("case", ["streamrefit_k6", "streamrefit_k8_weighted_thresh"])
def test_streamed_refit_matches_reference(oracle, case):
    g = load_golden(case)
    X = golden_csr(g)
    U, trace, iters = oracle.streamed_plsa_refit(X, g["topics"], g["sw"], block_size=int(g["block_size"]),
                                                n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]),
                                                tolerance=float(g["tol"]), e_step_thresh=float(g["thresh"]),
                                                random_state=np.random.RandomState(42), return_trace=True)
    assert iters == int(g["iters"]) == int(g["n_iter"])          # never stops early (streamed_plsa.py:949)
    np.testing.assert_array_equal(U, g["U"])
    np.testing.assert_allclose(trace, g["ll_trace"], rtol=2e-6)
    # the caller's e_step_thresh never reaches the E-step (streamed_plsa.py:932-943) ...
    np.testing.assert_array_equal(g["U"], g["U_default_thresh"])
    # ... while plsa.py's refit uses it: same bits at the default, different vectors at 2e-3
    Up = oracle.plsa_refit(X, g["topics"], g["sw"], n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]),
                           tolerance=float(g["tol"]), e_step_thresh=float(g["thresh"]),
                           random_state=np.random.RandomState(42))
    np.testing.assert_array_equal(Up, g["U_plsa_py"])
    assert np.array_equal(g["U_plsa_py"], g["U"]) == (float(g["thresh"]) == 1e-32)


# ------------------------------------------------------------------------------------------------
# the reference itself at BASELINE config 1's exact shape (fit_cfg1_shape.npz, round 4)
# ------------------------------------------------------------------------------------------------
def cfg1_reference_views(g, U, V):
    """what the fixture keeps of a fit: P(z|d) in full, 4000 columns of P(w|z), its row sums and maxima"""
    return dict(U=U, V_sample=np.ascontiguousarray(V[:, g["V_cols"]]),
                V_rowsum64=V.astype(np.float64).sum(axis=1), V_max=V.max(axis=1),
                V_abs_checksum64=np.float64(np.abs(V.astype(np.float64)).sum()))


def test_cfg1_shape_reference_run_is_reproduced_bit_for_bit(oracle):
    """fit_cfg1_shape: enstop/plsa.py plsa_fit run by the reference on the 18 846 x 173 762 / 2 948 108-nnz corpus of
    BASELINE config 1 (k = 20, two iterations, 226 s of pure-Python time).  (i) the strict oracle reproduces every
    stored output bit for bit -- the inference "oracle == reference at BASELINE sizes" now rests on the reference's
    own output at a BASELINE size; (ii) the float64 build of the same algorithm is 1e-4 ... 1e-2 of the largest
    P(w|z) entry away: the reference's float32 running sums (plsa.py:193, 322) are the inexact side here."""
    import hashlib
    from oracle.plsa_oracle import Oracle
    g = load_golden("fit_cfg1_shape")
    X = golden_csr(g)
    assert X.shape == (18846, 173762) and X.nnz == 2948108
    h = hashlib.sha256()
    for a in (X.indptr.astype(np.int32), X.indices.astype(np.int32), X.data.astype(np.float32)):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == str(g["corpus_sha256"])
    kw = dict(n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
              e_step_thresh=float(g["thresh"]), random_state=int(g["fit_seed"]), return_trace=True)
    sw = np.ones(X.shape[0], np.float32)
    U, V, trace, iters = oracle.plsa_fit(X, int(g["k"]), sw, **kw)
    assert iters == int(g["iters"]) == 2
    got = cfg1_reference_views(g, U, V)
    for key in ("U", "V_sample", "V_rowsum64", "V_max", "V_abs_checksum64"):
        np.testing.assert_array_equal(got[key], g[key], err_msg=key)
    # one float32 running sum over 2.9 M terms; NumPy's float32 log vs logf differ in the last ulp per term
    np.testing.assert_allclose(trace, g["ll_trace"], rtol=2e-5)
    wide = Oracle(variant="wide")
    wide.set_threads(1)
    Uw, Vw, tw, _ = wide.plsa_fit(X, int(g["k"]), sw, **kw)
    gap_v = peak_rel(g["V_sample"], Vw[:, g["V_cols"]])
    gap_u = peak_rel(g["U"], Uw)
    gap_ll = float(np.max(np.abs(tw.astype(np.float64) - g["ll_trace"]) / np.abs(g["ll_trace"])))
    assert 1e-4 < gap_v < 2e-2, gap_v
    assert gap_u < 1e-3, gap_u               # measured 1.6e-3 (P(w|z)), 1.1e-4 (P(z|d))
    assert 1e-4 < gap_ll < 1e-2, gap_ll      # measured 3.4e-3: the float32 running sum of 2.9 M log terms


@pytest.mark.parametrize("case", ["fit_k8_tol0", "fit_k20_50it", "fit_k16_mid", "fit_k4_weighted", "fit_k5_earlystop"])
def test_numba_compiled_fixtures_belong_to_these_inputs(oracle, case):
    """tests/golden/numba_small.npz holds what the reference computes when COMPILED by numba (fastmath, parallel;
    tests/golden/numba_reference.py, build container).  The strict oracle, run from the same initial factors, must sit at
    exactly the recorded distance from it (the oracle IS the sequential source semantics, bit for bit) and stop at the same
    iteration: the fixture belongs to these inputs, and compilation moves the reference by 2e-7 ... 2e-5 here."""
    g = load_golden(case)
    nb = load_golden("numba_small")
    X = golden_csr(g)
    r, c, v = coo_arrays(X)
    sw = g["sw"].astype(np.float32)
    U, V = g["U0"].copy(), g["V0"].copy()
    _, _, trace, iters = oracle.plsa_fit_inner(r, c, v, V, U, sw, n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]),
                                               tolerance=float(g["tol"]), e_step_thresh=float(g["thresh"]),
                                               use_sample_weights=bool(np.any(sw != 1.0)), return_trace=True)
    assert iters == int(nb[case + "__iters"]) == int(g["iters"])
    assert abs(peak_rel(nb[case + "__U"], U) - float(nb[case + "__dev_from_sequential_U"])) < 1e-12
    assert abs(peak_rel(nb[case + "__V"], V) - float(nb[case + "__dev_from_sequential_V"])) < 1e-12
    assert float(nb[case + "__dev_from_sequential_U"]) < 5e-5 and float(nb[case + "__dev_from_sequential_V"]) < 5e-5


def _block_streamed_fixture():
    g = load_golden("numba_block_streamed")
    X = sp.csr_matrix((g["data_u8"].astype(np.float32), g["indices"], g["indptr"]), shape=tuple(int(v) for v in g["shape"]))
    return g, X


def test_numba_compiled_block_and_streamed_modules_midsize(oracle):
    """tests/golden/numba_block_streamed.npz: enstop/block_parallel_plsa.py (8 x 8 tiles) and enstop/streamed_plsa.py
    (11 blocks of 65 536 non-zeros) COMPILED BY NUMBA on a 12 000 x 6 000 corpus (717 k non-zeros, k = 16, 30 iterations;
    tests/golden/numba_reference.py blocks, build container).  The compiled streamed fit equals the compiled plsa.py fit bit
    for bit (recorded there); the oracle's block-streamed restatement of the same loop sits within compile-level rounding of
    it, weights included; the compiled block-parallel fit is itself 1.4e-4 / 5.9e-4 from plsa.py (float32 tile sums)."""
    g, X = _block_streamed_fixture()
    assert bool(g["streamed_equals_plsa_bitwise"])
    n, k, st = X.shape[0], int(g["k"]), int(g["u_stride"])
    kw = dict(n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]), tolerance=0.0, random_state=int(g["fit_seed"]))
    U, V = oracle.streamed_plsa_fit(X, k, np.ones(n, np.float32), block_size=65536, **kw)
    assert peak_rel(U[::st], g["U_streamed"]) < 1e-4 and peak_rel(V, g["V_streamed"]) < 1e-4
    sww = np.exp(np.random.RandomState(int(g["sample_weight_seed"])).uniform(-1, 1, n)).astype(np.float32)
    U, V = oracle.streamed_plsa_fit(X, k, sww, block_size=65536, **kw)
    assert peak_rel(U[::st], g["U_streamed_weighted"]) < 1e-4 and peak_rel(V, g["V_streamed_weighted"]) < 1e-4
    assert 5e-5 < float(g["block_vs_plsa"][1]) < 2e-3


def test_numba_compiled_refit_at_cfg1_shape(oracle):
    """tests/golden/numba_cfg1_refit.npz: `plsa_refit` of the reference COMPILED BY NUMBA on config 1's exact corpus (the
    call PLSA.transform makes: 50 iterations, a test every 5, tolerance 0.001, RandomState(42); fixed topics both sides
    rebuild bit for bit).  The strict oracle sits at the recorded distance from it (1e-6: compilation moves the
    reference that little where no corpus-long float32 sum is involved)."""
    g = load_golden("numba_cfg1_refit")
    X = golden_csr(load_golden("fit_cfg1_shape"))
    n, m = X.shape
    k = int(g["k"])
    w = np.arange(m, dtype=np.int64)[None, :]
    z = np.arange(k, dtype=np.int64)[:, None]
    T = ((w * 7 + z * 131) % 97 + 1).astype(np.float64)
    topics = (T / T.sum(axis=1, keepdims=True)).astype(np.float32)
    assert float(topics.astype(np.float64).sum()) == float(g["topics_checksum"])
    oracle.set_threads(8)
    U = oracle.plsa_refit(X, topics, np.ones(n, np.float32), n_iter=50, n_iter_per_test=5, tolerance=0.001,
                          e_step_thresh=1e-32, random_state=42)
    d = peak_rel(g["U_every_second_row"], U[::2])
    # the recorded figure is over all rows, this one over the stored half of them
    assert d < 1e-5 and 0.5 * float(g["compiled_vs_strict"]) < d < 2.0 * float(g["compiled_vs_strict"]), (d, float(g["compiled_vs_strict"]))


def test_sequential_likelihood_switch_is_the_one_thread_result():
    """oracle_set_ll_sequential: the log-likelihood reduction alone on one thread while the E-step keeps its threads (the at-scale
    GPU tests use it for "the reference's source on one thread"): the same bits as set_threads(1), on a problem large enough
    (1.5 M non-zeros) for the threaded reduction to differ."""
    from oracle.plsa_oracle import Oracle
    g = load_golden("fit_k4_big")
    r, c, v = coo_arrays(golden_csr(g))
    ones = np.ones(g["U"].shape[0], np.float32)
    o = Oracle(variant="strict")
    o.set_threads(1)
    one = o.log_likelihood(r, c, v, g["V"], g["U"], ones)
    o.set_threads(8)
    many = o.log_likelihood(r, c, v, g["V"], g["U"], ones)
    o.set_ll_sequential(True)
    seq = o.log_likelihood(r, c, v, g["V"], g["U"], ones)
    o.set_ll_sequential(False)
    assert seq == one and many != one
    again = o.log_likelihood(r, c, v, g["V"], g["U"], ones)            # the switch is off again: the threaded reduction
    assert again != one and abs(float(again) - float(many)) <= 1e-6 * abs(float(many))   # (OpenMP combines the partials in any order)
