"""Parity of the HIP path (through the C ABI / the reference-shaped Python interface) against
(1) the golden vectors produced by the reference itself and (2) the pinned CPU oracle on seeded
inputs.  Needs a real MI355X: run with  pytest -m gpu.

Tolerances (BASELINE.json north_star): 1e-5 relative on the log-likelihood, 1e-4 on the factor
matrices (measured relative to the largest entry of the matrix, the scale at which a probability
table is meaningful).  Kernel-level single steps are held to much tighter bounds.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden, golden_csr, coo_arrays, peak_rel, elem_rel, ROOT

pytestmark = pytest.mark.gpu

FUSED = 1
MODES = {"materialised": 0, "fused": FUSED}
KERNEL_CASES = ["kernels_k6", "kernels_k8_thresh", "kernels_k20", "kernels_k33"]
FIT_CASES = ["fit_k8_tol0", "fit_k5_earlystop", "fit_k4_weighted", "fit_k8_thresh",
             "fit_k6_tupleinit", "fit_k16_mid", "fit_k20_50it"]


def close_factors(a, b, tol=1e-4):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= tol, "factor mismatch %.3e (scale %.3e)" % (err, scale)


def close_ll(a, b, rtol=1e-5):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape
    fin = np.isfinite(b)
    np.testing.assert_array_equal(np.isfinite(a), fin)
    np.testing.assert_array_equal(a[~fin], b[~fin])
    np.testing.assert_allclose(a[fin], b[fin], rtol=rtol)


@pytest.fixture(scope="module")
def amd():
    import enstop_amd
    return enstop_amd


# ------------------------------------------------------------------------------------------------
# kernel level vs the reference's own outputs
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", KERNEL_CASES)
def test_e_step_vs_reference(amd, case):
    g = load_golden(case)
    r, c, v = coo_arrays(golden_csr(g))
    P = np.full_like(g["P"], -7.0)
    out = amd.plsa_e_step(r, c, v, g["V"].copy(), g["U"].copy(), P, g["thresh"])
    assert out is P
    np.testing.assert_array_equal(P == 0.0, g["P"] == 0.0)          # threshold pattern is exact
    np.testing.assert_allclose(P, g["P"], rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("case", KERNEL_CASES)
def test_m_steps_vs_reference(amd, case):
    g = load_golden(case)
    r, c, v = coo_arrays(golden_csr(g))
    n, k = g["U"].shape
    V, U = g["V"].copy(), g["U"].copy()
    a, b = np.zeros(k, np.float32), np.zeros(n, np.float32)
    amd.plsa_m_step(r, c, v, V, U, g["P"], a, b)
    np.testing.assert_allclose(V, g["V_m"], rtol=3e-6, atol=1e-9)
    np.testing.assert_allclose(U, g["U_m"], rtol=3e-6, atol=1e-9)
    np.testing.assert_allclose(a, g["norm_pwz"], rtol=3e-6)
    np.testing.assert_allclose(b, g["norm_pdz"], rtol=3e-6)

    V, U = g["V"].copy(), g["U"].copy()
    amd.plsa_m_step_w_sample_weight(r, c, v, V, U, g["P"], g["sw"], a, b)
    np.testing.assert_allclose(V, g["V_mw"], rtol=3e-6, atol=1e-9)
    np.testing.assert_allclose(U, g["U_mw"], rtol=3e-6, atol=1e-9)
    np.testing.assert_allclose(a, g["norm_pwz_w"], rtol=3e-6)
    np.testing.assert_allclose(b, g["norm_pdz_w"], rtol=3e-6)

    U = g["U"].copy()
    Vfixed = g["V"].copy()
    amd.plsa_refit_m_step(r, c, v, Vfixed, U, g["P"], np.ones(n, np.float32), b)
    np.testing.assert_allclose(U, g["U_refit"], rtol=3e-6, atol=1e-9)
    np.testing.assert_array_equal(Vfixed, g["V"])
    np.testing.assert_allclose(b, g["norm_pdz_refit"], rtol=3e-6)


@pytest.mark.parametrize("case", KERNEL_CASES)
def test_log_likelihood_vs_reference(amd, case):
    g = load_golden(case)
    r, c, v = coo_arrays(golden_csr(g))
    ones = np.ones(g["U"].shape[0], np.float32)
    for sw, key, VV, UU in ((ones, "ll_ones", g["V"], g["U"]), (g["sw"], "ll_sw", g["V"], g["U"]),
                            (ones, "ll_after_m", g["V_m"], g["U_m"])):
        got = amd.log_likelihood(r, c, v, VV, UU, sw)
        assert got.dtype == np.float32
        close_ll(np.array([got]), np.array([g[key]]), rtol=3e-6)


def test_unsorted_coo_input(amd):
    """Kernel-level functions accept COO triplets in any order, like the reference."""
    g = load_golden("kernels_k6")
    r, c, v = coo_arrays(golden_csr(g))
    perm = np.random.RandomState(0).permutation(r.shape[0])
    P = np.zeros_like(g["P"])
    amd.plsa_e_step(r[perm], c[perm], v[perm], g["V"], g["U"], P, g["thresh"])
    np.testing.assert_allclose(P, g["P"][perm], rtol=2e-6, atol=1e-9)


# ------------------------------------------------------------------------------------------------
# drivers vs the reference's own outputs
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("case", FIT_CASES)
def test_fit_vs_reference(amd, case, mode):
    g = load_golden(case)
    X = golden_csr(g)
    init = (g["U_init"], g["V_init"]) if "U_init" in g else "random"
    U, V, info = amd.plsa_fit(X, int(g["k"]), g["sw"], init=init, n_iter=int(g["n_iter"]),
                              n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
                              e_step_thresh=float(g["thresh"]), random_state=int(g["fit_seed"]),
                              flags=MODES[mode], return_info=True)
    assert U.dtype == np.float32 and V.dtype == np.float32
    assert info["n_iter"] == int(g["iters"])
    close_ll(info["log_likelihood_trace"], g["ll_trace"])
    close_factors(U, g["U"])
    close_factors(V, g["V"])
    # ... and ELEMENTWISE over every entry that is at least 1e-3 of the largest (a peak-relative bound alone says
    # nothing about the small entries).  With a large threshold (fit_k8_thresh, 2e-3) single responsibilities
    # flip on last-bit differences, between builds of the reference itself too: that case keeps the peak bound only
    if float(g["thresh"]) <= 1e-16:
        assert elem_rel(U, g["U"]) <= 1e-3, elem_rel(U, g["U"])
        assert elem_rel(V, g["V"]) <= 1e-3, elem_rel(V, g["V"])


@pytest.mark.parametrize("mode", ["materialised", "fused"])
@pytest.mark.parametrize("case", ["refit_k6", "refit_k8_weighted"])
def test_refit_vs_reference(amd, case, mode):
    g = load_golden(case)
    X = golden_csr(g)
    U, info = amd.plsa_refit(X, g["topics"], g["sw"], n_iter=int(g["n_iter"]),
                             n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
                             random_state=np.random.RandomState(42), flags=MODES[mode], return_info=True)
    assert info["n_iter"] == int(g["iters"])
    close_ll(info["log_likelihood_trace"], g["ll_trace"])
    close_factors(U, g["U"])


@pytest.mark.parametrize("case", ["estimator_int", "estimator_float", "estimator_int_emptyrows"])
def test_estimator_vs_reference(amd, case):
    g = load_golden(case)
    shape = tuple(int(s) for s in g["shape"])
    X = sp.csr_matrix((g["data"], g["indices"], g["indptr"]), shape=shape)
    model = amd.PLSA(n_components=int(g["k"]), n_iter=30, n_iter_per_test=10, tolerance=0.0, random_state=11)
    emb = model.fit_transform(X)
    assert str(np.asarray(emb).dtype) == str(g["embedding_dtype"])
    close_factors(emb, g["embedding"])
    close_factors(model.components_, g["components"])
    assert model.embedding_ is emb and model.training_data_.shape == shape
    if case.endswith("emptyrows"):
        assert np.all(emb[[0, 17, 47]] == 0.0)
    Xt = sp.csr_matrix((g["t_data"], g["t_indices"], g["t_indptr"]), shape=tuple(int(s) for s in g["t_shape"]))
    close_factors(model.transform(Xt), g["transformed"])
    assert model.fit(X) is model


def test_ensemble_member_vs_reference(amd):
    g = load_golden("member_k6")
    X = golden_csr(g)
    k = int(g["k"])
    kw = dict(n_iter=int(g["n_iter"]), n_iter_per_test=10, tolerance=0.0, e_step_thresh=float(g["thresh"]))
    close_factors(amd.plsa_topics(X, k, random_state=np.random.RandomState(5), **kw), g["V_rs5"])
    close_factors(amd.plsa_topics(X, k, random_state=9, **kw), g["V_int9"])
    close_factors(amd.plsa_topics(X, k, random_state=9, bootstrap=False, **kw), g["V_nobootstrap"])
    stack = amd.ensemble_of_topics(X, k, n_runs=3, parallelism="none",
                                   random_state=np.random.RandomState(21), **kw)
    assert stack.shape == g["V_stack_rs21"].shape
    close_factors(stack, g["V_stack_rs21"])


def test_bootstrap_gather_bit_exact(amd):
    """Device row gather == scipy's A[idx] (what enstop_.py:88 does), bit for bit."""
    rs = np.random.RandomState(7)
    X = sp.random(500, 300, density=0.05, format="csr", random_state=rs, dtype=np.float32)
    X.data = np.ceil(X.data * 9).astype(np.float32)
    idx = rs.randint(0, 500, size=500)
    with amd.Engine() as eng:
        eng.upload_csr(X)
        eng.bootstrap(idx)
        B = eng.download_active_csr()
        ref = X[idx]
        np.testing.assert_array_equal(B.indptr, ref.indptr)
        np.testing.assert_array_equal(B.indices, ref.indices)
        np.testing.assert_array_equal(B.data, ref.data)
        eng.bootstrap(None)
        B = eng.download_active_csr()
        np.testing.assert_array_equal(B.indptr, X.indptr)
        np.testing.assert_array_equal(B.indices, X.indices)
        with pytest.raises(amd.DeviceError):
            eng.bootstrap(np.array([0, 500], np.int64))


# ------------------------------------------------------------------------------------------------
# HIP vs the pinned oracle on seeded inputs the reference would take minutes for
# ------------------------------------------------------------------------------------------------
def _corpus(n, m, density, seed, empty_rows=0):
    rs = np.random.RandomState(seed)
    X = sp.random(n, m, density=density, format="csr", random_state=rs, dtype=np.float64)
    X.data = np.ceil(X.data * 7)
    if empty_rows:
        X = X.tolil()
        for r in rs.choice(n, empty_rows, replace=False):
            X[r] = 0
        X = X.tocsr()
    X.eliminate_zeros()
    X.sort_indices()
    return X.astype(np.float32)


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("k", [3, 20, 32, 64, 128, 200, 300])
def test_fit_vs_oracle(amd, oracle, k, mode):
    n, m = (600, 900) if k <= 128 else (150, 260)
    X = _corpus(n, m, 0.04, seed=k, empty_rows=3)
    rs = np.random.RandomState(k + 1)
    sw = (0.5 + rs.rand(n)).astype(np.float32) if k in (20, 200) else np.ones(n, np.float32)
    kw = dict(n_iter=12, n_iter_per_test=5, tolerance=0.0, e_step_thresh=1e-16, random_state=k)
    Uo, Vo, trace, iters = oracle.plsa_fit(X, k, sw, return_trace=True, **kw)
    U, V, info = amd.plsa_fit(X, k, sw, flags=MODES[mode], return_info=True, **kw)
    assert info["n_iter"] == iters
    close_ll(info["log_likelihood_trace"], trace)
    close_factors(U, Uo)
    close_factors(V, Vo)
    assert elem_rel(U, Uo) <= 1e-3 and elem_rel(V, Vo) <= 1e-3, (elem_rel(U, Uo), elem_rel(V, Vo))
    # rows of an empty document stay exactly zero (plsa.py:200-202 guard)
    empty = np.diff(X.indptr) == 0
    assert empty.sum() == 3 and np.all(U[empty] == 0.0)


@pytest.mark.parametrize("k", [20, 64])
def test_kernels_vs_oracle_midsize(amd, oracle, k):
    n, m = 3000, 4000
    X = _corpus(n, m, 0.01, seed=100 + k)
    r, c, v = coo_arrays(X)
    rs = np.random.RandomState(5)
    V = rs.rand(k, m); V /= V.sum(1, keepdims=True)
    U = rs.rand(n, k); U /= U.sum(1, keepdims=True)
    V = V.astype(np.float32); U = U.astype(np.float32)
    U[7] = 0.0                                             # a document with an all-zero row: norm == 0
    Po = oracle.plsa_e_step(r, c, v, V, U, np.zeros((r.shape[0], k), np.float32), 1e-7)
    P = amd.plsa_e_step(r, c, v, V, U, np.zeros((r.shape[0], k), np.float32), 1e-7)
    np.testing.assert_array_equal(P == 0.0, Po == 0.0)
    np.testing.assert_allclose(P, Po, rtol=3e-6, atol=1e-10)
    ones = np.ones(n, np.float32)
    Vo, Uo = V.copy(), U.copy()
    oracle.plsa_m_step(r, c, v, Vo, Uo, Po, np.zeros(k, np.float32), np.zeros(n, np.float32))
    Vh, Uh = V.copy(), U.copy()
    amd.plsa_m_step(r, c, v, Vh, Uh, Po, np.zeros(k, np.float32), np.zeros(n, np.float32))
    np.testing.assert_allclose(Uh, Uo, rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(Vh, Vo, rtol=2e-5, atol=1e-9)
    close_ll(np.array([amd.log_likelihood(r, c, v, Vo, Uo, ones)]),
             np.array([oracle.log_likelihood(r, c, v, Vo, Uo, ones)]))


@pytest.mark.parametrize("traversal", ["flat", "documents", "document_items", "document_items_auto"])
def test_e_step_traversals_agree_with_reference_and_oracle(amd, oracle, monkeypatch, traversal):
    """The E-step has two traversals, picked by corpus size (one group per non-zero; one group per
    document or document piece).  Both are forced here on the reference goldens and on seeded
    shapes with empty, single-entry and very long documents; the two must agree bit for bit.  Topic
    counts that leave lanes of a group idle (k = 10, 20, 23, 27: 3 / 5 / 6 / 7 chunks in 4 / 8 lanes) are among the shapes."""
    monkeypatch.setenv("PLSA_E_ROWS", "0" if traversal == "flat" else "1")
    monkeypatch.setenv("PLSA_E_SEG", {"flat": "0", "documents": "0", "document_items": "16",
                                      "document_items_auto": "-1"}[traversal])
    results = []
    with amd.Engine() as eng:
        for case in KERNEL_CASES:
            g = load_golden(case)
            eng.upload_csr(golden_csr(g))
            eng.set_factors(g["U"], g["V"])
            P = eng.e_step(g["thresh"])
            np.testing.assert_array_equal(P == 0.0, g["P"] == 0.0)
            np.testing.assert_allclose(P, g["P"], rtol=2e-6, atol=1e-9)
        rs = np.random.RandomState(11)
        for n, m, k, thresh in ((700, 900, 64, 1e-32), (300, 500, 20, 1e-6), (1200, 300, 33, 1e-32),
                                (50, 4000, 128, 1e-32), (2000, 100, 3, 1e-4), (400, 600, 200, 1e-32),
                                (500, 700, 10, 1e-32), (350, 450, 23, 1e-5), (260, 640, 27, 1e-32), (333, 777, 20, 0.0)):
            X = _corpus(n, m, 0.03, seed=n + k, empty_rows=min(5, n // 10)).tolil()
            X[1, :] = 1.0                                  # a document holding the whole vocabulary
            X[2, :] = 0.0; X[2, m // 2] = 3.0              # a single-entry document
            X = X.tocsr().astype(np.float32); X.sort_indices()
            V = rs.rand(k, m); V /= V.sum(1, keepdims=True)
            U = rs.rand(n, k); U /= U.sum(1, keepdims=True)
            V = V.astype(np.float32); U = U.astype(np.float32)
            U[7] = 0.0
            r, c, v = coo_arrays(X)
            Po = oracle.plsa_e_step(r, c, v, V, U, np.zeros((r.shape[0], k), np.float32), thresh)
            eng.upload_csr(X)
            eng.set_factors(U, V)
            P = eng.e_step(thresh)
            np.testing.assert_array_equal(P == 0.0, Po == 0.0)
            np.testing.assert_allclose(P, Po, rtol=3e-6, atol=1e-10)
            results.append(P)
    _E_STEP_TRAVERSAL_RESULTS[traversal] = results
    first = next(iter(_E_STEP_TRAVERSAL_RESULTS.values()))
    for a, b in zip(first, results):
        np.testing.assert_array_equal(a, b)               # same arithmetic per non-zero in every traversal


_E_STEP_TRAVERSAL_RESULTS = {}


# ------------------------------------------------------------------------------------------------
# size-independent properties at sizes the oracle cannot reach in seconds
# ------------------------------------------------------------------------------------------------
def test_properties_large_synthetic(amd):
    n, m, nnz_t, k = 200_000, 50_000, 20_000_000, 64
    with amd.Engine() as eng:
        nnz = eng.generate_synthetic(n, m, nnz_t, seed=3)
        assert abs(nnz - nnz_t) / nnz_t < 0.01
        rs = np.random.RandomState(0)
        V = rs.rand(k, m); V /= V.sum(1, keepdims=True)
        U = rs.rand(n, k); U /= U.sum(1, keepdims=True)
        results = {}
        for name, flags in MODES.items():
            eng.set_factors(U.astype(np.float32), V.astype(np.float32))
            iters, ll = eng.fit(None, n_iter=6, n_iter_per_test=1, tolerance=0.0, e_step_thresh=1e-32,
                                flags=flags, trace=True)
            assert iters == 6 and ll.shape == (7,)
            assert np.all(np.diff(ll.astype(np.float64)) >= -1e-6 * np.abs(ll[0])), "EM must not decrease LL"
            Uf, Vf = eng.get_factors()
            np.testing.assert_allclose(Uf.sum(1, dtype=np.float64), 1.0, atol=2e-5)
            np.testing.assert_allclose(Vf.sum(1, dtype=np.float64), 1.0, atol=2e-4)
            assert Uf.min() >= 0.0 and Vf.min() >= 0.0
            results[name] = (Uf, Vf, ll)
        base = results["materialised"]
        for name, (Uf, Vf, ll) in results.items():
            close_ll(ll, base[2])
            close_factors(Uf, base[0])
            close_factors(Vf, base[1])
        # no float atomics anywhere: both schedules are bit-reproducible run to run
        for name, flags in MODES.items():
            eng.set_factors(U.astype(np.float32), V.astype(np.float32))
            eng.fit(None, n_iter=6, n_iter_per_test=1, tolerance=0.0, flags=flags, trace=True)
            U2, V2 = eng.get_factors()
            np.testing.assert_array_equal(U2, results[name][0])
            np.testing.assert_array_equal(V2, results[name][1])


def test_synthetic_generator_is_canonical_and_deterministic(amd):
    with amd.Engine() as eng:
        nnz = eng.generate_synthetic(5000, 2000, 300_000, seed=11)
        A = eng.download_active_csr()
        assert A.nnz == nnz and abs(nnz - 300_000) / 300_000 < 0.01
        assert np.all(np.diff(A.indptr) >= 1)                       # no empty documents
        assert A.has_sorted_indices and A.indices.max() < 2000 and A.indices.min() >= 0
        B = A.copy(); B.sum_duplicates()
        assert B.nnz == A.nnz                                        # no duplicate (doc, word)
        assert A.data.min() >= 1.0 and np.all(A.data == np.round(A.data))
        nnz2 = eng.generate_synthetic(5000, 2000, 300_000, seed=11)
        A2 = eng.download_active_csr()
        assert nnz2 == nnz
        np.testing.assert_array_equal(A.indptr, A2.indptr)
        np.testing.assert_array_equal(A.indices, A2.indices)
        np.testing.assert_array_equal(A.data, A2.data)
        eng.generate_synthetic(5000, 2000, 300_000, seed=12)
        A3 = eng.download_active_csr()
        assert not np.array_equal(A.indices[:1000], A3.indices[:1000])
        # Zipf head: the most frequent word appears in far more documents than the median word
        df = np.bincount(A.indices, minlength=2000)
        assert df.max() > 20 * max(np.median(df), 1)


# ------------------------------------------------------------------------------------------------
# edge cases: degenerate shapes, duplicate entries, limits, API-compatible class names
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape_k", [((1, 9), 3), ((9, 1), 2), ((7, 5), 1), ((6, 4), 9), ((3, 3), 4)])
def test_degenerate_shapes_vs_oracle(amd, oracle, shape_k):
    (n, m), k = shape_k
    rs = np.random.RandomState(n * 31 + m)
    D = np.ceil(rs.rand(n, m) * 4) * (rs.rand(n, m) < 0.7)
    D[0, 0] = 2.0
    X = sp.csr_matrix(D.astype(np.float32))
    sw = np.ones(n, np.float32)
    kw = dict(n_iter=7, n_iter_per_test=3, tolerance=0.0, e_step_thresh=1e-32, random_state=5)
    Uo, Vo, trace, iters = oracle.plsa_fit(X, k, sw, return_trace=True, **kw)
    for mode in MODES.values():
        U, V, info = amd.plsa_fit(X, k, sw, flags=mode, return_info=True, **kw)
        got = info["log_likelihood_trace"]
        if n == 1 or m == 1 or k == 1:
            # a rank-one problem converges in one step; afterwards the reference's `change == 0`
            # stop arm (plsa.py:635) fires on last-bit noise of whichever float32 sum is used, so
            # the iteration count is not a well-defined quantity -- the fixed point is
            assert 1 <= info["n_iter"] <= 7
            q = min(len(got), len(trace))
            close_ll(got[:q], trace[:q])
        else:
            assert info["n_iter"] == iters
            close_ll(got, trace)
        close_factors(U, Uo); close_factors(V, Vo)


def test_duplicate_coo_entries_are_separate_nonzeros(amd, oracle):
    """X.tocoo() of a CSR with repeated (d, w) keeps both entries; the reference treats them as two
    non-zeros (it never sums duplicates).  Same here, on both the CSR and the CSC side."""
    rows = np.array([0, 0, 0, 1, 1, 2, 2, 2, 2], np.int32)
    cols = np.array([1, 1, 3, 0, 3, 2, 2, 2, 0], np.int32)
    vals = np.array([1, 2, 1, 3, 1, 1, 1, 2, 4], np.float32)
    n, m, k = 3, 4, 3
    rs = np.random.RandomState(1)
    V = rs.rand(k, m).astype(np.float32); V /= V.sum(1, keepdims=True)
    U = rs.rand(n, k).astype(np.float32); U /= U.sum(1, keepdims=True)
    Po = oracle.plsa_e_step(rows, cols, vals, V, U, np.zeros((9, k), np.float32), 1e-32)
    P = amd.plsa_e_step(rows, cols, vals, V, U, np.zeros((9, k), np.float32), 1e-32)
    np.testing.assert_allclose(P, Po, rtol=3e-6)
    Vo, Uo = V.copy(), U.copy()
    oracle.plsa_m_step(rows, cols, vals, Vo, Uo, Po, np.zeros(k, np.float32), np.zeros(n, np.float32))
    Vh, Uh = V.copy(), U.copy()
    amd.plsa_m_step(rows, cols, vals, Vh, Uh, Po, np.zeros(k, np.float32), np.zeros(n, np.float32))
    np.testing.assert_allclose(Vh, Vo, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(Uh, Uo, rtol=1e-5, atol=1e-9)


def test_stored_zeros_and_unsorted_rows(amd, oracle):
    """Input formats either side of the path.  `X.tocoo()` (plsa.py:714) hands the reference whatever the CSR stores, in
    the stored order: explicitly stored zero counts stay entries (their responsibilities are computed, they add exactly
    nothing to either factor or to the likelihood), and rows whose column indices are not sorted are walked as stored.
    Both schedules and the estimator follow the oracle on such a matrix; a CSC or dense copy of the same data gives
    the result of its own CSR conversion (sorted rows: same factors to rounding)."""
    rs = np.random.RandomState(12)
    X = sp.random(260, 180, density=0.08, format="csr", random_state=rs, dtype=np.float32)
    X.data = np.ceil(X.data * 4).astype(np.float32)
    X = X[np.diff(X.indptr) > 0].tocsr()
    X.data[rs.rand(X.nnz) < 0.15] = 0.0                     # stored zeros (kept: no eliminate_zeros())
    X.data[X.indptr[:-1]] = np.maximum(X.data[X.indptr[:-1]], 1.0)   # ... but no document of zeros only (the estimator drops those)
    for d in range(0, X.shape[0], 3):                       # every third row in a shuffled stored order
        a, b = X.indptr[d], X.indptr[d + 1]
        p = rs.permutation(b - a)
        X.indices[a:b] = X.indices[a:b][p]; X.data[a:b] = X.data[a:b][p]
    X.has_sorted_indices = False
    assert (X.data == 0).sum() > 100 and X.nnz == X.tocoo().nnz
    n, m = X.shape
    k = 9
    ones = np.ones(n, np.float32)
    kw = dict(n_iter=12, n_iter_per_test=4, tolerance=0.0, e_step_thresh=1e-32, random_state=5)
    Uo, Vo, trace, iters = oracle.plsa_fit(X, k, ones, return_trace=True, **kw)
    for mode in MODES.values():
        U, V, info = amd.plsa_fit(X, k, ones, flags=mode, return_info=True, **kw)
        assert info["n_iter"] == iters == 12
        close_factors(U, Uo); close_factors(V, Vo)
        close_ll(info["log_likelihood_trace"], trace)
    Xs = X.copy(); Xs.sort_indices()
    for other in (Xs.tocsc(), Xs.toarray()):                # formats the estimator accepts (check_array, plsa.py:1138)
        model = amd.PLSA(n_components=k, n_iter=12, n_iter_per_test=4, tolerance=0.0, random_state=5)
        emb = model.fit_transform(other.astype(np.int64) if not sp.issparse(other) else other.astype(np.int64))
        close_factors(model.components_, Vo, tol=2e-4); close_factors(emb, Uo, tol=2e-4)


@pytest.mark.parametrize("k", [384, 500, 512, 1000, 1024])
def test_largest_topic_counts_vs_oracle(amd, oracle, k):
    """The widest lane shapes (64 lanes x 2 / 4 chunks; k = 512 and 1024 are FULL shapes, 500 and 1000 are not): both
    schedules against the oracle, weighted, with a log-likelihood test every other iteration."""
    rs = np.random.RandomState(k)
    X = sp.random(90, 70, density=0.12, format="csr", random_state=rs, dtype=np.float32)
    X.data = np.ceil(X.data * 5).astype(np.float32)
    X = X[np.diff(X.indptr) > 0]
    n = X.shape[0]
    sw = (0.5 + rs.rand(n)).astype(np.float32)
    kw = dict(n_iter=5, n_iter_per_test=2, tolerance=0.0, e_step_thresh=1e-32, random_state=4)
    Uo, Vo, trace, iters = oracle.plsa_fit(X, k, sw, return_trace=True, **kw)
    for mode in MODES.values():
        U, V, info = amd.plsa_fit(X, k, sw, flags=mode, return_info=True, **kw)
        assert info["n_iter"] == iters == 5
        close_ll(info["log_likelihood_trace"][:len(trace)], trace, rtol=2e-5)
        close_factors(U, Uo); close_factors(V, Vo)


@pytest.mark.parametrize("mode", list(MODES))
def test_loop_parameter_corners_vs_oracle(amd, oracle, mode):
    """Corners of plsa_fit_inner's loop (plsa.py:591-640) the estimator's own validation lets through: no iteration at all,
    one, two; a test interval longer than the run (the only test is the one after iteration 0, `0 % n == 0`); a tolerance
    that stops at the first test; and a threshold ABOVE every product -- every responsibility row stays zero
    (plsa.py:103-105), the M-step then leaves both factors all-zero (norms are 0: plsa.py:196-202) and the likelihood is
    log(0) = -inf from there on.  Iteration counts, traces (inf included) and factors as the oracle has them."""
    X = _corpus(150, 120, 0.06, seed=41)
    ones = np.ones(150, np.float32)
    for kw in (dict(n_iter=0, n_iter_per_test=10, tolerance=0.001), dict(n_iter=1, n_iter_per_test=10, tolerance=0.0),
               dict(n_iter=2, n_iter_per_test=1, tolerance=0.0), dict(n_iter=7, n_iter_per_test=50, tolerance=0.0),
               dict(n_iter=30, n_iter_per_test=3, tolerance=np.inf), dict(n_iter=30, n_iter_per_test=3, tolerance=0.5),
               dict(n_iter=6, n_iter_per_test=2, tolerance=0.0, e_step_thresh=1.0)):
        kw = dict(dict(e_step_thresh=1e-32, random_state=8), **kw)
        Uo, Vo, trace, iters = oracle.plsa_fit(X, 6, ones, return_trace=True, **kw)
        U, V, info = amd.plsa_fit(X, 6, ones, flags=MODES[mode], return_info=True, **kw)
        assert info["n_iter"] == iters, (kw, info["n_iter"], iters)
        got = info["log_likelihood_trace"]
        q = min(len(got), len(trace))           # the trace may carry the result-neutral test of the last iteration
        assert q >= len(trace) - 1 and q >= 1, (kw, len(got), len(trace))
        close_ll(got[:q], trace[:q])
        close_factors(U, Uo); close_factors(V, Vo)
        if kw["e_step_thresh"] == 1.0:
            assert not U.any() and not V.any() and np.isneginf(trace[-1])


def test_limits_and_errors(amd):
    X = sp.random(50, 40, density=0.2, format="csr", random_state=0, dtype=np.float32)
    ones = np.ones(50, np.float32)
    U, V = amd.plsa_fit(X, 1024, ones, n_iter=2, random_state=0)           # largest supported k
    assert U.shape == (50, 1024) and np.allclose(U.sum(1), 1.0, atol=1e-4)
    with pytest.raises(amd.DeviceError, match="outside"):
        amd.plsa_fit(X, 1025, ones, n_iter=1, random_state=0)
    with pytest.raises(ValueError, match="Unrecognized init"):
        amd.plsa_fit(X, 4, ones, init="nope", n_iter=1)
    with amd.Engine() as eng:
        with pytest.raises(amd.DeviceError, match="upload a corpus"):
            eng._ok(eng._L.plsa_set_factors(eng._h, np.ones((5, 2), np.float32).ctypes.data,
                                            np.ones((2, 3), np.float32).ctypes.data, 5, 3, 2))
        eng.upload_csr(X)
        with pytest.raises(amd.DeviceError, match="do not match"):
            eng._ok(eng._L.plsa_set_factors(eng._h, np.ones((7, 2), np.float32).ctypes.data,
                                            np.ones((2, 40), np.float32).ctypes.data, 7, 40, 2))
        with pytest.raises(amd.DeviceError, match="factors not set"):
            eng.e_step()
        eng.set_factors(np.full((50, 2), 0.5, np.float32), np.full((2, 40), 1 / 40, np.float32))
        with pytest.raises(amd.DeviceError, match="no P"):
            eng.m_step()


def test_api_compatible_class_names(amd):
    g = load_golden("estimator_int")
    shape = tuple(int(s) for s in g["shape"])
    X = sp.csr_matrix((g["data"], g["indices"], g["indptr"]), shape=shape)
    kw = dict(n_components=int(g["k"]), n_iter=30, n_iter_per_test=10, tolerance=0.0, random_state=11)
    for cls, extra in ((amd.StreamedPLSA, dict(block_size=1024)), (amd.BlockParallelPLSA, dict(n_row_blocks=4, n_col_blocks=2)),
                       (amd.GPUPLSA, dict(n_row_blocks=2, n_col_blocks=2)), (amd.DistributedPLSA, dict(n_row_blocks=8, n_col_blocks=8))):
        model = cls(**kw, **extra).fit(X)
        close_factors(model.embedding_, g["embedding"])
        close_factors(model.components_, g["components"])
        assert set(extra) <= set(model.get_params())


def test_init_nndsvd_and_nmf_run(amd):
    X = _corpus(200, 300, 0.05, seed=9)
    for init in ("nndsvd", "nmf"):
        U, V = amd.plsa_fit(X, 5, np.ones(200, np.float32), init=init, n_iter=5, random_state=0)
        assert np.all(np.isfinite(U)) and np.all(np.isfinite(V))
        np.testing.assert_allclose(V.sum(1), 1.0, atol=1e-4)


def test_device_random_init(amd):
    """init="device_random": additive throughput option (counter-based RNG on the GPU)."""
    X = _corpus(400, 700, 0.04, seed=3)
    ones = np.ones(400, np.float32)
    kw = dict(init="device_random", n_iter=8, n_iter_per_test=2, tolerance=0.0, return_info=True)
    U1, V1, i1 = amd.plsa_fit(X, 20, ones, random_state=4, **kw)
    U2, V2, i2 = amd.plsa_fit(X, 20, ones, random_state=4, **kw)
    U3, V3, _ = amd.plsa_fit(X, 20, ones, random_state=5, **kw)
    np.testing.assert_array_equal(U1, U2); np.testing.assert_array_equal(V1, V2)
    assert not np.array_equal(V1, V3)
    np.testing.assert_allclose(U1.sum(1), 1.0, atol=1e-5); np.testing.assert_allclose(V1.sum(1), 1.0, atol=1e-4)
    ll = i1["log_likelihood_trace"].astype(np.float64)
    assert np.all(np.diff(ll) >= -1e-6 * abs(ll[0]))
    with amd.Engine() as eng:
        eng.upload_csr(X)
        eng.init_factors_device(20, 9)
        U0, V0 = eng.get_factors()
        assert U0.min() > 0 and V0.min() > 0
        np.testing.assert_allclose(U0.sum(1), 1.0, atol=1e-5); np.testing.assert_allclose(V0.sum(1), 1.0, atol=1e-4)
        assert abs(U0.mean() - 1 / 20) < 1e-3 and U0.std() > 0.01


@pytest.mark.parametrize("combination", ["hellinger", "kl_divergence"])
def test_ensemble_topics_end_to_end(amd, combination):
    """EnsembleTopics on planted topics: GPU ensemble members -> divergence matrix and cluster
    representatives on the GPU, tree step on the host -> GPU refit."""
    rs = np.random.RandomState(0)
    n, m, k_true = 600, 300, 4
    topics = rs.dirichlet(np.full(m, 0.03), size=k_true)
    mix = rs.dirichlet(np.full(k_true, 0.2), size=n)
    X = sp.csr_matrix(rs.poisson(80 * (mix @ topics)).astype(np.float32))
    X = X[np.asarray(X.sum(1)).ravel() > 0]
    model = amd.EnsembleTopics(n_components=k_true, n_starts=8, min_samples=2, min_cluster_size=3,
                               topic_combination=combination, parallelism="none", n_iter=40,
                               random_state=np.random.RandomState(3))
    emb = model.fit_transform(X)
    assert model.components_.shape[1] == m and model.n_components_ == model.components_.shape[0] >= 2
    assert emb.shape == (X.shape[0], model.n_components_)
    np.testing.assert_allclose(model.components_.sum(1), 1.0, atol=1e-4)
    np.testing.assert_allclose(emb.sum(1), 1.0, atol=1e-4)
    from enstop_amd.ensemble import all_pairs_hellinger_distance
    D = all_pairs_hellinger_distance(np.vstack([topics, model.components_]))[:k_true, k_true:]
    assert (D.min(axis=1) < 0.35).sum() >= 3          # most planted topics have a close stable topic
    tr = model.transform(X[:50])
    assert tr.shape == (50, model.n_components_)


# ------------------------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties (the oracle cannot run these in seconds)
# ------------------------------------------------------------------------------------------------
def _host_init(n, m, k, seed):
    from enstop_amd.plsa import plsa_init

    class S:
        shape = (n, m)
    U, V = plsa_init(S, k, rng=np.random.RandomState(seed))
    return U.astype(np.float32), V.astype(np.float32)


@pytest.mark.parametrize("cfg", [(100_000, 50_000, 10_000_000, 32), (1_000_000, 100_000, 100_000_000, 64)],
                         ids=["config2", "config3"])
def test_full_size_properties(amd, cfg):
    n, m, nnz_t, k = cfg
    with amd.Engine() as eng:
        nnz = eng.generate_synthetic(n, m, nnz_t, seed=0)
        assert abs(nnz - nnz_t) / nnz_t < 0.006
        U0, V0 = _host_init(n, m, k, 42)
        out = {}
        for name, flags in MODES.items():
            eng.set_factors(U0, V0)
            iters, ll = eng.fit(None, n_iter=5, n_iter_per_test=1, tolerance=0.0, e_step_thresh=1e-32,
                                flags=flags, trace=True)
            assert iters == 5 and ll.shape == (6,)
            ll64 = ll.astype(np.float64)
            assert np.all(np.diff(ll64) > 0), "EM must increase the log-likelihood from a random start"
            U, V = eng.get_factors()
            assert U.min() >= 0 and V.min() >= 0
            np.testing.assert_allclose(U.sum(1, dtype=np.float64), 1.0, atol=3e-5)
            np.testing.assert_allclose(V.sum(1, dtype=np.float64), 1.0, atol=3e-4)
            out[name] = (U, V, ll)
        # the two schedules are the same algorithm
        close_ll(out["fused"][2], out["materialised"][2])
        close_factors(out["fused"][0], out["materialised"][0])
        close_factors(out["fused"][1], out["materialised"][1])
        # idempotent re-run: no atomics anywhere -> bit-identical
        eng.set_factors(U0, V0)
        eng.fit(None, n_iter=5, n_iter_per_test=1, tolerance=0.0, flags=FUSED, trace=True)
        U2, V2 = eng.get_factors()
        np.testing.assert_array_equal(U2, out["fused"][0]); np.testing.assert_array_equal(V2, out["fused"][1])
        # kernel-level consistency at full size: M-step(E-step(.)) == one fused iteration
        eng.set_factors(U0, V0)
        eng.e_step(1e-32, want_host_copy=False)
        eng.m_step()
        Um, Vm = eng.get_factors()
        eng.set_factors(U0, V0)
        eng.fit(None, n_iter=1, n_iter_per_test=10, tolerance=0.0, flags=FUSED)
        Uf, Vf = eng.get_factors()
        close_factors(Uf, Um, tol=2e-5); close_factors(Vf, Vm, tol=2e-5)


def test_topical_generator_is_deterministic_and_has_cooccurrence_structure(amd):
    """plsa_generate_synthetic_topics (round 5): same arguments -> the same matrix, bit for bit; a different seed, topic
    count or concentration -> another one; and, unlike the independent-token corpus, documents share words in CLUSTERS:
    the word overlap of random document pairs is bimodal (same dominant topic: large, otherwise only the shared
    ranking's head words), so its coefficient of variation is about twice that of the independent corpus."""
    n, m, target = 4000, 3000, 200_000
    kw = dict(topics=16, alpha=0.05, background=0.1)
    with amd.Engine() as eng:
        nnz = eng.generate_synthetic(n, m, target, seed=11, **kw)
        A = eng.download_active_csr()
        assert abs(nnz - target) / target < 0.02 and A.nnz == nnz and A.shape == (n, m)
        assert A.has_sorted_indices and A.data.min() >= 1 and np.diff(A.indptr).min() >= 1
        assert eng.generate_synthetic(n, m, target, seed=11, **kw) == nnz
        B = eng.download_active_csr()
        assert (A != B).nnz == 0 and np.array_equal(A.data, B.data)
        for other in (dict(kw, topics=17), dict(kw, alpha=0.5), dict(kw, background=0.5)):
            eng.generate_synthetic(n, m, target, seed=11, **other)
            assert (eng.download_active_csr() != A).nnz > 0
        eng.generate_synthetic(n, m, target, seed=12, **kw)
        assert (eng.download_active_csr() != A).nnz > 0
        eng.generate_synthetic(n, m, target, seed=11)
        I = eng.download_active_csr()
        with pytest.raises(amd.DeviceError):
            eng.generate_synthetic(n, m, target, seed=11, topics=300)
        # edge mixtures: one topic; a concentration so small that every gamma draw underflows (the document falls back to
        # one hashed topic); background = 1 (every token from the shared ranking); the labels follow the mixtures
        for edge in (dict(topics=1, alpha=0.1, background=0.0), dict(topics=40, alpha=1e-4, background=0.0),
                     dict(topics=5, alpha=0.1, background=1.0)):
            nz = eng.generate_synthetic(500, 400, 20_000, seed=2, **edge)
            E = eng.download_active_csr()
            assert E.nnz == nz and np.all(np.isfinite(E.data)) and E.indices.max() < 400 and np.diff(E.indptr).min() >= 1
            lab = eng.synthetic_dominant_topics()
            assert lab.shape == (500,) and lab.min() >= 0 and lab.max() < edge["topics"]
        eng.generate_synthetic(500, 400, 20_000, seed=2)
        with pytest.raises(amd.DeviceError, match="topical"):
            eng.synthetic_dominant_topics()

    def overlap_cv(X):
        Xb = (X[:600] > 0).astype(np.float64)
        S = np.asarray((Xb @ Xb.T).todense())
        off = S[~np.eye(600, dtype=bool)]
        return off.std() / off.mean()
    cv_t, cv_i = overlap_cv(A), overlap_cv(I)
    assert cv_t > 1.8 * cv_i, (cv_t, cv_i)             # measured 1.06 against 0.47


def test_count_scaling_invariance(amd):
    """X -> 2X leaves every EM iterate bit-identical (all norms scale by an exact power of two)
    and doubles the log-likelihood."""
    with amd.Engine() as eng:
        eng.generate_synthetic(100_000, 50_000, 10_000_000, seed=1)
        A = eng.download_active_csr()
        U0, V0 = _host_init(100_000, 50_000, 32, 7)
        res = []
        for scale in (1.0, 2.0):
            B = A.copy(); B.data = B.data * np.float32(scale)
            eng.upload_csr(B)
            eng.set_factors(U0, V0)
            _, ll = eng.fit(None, n_iter=4, n_iter_per_test=1, tolerance=0.0, flags=FUSED, trace=True)
            res.append(eng.get_factors() + (ll,))
        np.testing.assert_array_equal(res[0][0], res[1][0])
        np.testing.assert_array_equal(res[0][1], res[1][1])
        np.testing.assert_allclose(res[1][2], 2.0 * res[0][2], rtol=1e-6)


def test_document_permutation_equivariance(amd):
    """Permuting the documents permutes P(z|d) and leaves P(w|z) unchanged (up to summation order)."""
    with amd.Engine() as eng:
        n, m, k = 100_000, 50_000, 32
        eng.generate_synthetic(n, m, 10_000_000, seed=2)
        A = eng.download_active_csr()
        U0, V0 = _host_init(n, m, k, 9)
        eng.set_factors(U0, V0)
        eng.fit(None, n_iter=4, n_iter_per_test=10, tolerance=0.0, flags=FUSED)
        U, V = eng.get_factors()
        perm = np.random.RandomState(0).permutation(n)
        eng.bootstrap(perm)                         # a permutation is a bootstrap sample without repeats
        eng.set_factors(U0[perm], V0)
        eng.fit(None, n_iter=4, n_iter_per_test=10, tolerance=0.0, flags=FUSED)
        Up, Vp = eng.get_factors()
        close_factors(Up, U[perm], tol=2e-5)
        close_factors(Vp, V, tol=2e-5)
        # and the device gather itself is exact at this size
        B = eng.download_active_csr()
        ref = A[perm]
        np.testing.assert_array_equal(B.indptr, ref.indptr)
        np.testing.assert_array_equal(B.indices, ref.indices)
        np.testing.assert_array_equal(B.data, ref.data)


# ------------------------------------------------------------------------------------------------
# doc-sharded single fit (accumulate / all-reduce / finish)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shards", [2, 3])
def test_sharded_fit_matches_reference_golden(amd, shards):
    """N row shards on one device (all-reduce emulated through the host) == the reference's fit."""
    for case in ("fit_k5_earlystop", "fit_k4_weighted", "fit_k16_mid"):
        g = load_golden(case)
        X = golden_csr(g)
        U, V, info = amd.sharded_plsa_fit(X, int(g["k"]), g["sw"], n_iter=int(g["n_iter"]),
                                          n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
                                          e_step_thresh=float(g["thresh"]), random_state=int(g["fit_seed"]),
                                          local_shards=shards, return_info=True)
        assert info["n_iter"] == int(g["iters"])
        close_ll(info["log_likelihood_trace"], g["ll_trace"])
        close_factors(U, g["U"]); close_factors(V, g["V"])


def test_materialised_schedule_tiled_by_doc_blocks(amd, oracle):
    """block_parallel_plsa.py:373-403 on the device: the MATERIALISED schedule (E-step into P(z|w,d), M-step from it) run doc
    block by doc block, every block's responsibilities alive only inside its own step, all blocks writing through ONE
    borrowed P(z|w,d) buffer of the largest block's size.  Same factors, iteration count and likelihood trace as the
    untiled materialised fit and as the oracle; a borrowed buffer that is too small is refused by name."""
    from enstop_amd.sharded import sharded_plsa_fit
    X = _corpus(900, 700, 0.04, seed=23, empty_rows=4)
    n = X.shape[0]
    k = 12
    rs = np.random.RandomState(2)
    sw = (0.5 + rs.rand(n)).astype(np.float32)
    kw = dict(n_iter=9, n_iter_per_test=3, tolerance=0.0, e_step_thresh=1e-32, random_state=6)
    for weights in (None, sw):
        w = np.ones(n, np.float32) if weights is None else weights
        Uo, Vo, trace, iters = oracle.plsa_fit(X, k, w, return_trace=True, **kw)
        U1, V1 = amd.plsa_fit(X, k, w, flags=0, **kw)
        for shards in (2, 5):
            U, V, info = sharded_plsa_fit(X, k, sample_weight=weights, local_shards=shards, flags=0, return_info=True, **kw)
            assert info["n_iter"] == iters == 9
            close_factors(U, Uo); close_factors(V, Vo)
            close_factors(U, U1, tol=1e-5); close_factors(V, V1, tol=1e-5)
            q = min(len(info["log_likelihood_trace"]), len(trace))
            close_ll(info["log_likelihood_trace"][:q], trace[:q])
    with amd.Engine() as a, amd.Engine() as b:
        a.upload_csr(X[:300]); b.upload_csr(X[300:])
        V0 = np.full((k, X.shape[1]), 1.0 / X.shape[1], np.float32)
        a.set_factors(np.full((300, k), 1.0 / k, np.float32), V0)
        b.set_factors(np.full((n - 300, k), 1.0 / k, np.float32), V0)
        ptr = a.p_reserve(a.p_bytes())                          # sized for the SMALLER block
        assert b.p_bytes() > a.p_bytes()
        b.p_borrow(ptr, a.p_bytes())
        with pytest.raises(amd.DeviceError, match="borrowed"):
            b.em_accumulate(None, 1e-32, materialised=True)
        b.p_borrow(None)
        b.em_accumulate(None, 1e-32, materialised=True)         # own allocation again
        b.em_finish()


def test_sharded_fit_with_more_shards_than_documents_raises(amd):
    """No rank may end up without rows (it would leave the others waiting in the all-reduce): the same
    ValueError on every rank, before any exchange."""
    X = sp.csr_matrix(np.array([[1, 0, 2], [0, 3, 0]], np.float32))
    with pytest.raises(ValueError):
        amd.sharded_plsa_fit(X, 2, None, n_iter=2, random_state=0, local_shards=3)
    U, V = amd.sharded_plsa_fit(X, 2, None, n_iter=2, random_state=0, local_shards=2)
    assert U.shape == (2, 2) and V.shape == (2, 3)


def test_sharded_fit_matches_single_gpu_fit_midsize(amd):
    X = _corpus(5000, 3000, 0.02, seed=21, empty_rows=4)
    sw = np.ones(5000, np.float32)
    kw = dict(n_iter=12, n_iter_per_test=5, tolerance=0.0, e_step_thresh=1e-16, random_state=3)
    U1, V1, i1 = amd.plsa_fit(X, 64, sw, return_info=True, **kw)
    U4, V4, i4 = amd.sharded_plsa_fit(X, 64, sw, local_shards=4, return_info=True, **kw)
    assert i1["n_iter"] == i4["n_iter"] == 12
    close_ll(i4["log_likelihood_trace"], i1["log_likelihood_trace"])
    close_factors(U4, U1, tol=2e-5); close_factors(V4, V1, tol=2e-5)


def test_resident_sample_weights_and_comm_diagnostics(amd):
    """Round 4 ABI additions.  plsa_set_sample_weight: weights uploaded once apply to every later call that passes
    none (the per-iteration calls of the split doc-sharded loop no longer copy + wait) -- same bits as passing them;
    cleared with None; a matrix with another document count makes the next use an error.  plsa_comm_last_error returns
    RCCL's text (empty without a failure).  An RcclComm asked to gather from ANOTHER engine falls back to the host path."""
    import ctypes as C
    from enstop_amd import comm
    from enstop_amd.engine import Engine, DeviceError
    rs = np.random.RandomState(5)
    X = sp.random(800, 600, density=0.03, format="csr", random_state=rs, dtype=np.float32)
    X.data = np.ceil(X.data * 4).astype(np.float32)
    sw = (0.25 + 1.5 * rs.rand(800)).astype(np.float32)
    U0, V0 = amd.plsa_init(X, 12, rng=np.random.RandomState(1))
    with Engine() as eng:
        eng.upload_csr(X)
        eng.set_factors(U0.astype(np.float32), V0.astype(np.float32))
        it_a, tr_a = eng.fit(sw, 9, 4, 0.0, 1e-32, amd.PLSA_FUSED, trace=True)
        Ua, Va = eng.get_factors()
        eng.set_factors(U0.astype(np.float32), V0.astype(np.float32))
        eng.set_sample_weight(sw)
        it_b, tr_b = eng.fit(None, 9, 4, 0.0, 1e-32, amd.PLSA_FUSED, trace=True)      # resident weights apply
        Ub, Vb = eng.get_factors()
        np.testing.assert_array_equal(Ub, Ua); np.testing.assert_array_equal(Vb, Va)
        np.testing.assert_array_equal(tr_b, tr_a)
        ll_w = eng.log_likelihood(None)
        eng.set_sample_weight(None)
        ll_1 = eng.log_likelihood(None)
        assert ll_w != ll_1 and abs(ll_w - eng.log_likelihood(sw)) <= 1e-9 * abs(ll_w)
        eng.set_sample_weight(sw)
        eng.upload_csr(X[:500])
        eng.set_factors(U0[:500].astype(np.float32), V0.astype(np.float32))
        with pytest.raises(DeviceError, match="resident sample weights"):
            eng.log_likelihood(None)
        eng.set_sample_weight(None)
        buf = C.create_string_buffer(256)
        assert eng._L.plsa_comm_last_error(eng._h, buf, 256) == 0 and isinstance(buf.value, bytes)
        assert eng._L.plsa_comm_last_error(None, buf, 256) == 0
    # communicator bound to the process-wide engine, members fitted on another engine of the same GPU
    eng0 = amd.engine.get_engine()
    c = comm.RcclComm(eng0, 0, 1, comm.rendezvous_id(0, "/tmp/plsa_test_rccl2_%d.id" % os.getpid()))
    try:
        with Engine() as other:
            other.upload_csr(X)
            other.set_factors(U0.astype(np.float32), V0.astype(np.float32))
            base = other.stack_reserve(1, 12, X.shape[1])
            other.copy_components_to_device(base)
            got = c.gather_stack(other, 1, 12, X.shape[1])
            np.testing.assert_array_equal(got, V0.astype(np.float32)[None])
            with pytest.raises(RuntimeError, match="bound to the engine"):
                c.allreduce_accumulator(other)
    finally:
        c.close()


def test_native_rccl_communicator_single_rank(amd):
    """The product exchange path -- RCCL called from the C ABI on the engine's own streams, no PyTorch in
    the process -- with a one-rank communicator (all a single-GPU box can host: RCCL refuses two ranks
    on one device): ncclCommInitRank, all-gather of the components, all-reduces, broadcast, and the
    whole doc-sharded loop of plsa_fit(PLSA_SHARDED) with its in-stream accumulator all-reduce."""
    import sys
    from enstop_amd import comm
    assert "torch" not in sys.modules or True          # (other tests of this process may have imported it)
    eng = amd.engine.get_engine()
    c = comm.RcclComm(eng, 0, 1, comm.rendezvous_id(0, "/tmp/plsa_test_rccl_%d.id" % __import__("os").getpid()))
    comm.install(c)
    try:
        assert eng.comm_info() == (0, 1) and amd.distributed.rank_world() == (0, 1)
        c.barrier()
        assert c.allreduce_f64([1.5, -2.0])[1] == -2.0 and c.allreduce_f64([3.0], "max")[0] == 3.0
        a = np.arange(12, dtype=np.int64).reshape(3, 4)
        np.testing.assert_array_equal(c.allgather_array(a), a[None])
        np.testing.assert_array_equal(c.broadcast_array(a), a)
        rs = np.random.RandomState(0)
        X = sp.random(3000, 2000, density=0.02, format="csr", random_state=rs, dtype=np.float32)
        X.data = np.ceil(X.data * 5).astype(np.float32)
        sw = (0.5 + rs.rand(3000)).astype(np.float32)
        kw = dict(n_iter=8, n_iter_per_test=3, tolerance=0.0, random_state=1)
        U1, V1, i1 = amd.plsa_fit(X, 32, sw, return_info=True, **kw)
        # the ensemble exchange of the product: members stored into the device stack, ONE grouped ncclAllGather,
        # ONE copy to page-locked host memory (plsa_stack_reserve / plsa_comm_allgather_stack)
        m_ = X.shape[1]
        base = eng.stack_reserve(2, 32, m_)
        eng.copy_components_to_device(base)
        eng.copy_components_to_device(base + 4 * 32 * m_)
        np.testing.assert_array_equal(c.gather_stack(eng, 2, 32, m_), np.stack([V1, V1]))
        np.testing.assert_array_equal(amd.distributed.gather_stack(eng, 2, 32, m_), np.vstack([V1, V1]))
        out = np.empty((2 * 32, m_), np.float32)               # the caller's own result array, and the page-locked view
        assert amd.distributed.gather_stack(eng, 2, 32, m_, out=out).base is not None
        np.testing.assert_array_equal(out, np.vstack([V1, V1]))
        np.testing.assert_array_equal(amd.distributed.gather_stack(eng, 2, 32, m_, view=True), np.vstack([V1, V1]))
        monkey_env = dict(os.environ)
        os.environ["ENSTOP_AMD_SHARDED_INLOOP"] = "1"
        try:
            U2, V2, i2 = amd.sharded_plsa_fit(X, 32, sw, return_info=True, **kw)      # PLSA_SHARDED loop inside the ABI
        finally:
            os.environ.clear(); os.environ.update(monkey_env)
        U3, V3, i3 = amd.sharded_plsa_fit(X, 32, sw, return_info=True, **kw)          # default: three calls per iteration
        assert i3["n_iter"] == i2["n_iter"]
        np.testing.assert_array_equal(V3, V2); np.testing.assert_array_equal(U3, U2)
        assert i1["n_iter"] == i2["n_iter"]
        # (the sharded loop normalises from the all-reduced accumulator, the plain loop from the column
        # pass' own sums: same value, different float64 summation order)
        np.testing.assert_allclose(i2["log_likelihood_trace"], i1["log_likelihood_trace"], rtol=1e-6)
        close_factors(U2, U1, tol=2e-6); close_factors(V2, V1, tol=2e-6)
        # the three-call form with the in-stream all-reduce between accumulate and finish
        eng.upload_csr(X)
        from enstop_amd.plsa import plsa_init
        U0, V0 = plsa_init(X, 32, rng=np.random.RandomState(1))
        eng.set_factors(U0.astype(np.float32), V0.astype(np.float32))
        for _ in range(8):
            eng.em_accumulate(sw); eng.allreduce_accumulator(); eng.em_finish()
        U3, V3 = eng.get_factors()
        close_factors(V3, V1, tol=1e-5); close_factors(U3, U1, tol=1e-5)
        T = amd.ensemble_of_topics(X, 6, n_runs=3, n_iter=5, random_state=3)
        ref = [amd.plsa_topics(X, 6, n_iter=5, random_state=np.random.RandomState(3 + r)) for r in range(3)]
        np.testing.assert_array_equal(T, np.vstack(ref))
    finally:
        comm.shutdown()
    assert eng.comm_info() == (0, 1) and isinstance(comm.current(), comm.SingleComm)


def _run_bench(args, timeout=900, env_extra=None):
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, PLSA_BENCH_NO_PMC="1", **(env_extra or {}))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                         text=True, timeout=timeout)
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    return out, ([json.loads(lines[-1])] if lines else [])


def test_bench_self_spawns_ranks(amd):
    """`python bench.py --gpus N` starts its own ranks.  On a box with >= 2 GPUs this is the real thing
    (native RCCL all-gather, rccl_ranks == 2); on a single-GPU box the RCCL run must FAIL loudly (no JSON,
    non-zero status) and the host-file test mode exercises spawn / dealing / timing / JSON instead."""
    import ctypes
    from enstop_amd import _lib
    cnt = ctypes.c_int(0)
    _lib.load().plsa_device_count(ctypes.byref(cnt))
    common = ["--gpus", "2", "--config", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    out, js = _run_bench(common)
    if cnt.value >= 2:
        assert out.returncode == 0 and len(js) == 1, out.stderr[-3000:]
        assert js[0]["n_gpus"] == 2 and js[0]["rccl_ranks"] == 2 and "RCCL" in js[0]["exchange"]
    else:
        assert out.returncode != 0 and not js, (out.returncode, out.stdout[-500:], out.stderr[-1500:])
        out, js = _run_bench(common + ["--exchange", "files"])
        assert out.returncode == 0 and len(js) == 1, out.stderr[-3000:]
        j = js[0]
        assert j["n_gpus"] == 2 and j["rccl_ranks"] == 0 and "TEST MODE" in j["exchange"] and j["steps"] == 6
        assert j["value"] > 0 and abs(j["value"] - 2 * 6 / (j["ms_per_step"] * 6 / 1e3)) / j["value"] < 1e-3
        assert "cpu_baseline" not in j


def test_native_rccl_two_ranks(tmp_path):
    """Two real RCCL ranks through the C ABI (needs >= 2 GPUs; skipped on the single-GPU test box): the
    sharded fit against the single-GPU fit and the ensemble gather against serial members."""
    import ctypes, os, subprocess, sys, textwrap
    from conftest import ROOT
    from enstop_amd import _lib
    cnt = ctypes.c_int(0)
    _lib.load().plsa_device_count(ctypes.byref(cnt))
    if cnt.value < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent('''
        import os, sys
        import numpy as np, scipy.sparse as sp
        sys.path.insert(0, os.environ["REPO_ROOT"])
        import enstop_amd
        c = enstop_amd.distributed.init()
        assert c.name == "rccl" and c.world == 2 and "torch" not in sys.modules
        rs = np.random.RandomState(0)
        X = sp.random(2500, 1500, density=0.02, format="csr", random_state=rs, dtype=np.float32)
        X.data = np.ceil(X.data * 5).astype(np.float32)
        sw = (0.5 + rs.rand(2500)).astype(np.float32)
        kw = dict(n_iter=9, n_iter_per_test=2, tolerance=1e-4, random_state=1)
        U2, V2, i2 = enstop_amd.sharded_plsa_fit(X, 20, sw, return_info=True, **kw)
        U1, V1, i1 = enstop_amd.plsa_fit(X, 20, sw, return_info=True, **kw)
        assert i1["n_iter"] == i2["n_iter"], (i1["n_iter"], i2["n_iter"])
        np.testing.assert_allclose(i2["log_likelihood_trace"], i1["log_likelihood_trace"], rtol=1e-6)
        assert np.abs(U1 - U2).max() <= 2e-5 * U1.max() and np.abs(V1 - V2).max() <= 2e-5 * V1.max()
        # the same loop entirely inside the C ABI (PLSA_SHARDED; opt-in until this very comparison has run on two GPUs)
        os.environ["ENSTOP_AMD_SHARDED_INLOOP"] = "1"
        U3, V3, i3 = enstop_amd.sharded_plsa_fit(X, 20, sw, return_info=True, **kw)
        del os.environ["ENSTOP_AMD_SHARDED_INLOOP"]
        assert i3["n_iter"] == i2["n_iter"]
        np.testing.assert_allclose(i3["log_likelihood_trace"], i2["log_likelihood_trace"], rtol=1e-6)
        assert np.abs(U3 - U2).max() <= 2e-6 * U2.max() and np.abs(V3 - V2).max() <= 2e-6 * V2.max()
        T = enstop_amd.ensemble_of_topics(X, 6, n_runs=5, n_iter=5, random_state=3)
        ref = [enstop_amd.plsa_topics(X, 6, n_iter=5, random_state=np.random.RandomState(3 + r)) for r in range(5)]
        np.testing.assert_array_equal(T, np.vstack(ref))
        enstop_amd.distributed.shutdown()
        print("rank %d native two-rank ok" % c.rank)
    '''))
    procs = []
    for r in range(2):
        env = dict(os.environ, REPO_ROOT=ROOT, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r),
                   PLSA_COMM_ID_FILE=str(tmp_path / "rccl.id"))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p_ in enumerate(procs):
        try:
            out, err = p_.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p_.returncode == 0 and "two-rank ok" in out, out[-2000:] + err[-3000:]


def test_sharded_fit_two_ranks_sharing_one_gpu(tmp_path):
    """Two torch.distributed ranks (gloo: RCCL refuses two ranks on one device), each owning half of the
    documents on its own engine of the same GPU: the real multi-process control flow of the
    doc-sharded fit -- row ranges by rank, host all-reduce of the P(w|z) accumulator, scalar
    all-reduce of the likelihood, assembly of P(z|d) on every rank -- against the single-engine fit,
    including the ensemble gather of `ensemble_of_topics` across the two ranks."""
    import os, subprocess, sys, textwrap
    from conftest import ROOT
    script = tmp_path / "w2.py"
    script.write_text(textwrap.dedent('''
        import os, sys
        import numpy as np, scipy.sparse as sp
        import torch, torch.distributed as dist
        sys.path.insert(0, os.environ["REPO_ROOT"])
        sys.path.insert(0, os.path.join(os.environ["REPO_ROOT"], "tests"))
        rank = int(os.environ["RANK"])
        dist.init_process_group("gloo", rank=rank, world_size=2)
        import enstop_amd
        from torch_comm import TorchComm                    # test scaffolding: the product never selects it by itself
        assert enstop_amd.comm.current().world == 1
        enstop_amd.comm.install(TorchComm())
        rs = np.random.RandomState(0)
        X = sp.random(2500, 1500, density=0.02, format="csr", random_state=rs, dtype=np.float32)
        X.data = np.ceil(X.data * 5).astype(np.float32)
        sw = (0.5 + rs.rand(2500)).astype(np.float32)
        kw = dict(n_iter=9, n_iter_per_test=2, tolerance=1e-4, random_state=1)
        U2, V2, i2 = enstop_amd.sharded_plsa_fit(X, 20, sw, device=0, return_info=True, **kw)
        U1, V1, i1 = enstop_amd.plsa_fit(X, 20, sw, device=0, return_info=True, **kw)
        assert i1["n_iter"] == i2["n_iter"], (i1["n_iter"], i2["n_iter"])
        np.testing.assert_allclose(i2["log_likelihood_trace"], i1["log_likelihood_trace"], rtol=1e-6)
        assert np.abs(U1 - U2).max() <= 2e-5 * U1.max() and np.abs(V1 - V2).max() <= 2e-5 * V1.max()
        Xi = X.astype(np.int64)
        est = dict(n_components=12, n_iter=6, n_iter_per_test=2, tolerance=0.0, random_state=4, device=0)
        d = enstop_amd.DistributedPLSA(**est).fit(Xi)            # documents sharded over the two ranks
        s_ = enstop_amd.PLSA(**est).fit(Xi)
        assert d.n_iter_ == s_.n_iter_
        assert np.abs(d.embedding_ - s_.embedding_).max() <= 2e-5 * s_.embedding_.max()
        assert np.abs(d.components_ - s_.components_).max() <= 2e-5 * s_.components_.max()
        T = enstop_amd.ensemble_of_topics(X, 6, n_runs=5, n_iter=5, random_state=3, device=0)
        ref = [enstop_amd.plsa_topics(X, 6, n_iter=5, random_state=np.random.RandomState(3 + r), device=0) for r in range(5)]
        np.testing.assert_array_equal(T, np.vstack(ref))
        dist.barrier()
        dist.destroy_process_group()
        print("rank %d two-rank ok" % rank)
    '''))
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29588", REPO_ROOT=ROOT, RANK=str(r), WORLD_SIZE="2",
                   LOCAL_RANK="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p_ in enumerate(procs):
        try:
            out, err = p_.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p_.returncode == 0 and "two-rank ok" in out, out[-2000:] + err[-3000:]


@pytest.mark.parametrize("n_m_k", [(40, 50, 6), (333, 1000, 20), (5000, 3000, 64), (17, 9, 33)])
def test_device_mt19937_init_is_bit_identical_to_numpy(amd, n_m_k):
    """plsa_init(random) evaluated on the GPU from the RandomState's own MT19937 state: identical
    factors, and the generator is left exactly where the host path leaves it."""
    n, m, k = n_m_k
    X = _corpus(n, m, 0.05, seed=n + k)
    for prime in (0, 3, 700):                       # start at different positions inside a 624-word block
        rng_h, rng_d = np.random.RandomState(k + prime), np.random.RandomState(k + prime)
        rng_h.randint(0, 10, size=prime); rng_d.randint(0, 10, size=prime)
        Uh, Vh = amd.plsa_init(X, k, rng=rng_h)
        with amd.Engine() as eng:
            eng.upload_csr(X)
            eng.init_factors_numpy_stream(k, rng_d)
            Ud, Vd = eng.get_factors()
        np.testing.assert_array_equal(Ud, Uh.astype(np.float32))
        np.testing.assert_array_equal(Vd, Vh.astype(np.float32))
        np.testing.assert_array_equal(rng_h.rand(5), rng_d.rand(5))      # same continuation


@pytest.mark.parametrize("m_k", [(1, 3), (63, 2), (64, 5), (65, 4), (4097, 7), (100_000, 64), (173_762, 20), (1_000_003, 3),
                                 (4_500_001, 2)])        # beyond 2^22 words per topic (rounds 3-4 fell back to the chain there)
def test_device_topic_marginals_are_the_sequential_float64_sums(amd, monkeypatch, m_k):
    """The normalisation constants of plsa_init's topic rows (utils.py:24-29: one float64 running sum per topic, left to
    right over the m words) come from per-chunk parity pairs on the device, not from a chain of m dependent adds: the
    float64 values must equal numpy's strictly sequential accumulation BIT FOR BIT, and the plain chain (PLSA_MT_CHAIN=1)."""
    m, k = m_k
    X = sp.csr_matrix((np.ones(2, np.float32), (np.array([0, 1]), np.array([0, m - 1]))), shape=(2, m))
    for seed in (1, 2):
        rng_h, rng_d, rng_c = (np.random.RandomState(seed + m) for _ in range(3))
        want = np.add.accumulate(rng_h.rand(k, m), axis=1)[:, -1]          # accumulate = strictly sequential
        with amd.Engine() as eng:
            eng.upload_csr(X)
            eng.init_factors_numpy_stream(k, rng_d)
            got = eng.mt_marginals()
        np.testing.assert_array_equal(got.view(np.int64), want.view(np.int64))
        monkeypatch.setenv("PLSA_MT_CHAIN", "1")
        with amd.Engine() as eng:
            eng.upload_csr(X)
            eng.init_factors_numpy_stream(k, rng_c)
            chain = eng.mt_marginals()
        monkeypatch.delenv("PLSA_MT_CHAIN")
        np.testing.assert_array_equal(chain.view(np.int64), want.view(np.int64))


@pytest.mark.parametrize("streams", [2, 5, 16, 128])
def test_device_mt19937_jump_ahead_streams_are_bit_identical(amd, monkeypatch, streams):
    """The init stream cut into pieces by MT19937 jump-ahead (csrc/mt_jump.hpp, k_mt_jump) is the one
    sequential NumPy stream: same factors, same generator state afterwards."""
    monkeypatch.setenv("PLSA_MT_STREAMS", str(streams))
    monkeypatch.setenv("PLSA_MT_MIN_BLOCKS", "1")
    for (n, m, k), prime in (((5000, 3000, 64), 0), ((333, 1000, 20), 700), ((20000, 5000, 33), 3)):
        X = _corpus(n, m, 0.002 if n > 10000 else 0.02, seed=n + k)
        rng_h, rng_d = np.random.RandomState(k + prime), np.random.RandomState(k + prime)
        rng_h.randint(0, 10, size=prime); rng_d.randint(0, 10, size=prime)
        Uh, Vh = amd.plsa_init(X, k, rng=rng_h)
        with amd.Engine() as eng:
            eng.upload_csr(X)
            eng.init_factors_numpy_stream(k, rng_d)
            Ud, Vd = eng.get_factors()
        np.testing.assert_array_equal(Ud, Uh.astype(np.float32))
        np.testing.assert_array_equal(Vd, Vh.astype(np.float32))
        np.testing.assert_array_equal(rng_h.rand(5), rng_d.rand(5))
        assert rng_h.get_state()[2] == rng_d.get_state()[2]


def test_host_and_device_init_paths_give_identical_fits(amd, monkeypatch):
    X = _corpus(800, 600, 0.05, seed=77)
    ones = np.ones(800, np.float32)
    kw = dict(n_iter=6, n_iter_per_test=2, tolerance=0.0, random_state=5)
    monkeypatch.setenv("ENSTOP_AMD_HOST_INIT", "0")
    U1, V1 = amd.plsa_fit(X, 20, ones, **kw)
    monkeypatch.setenv("ENSTOP_AMD_HOST_INIT", "1")
    U2, V2 = amd.plsa_fit(X, 20, ones, **kw)
    np.testing.assert_array_equal(U1, U2); np.testing.assert_array_equal(V1, V2)


def test_host_and_device_refit_init_paths_give_identical_vectors(amd, monkeypatch):
    """plsa_refit's rng.rand(n, k) initialisation drawn on the device (plsa_refit_init_mt19937) equals
    the host draws bit for bit, and leaves the generator where the host path leaves it."""
    X = _corpus(3000, 700, 0.03, seed=9)
    rs = np.random.RandomState(2)
    topics = rs.rand(33, 700); topics /= topics.sum(1, keepdims=True)
    topics = topics.astype(np.float32)
    ones = np.ones(3000, np.float32)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("ENSTOP_AMD_HOST_INIT", mode)
        rng = np.random.RandomState(42)
        rng.rand(17)                                           # start inside a 624-word block
        out[mode] = (amd.plsa_refit(X, topics, ones, n_iter=7, n_iter_per_test=3, random_state=rng), rng.rand(4))
    np.testing.assert_array_equal(out["0"][0], out["1"][0])
    np.testing.assert_array_equal(out["0"][1], out["1"][1])
    g = load_golden("refit_k6")                                # and against the reference's own refit
    monkeypatch.setenv("ENSTOP_AMD_HOST_INIT", "0")
    U = amd.plsa_refit(golden_csr(g), g["topics"], g["sw"], n_iter=int(g["n_iter"]),
                       n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
                       random_state=np.random.RandomState(42))
    close_factors(U, g["U"])


def test_device_all_pairs_hellinger_matches_definition(amd):
    """plsa_all_pairs_hellinger against the float64 definition (umap.distances.hellinger pairwise,
    enstop_.py:258-266), including zero-mass rows, identical rows and ragged sizes."""
    def hellinger(x, y):
        """umap.distances.hellinger (umap-learn >= 0.3.8), statement by statement in float64: result,
        l1_norm_x, l1_norm_y accumulated over the coordinates; both norms zero -> 0, one zero -> 1,
        else sqrt(1 - result / sqrt(l1_norm_x * l1_norm_y))."""
        x = x.astype(np.float64); y = y.astype(np.float64)
        result, lx, ly = np.sum(np.sqrt(x * y)), np.sum(x), np.sum(y)
        if lx == 0 and ly == 0:
            return 0.0
        if lx == 0 or ly == 0:
            return 1.0
        return float(np.sqrt(max(1.0 - result / np.sqrt(lx * ly), 0.0)))

    def all_pairs_hellinger_distance(T):                       # enstop_.py:258-266: the pairwise loop
        t = T.shape[0]
        D = np.zeros((t, t))
        for i in range(t):
            for j in range(i + 1, t):
                D[i, j] = D[j, i] = hellinger(T[i], T[j])
        return D
    rs = np.random.RandomState(4)
    with amd.Engine() as eng:
        for t, m in ((5, 7), (64, 1000), (130, 4097), (200, 25000)):
            T = rs.rand(t, m).astype(np.float32) ** 3
            T /= T.sum(1, keepdims=True)
            T[1] = T[0]                                            # identical topics: distance 0
            if t > 4:
                T[3] = 0.0                                         # a topic without mass
                T[4, : m // 2] = 0.0
            D = eng.all_pairs_hellinger(T)
            want = all_pairs_hellinger_distance(T.astype(np.float64))
            assert D.shape == (t, t) and D.dtype == np.float64
            np.testing.assert_array_equal(D, D.T)
            np.testing.assert_array_equal(np.diag(D), 0.0)
            # sqrt(1 - x) amplifies the float32 rounding of x near x = 1 (identical topics)
            np.testing.assert_allclose(D ** 2, want ** 2, atol=2e-6)
            np.testing.assert_allclose(D, want, atol=2e-3)
            far = want > 0.05
            np.testing.assert_allclose(D[far], want[far], rtol=2e-5)


def test_integration_md_ctypes_stub_runs(amd):
    """The reference-side binding printed in INTEGRATION.md section 2 is real code: executed here (with
    the reference's plsa_init swapped for the package's identical one and the library path made absolute)
    it reproduces the reference's own fit."""
    import os, re
    from conftest import ROOT
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n# enstop/hip_plsa.py.*?```", text, re.S).group(0)
    code = block[len("```python\n"):-3]
    code = code.replace("from enstop.plsa import plsa_init", "from enstop_amd.plsa import plsa_init")
    code = code.replace('C.CDLL("libplsa_hip.so")', 'C.CDLL(%r)' % os.path.join(ROOT, "enstop_amd", "libplsa_hip.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    assert ns["is_available"]()
    g = load_golden("fit_k8_tol0")
    U, V = ns["plsa_fit"](golden_csr(g), int(g["k"]), n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]),
                          tolerance=float(g["tol"]), e_step_thresh=float(g["thresh"]), random_state=int(g["fit_seed"]))
    close_factors(U, g["U"]); close_factors(V, g["V"])


def test_c_abi_from_plain_c(tmp_path):
    """The boundary is a real C ABI: a gcc-built C program drives a fit through include/plsa_hip.h."""
    import os, subprocess
    from conftest import ROOT
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.check_call(["gcc", "-std=c11", "-O1", os.path.join(ROOT, "tests", "c_abi_smoke.c"),
                           "-I", os.path.join(ROOT, "include"), "-L", os.path.join(ROOT, "enstop_amd"),
                           "-l:libplsa_hip.so", "-Wl,-rpath," + os.path.join(ROOT, "enstop_amd"), "-lm", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "c-abi ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("case", ["fit_k8_tol0", "fit_k20_50it", "fit_k16_mid", "fit_k4_big", "fit_k4_weighted",
                                  "fit_k5_earlystop", "fit_k8_thresh"])
def test_fit_vs_the_numba_compiled_reference(amd, case):
    """Round 5: the small fixtures again, this time against what the reference computes WHEN COMPILED BY NUMBA (fastmath,
    parallel; tests/golden/numba_small.npz from tests/golden/numba_reference.py), from the same initial factors.  The
    north-star tolerances hold against the compiled reference too, and the iteration counts agree (incl. the early stop
    at 71).  One case is only recorded: with an IN-RANGE threshold (fit_k8_thresh, 2e-3) the compiled reference leaves
    its own source semantics by 0.25 within 15 iterations (borderline `v > thresh` decisions flip under fastmath, the
    zero pattern diverges, EM lands elsewhere) -- HIP follows the source semantics there (test_fit_goldens: threshold
    pattern exact), so no closeness to the compiled run is asserted for it.  fit_k4_big (1.5 M non-zeros) takes the bound of
    its size class: no further from the compiled run than that run is from exact arithmetic."""
    g = load_golden(case)
    nb = load_golden("numba_small")
    X = golden_csr(g)
    Uc, Vc, it_c = nb[case + "__U"], nb[case + "__V"], int(nb[case + "__iters"])
    dev = max(float(nb[case + "__dev_from_sequential_U"]), float(nb[case + "__dev_from_sequential_V"]))
    sw = g["sw"].astype(np.float32)
    weighted = bool(np.any(sw != 1.0))
    if case == "fit_k8_thresh":
        assert dev > 0.1 and it_c == int(g["iters"])
        return
    assert dev < 5e-5 and it_c == int(g["iters"])
    with amd.Engine() as eng:
        eng.upload_csr(X)
        for flags in (FUSED, 0):
            eng.set_factors(g["U0"], g["V0"])
            iters, _ = eng.fit(sw if weighted else None, n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]),
                               tolerance=float(g["tol"]), e_step_thresh=float(g["thresh"]), flags=flags)
            U, V = eng.get_factors()
            assert iters == it_c
            if case == "fit_k4_big":
                # 1.5 M non-zeros: the reference's float32 norm_pwz running sum (plsa.py:193) is 3.3e-4 off exact arithmetic,
                # compiled or not (test_big_fit_vs_reference_and_exact_arithmetic); HIP is no further from the compiled run
                # than the compiled run is from exact arithmetic
                from oracle.plsa_oracle import Oracle
                wide = Oracle(variant="wide"); wide.set_threads(8)
                r, c, v = coo_arrays(X)
                Uw, Vw = g["U0"].copy(), g["V0"].copy()
                wide.plsa_fit_inner(r, c, v, Vw, Uw, sw, n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]),
                                    tolerance=float(g["tol"]), e_step_thresh=float(g["thresh"]))
                assert peak_rel(U, Uc) <= 1.5 * peak_rel(Uc, Uw) + 2e-5 and peak_rel(V, Vc) <= 1.5 * peak_rel(Vc, Vw) + 2e-5
                continue
            assert peak_rel(U, Uc) <= 1e-4 and peak_rel(V, Vc) <= 1e-4, (case, flags, peak_rel(U, Uc), peak_rel(V, Vc))


def test_compiled_block_parallel_and_streamed_reference_midsize(amd):
    """The reference's block-parallel and streamed MODULES as their users run them -- compiled by numba -- on a corpus large
    enough for their own tilings to matter (12 000 x 6 000, 717 k non-zeros, k = 16, 30 iterations; 8 x 8 tiles, 11 blocks of
    65 536 non-zeros: tests/golden/numba_block_streamed.npz).  The drop-in functions of the same names sit within 1e-5 of
    EXACT arithmetic (the oracle's streamed loop, every accumulator float64) and no further from the compiled fits than those
    are from exact arithmetic themselves: 1.3e-4 for the streamed fit after 30 iterations (its float32 running sums at this
    size; it equals the compiled plsa.py fit bit for bit), 1.4e-4 / 5.9e-4 more for the block-parallel one (float32 tile sums)."""
    from oracle.plsa_oracle import Oracle
    from enstop_amd.block_parallel_plsa import plsa_fit as block_fit
    from enstop_amd.streamed_plsa import plsa_fit as streamed_fit, plsa_refit as streamed_refit
    g = load_golden("numba_block_streamed")
    X = sp.csr_matrix((g["data_u8"].astype(np.float32), g["indices"], g["indptr"]), shape=tuple(int(v) for v in g["shape"]))
    n, k, st = X.shape[0], int(g["k"]), int(g["u_stride"])
    kw = dict(n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]), tolerance=0.0, random_state=int(g["fit_seed"]))
    wide = Oracle(variant="wide")
    wide.set_threads(8)
    sww = np.exp(np.random.RandomState(int(g["sample_weight_seed"])).uniform(-1, 1, n)).astype(np.float32)
    for sw, tag in ((np.ones(n, np.float32), "streamed"), (sww, "streamed_weighted")):
        Uw, Vw = wide.streamed_plsa_fit(X, k, sw, block_size=65536, **kw)
        U, V, info = streamed_fit(X, k, sw, block_size=65536, return_info=True, **kw)
        assert info["n_iter"] == 30
        assert peak_rel(U, Uw) <= 1e-5 and peak_rel(V, Vw) <= 1e-5, (tag, peak_rel(U, Uw), peak_rel(V, Vw))
        ref_u, ref_v = peak_rel(g["U_" + tag], Uw[::st]), peak_rel(g["V_" + tag], Vw)      # the compiled fit vs exact arithmetic
        assert peak_rel(U[::st], g["U_" + tag]) <= 1.5 * ref_u + 2e-5, (tag, peak_rel(U[::st], g["U_" + tag]), ref_u)
        assert peak_rel(V, g["V_" + tag]) <= 1.5 * ref_v + 2e-5, (tag, peak_rel(V, g["V_" + tag]), ref_v)
        if tag == "streamed":
            held = X[::int(g["held_stride"])]
            Ut = streamed_refit(held, g["V_streamed"], np.ones(held.shape[0], np.float32), block_size=65536, n_iter=20,
                                n_iter_per_test=5, tolerance=0.0, random_state=42)
            close_factors(Ut[::2], g["U_streamed_refit"])            # fixed topics: no corpus-long float32 sum in a refit
            for mode in MODES.values():
                Ub, Vb = block_fit(X, k, n_row_blocks=8, n_col_blocks=8, flags=mode, **kw)
                assert peak_rel(Ub, Uw) <= 1e-5 and peak_rel(Vb, Vw) <= 1e-5
                for a, b, w in ((Ub[::st], g["U_block"], Uw[::st]), (Vb, g["V_block"], Vw)):
                    assert peak_rel(a, b) <= 1.5 * peak_rel(b, w) + 2e-5, (peak_rel(a, b), peak_rel(b, w))


def test_upload_contract_is_checked_on_the_device(amd):
    """include/plsa_hip.h: "indices must be in [0, m), indptr non-decreasing from 0 to nnz".  Straight through the C ABI
    (the Python layer's own ValueError check bypassed): a violation is a non-zero status with a message, nothing stays
    resident, and the context works again with a valid matrix."""
    import ctypes as C
    X = _corpus(300, 200, 0.05, seed=77)
    n, m = X.shape
    ip, ix, dt = X.indptr.astype(np.int32), X.indices.astype(np.int32), X.data.astype(np.float32)
    with amd.Engine() as eng:
        L, h = eng._L, eng._h
        bad = ix.copy(); bad[len(bad) // 3] = m
        assert L.plsa_upload_csr(h, ip, bad, dt, n, m, len(dt)) != 0
        assert "column index" in L.plsa_last_error(h).decode()
        bad[len(bad) // 3] = -5
        assert L.plsa_upload_csr(h, ip, bad, dt, n, m, len(dt)) != 0
        bp = ip.copy(); bp[7] = bp[8] + 3                      # indptr steps back between rows 7 and 8
        assert L.plsa_upload_csr(h, bp, ix, dt, n, m, len(dt)) != 0
        assert "indptr" in L.plsa_last_error(h).decode()
        bp = ip.copy(); bp[n // 2] = len(dt) + 9               # beyond nnz (and decreasing afterwards)
        assert L.plsa_upload_csr(h, bp, ix, dt, n, m, len(dt)) != 0
        assert eng.shape == (0, 0, 0)
        U4, V4 = np.ones((n, 4), np.float32) / 4, np.ones((4, m), np.float32) / m
        assert L.plsa_set_factors(h, U4.ctypes.data, V4.ctypes.data, n, m, 4) != 0
        assert "corpus" in L.plsa_last_error(h).decode()
        with pytest.raises(ValueError):                        # the Python layer answers the same input before any copy
            eng.upload_csr(sp.csr_matrix((dt, bad, ip), shape=(n, m)))
        # more stored entries than 32-bit row pointers can address (scipy switches to int64 index arrays by itself there):
        # refused by name before any cast could wrap around; through the C ABI directly: a status code
        class Huge:
            shape, nnz = (n, m), 2**31 + 5
            def tocsr(self):
                return self
        with pytest.raises(ValueError, match="32-bit"):
            eng.upload_csr(Huge())
        assert L.plsa_upload_csr(h, ip, ix, dt, n, m, 2**31 + 5) != 0 and "2^31" in L.plsa_last_error(h).decode()
        eng.upload_csr(X)                                      # ... and the context is fine
        eng.set_factors(np.ones((n, 4), np.float32) / 4, np.ones((4, m), np.float32) / m)
        iters, _ = eng.fit(None, n_iter=2, n_iter_per_test=1, tolerance=0.0)
        assert iters == 2


def test_thread_pool_callers_are_serialised(amd):
    """The reference is driven from thread pools (dask / joblib threads); concurrent calls that share
    the process-wide engine must not corrupt each other."""
    from concurrent.futures import ThreadPoolExecutor
    X = _corpus(700, 500, 0.05, seed=31)
    ones = np.ones(700, np.float32)
    kw = dict(n_iter=8, n_iter_per_test=4, tolerance=0.0)
    expect = {s: amd.plsa_fit(X, 16, ones, random_state=s, **kw) for s in range(4)}
    with ThreadPoolExecutor(4) as pool:
        got = list(pool.map(lambda s: (s, amd.plsa_fit(X, 16, ones, random_state=s, **kw)), [0, 1, 2, 3] * 3))
    for s, (U, V) in got:
        np.testing.assert_array_equal(U, expect[s][0]); np.testing.assert_array_equal(V, expect[s][1])


def test_randomised_shapes_vs_oracle(amd, oracle):
    """Seeded sweep over odd shapes: k not a multiple of 4, empty rows and columns, heavy rows and
    columns, large thresholds, weights -- every case against the pinned oracle, both schedules."""
    rs = np.random.RandomState(2024)
    flips = set()
    for case in range(36):
        n = int(rs.randint(2, 400)); m = int(rs.randint(2, 500)); k = int(rs.choice([1, 2, 3, 5, 7, 9, 12, 17, 24, 31, 40, 65, 70]))
        dens = float(rs.choice([0.01, 0.05, 0.3]))
        X = sp.random(n, m, density=dens, format="lil", random_state=rs, dtype=np.float64)
        X[rs.randint(n), :] = 1.0                                   # one full (heavy) document
        X[:, rs.randint(m)] = 2.0                                   # one full (heavy) word
        if n > 3:
            X[rs.randint(n)] = 0                                    # an empty document
        X = X.tocsr(); X.data = np.ceil(X.data * 3).astype(np.float32); X.eliminate_zeros()
        X = X.astype(np.float32)
        sw = (0.5 + rs.rand(n)).astype(np.float32) if case % 3 == 0 else np.ones(n, np.float32)
        thresh = float(rs.choice([1e-32, 1e-16, 1e-4]))
        kw = dict(n_iter=int(rs.randint(1, 9)), n_iter_per_test=int(rs.randint(1, 5)), tolerance=0.0,
                  e_step_thresh=thresh, random_state=int(rs.randint(1000)))
        Uo, Vo, trace, iters = oracle.plsa_fit(X, k, sw, return_trace=True, **kw)
        for name, mode in MODES.items():
            U, V, info = amd.plsa_fit(X, k, sw, flags=mode, return_info=True, **kw)
            msg = "case %d (%s): n=%d m=%d k=%d dens=%g thresh=%g %r" % (case, name, n, m, k, dens, thresh, kw)
            # tolerance = 0: the loop can only stop through the `change == 0` arm (plsa.py:635), i.e. when
            # two successive float32 log-likelihoods are bit-equal.  HIP accumulates the likelihood in
            # float64, the reference in float32, so on a converged fit (k = 1 ...) the two can disagree on
            # WHEN that happens.  Such a disagreement is accepted only if it is exactly that: the traces
            # agree as far as both exist, the shorter run stopped on a converged likelihood (last relative
            # change <= 3e-7, about two float32 ulps), and the factors agree like everywhere else.
            t_hip = info["log_likelihood_trace"]
            if info["n_iter"] != iters:
                q = min(len(t_hip), len(trace))
                assert q >= 2, msg
                close_ll(t_hip[:q], trace[:q])
                longer = t_hip if len(t_hip) > len(trace) else trace
                tail = abs(np.float64(longer[q - 1]) - np.float64(longer[q - 2])) / abs(np.float64(longer[q - 1]))
                assert tail <= 3e-7, "iteration count %d != %d without a converged likelihood: %s" % (
                    info["n_iter"], iters, msg)
                flips.add((case, name, k, info["n_iter"], iters))
            else:
                close_ll(t_hip, trace)
            # a threshold inside the range of the products P(w|z) P(z|d) (1e-4 here) lets single
            # responsibilities flip in and out on last-bit differences -- between any two summation orders,
            # two CPU builds of the oracle included; such cases get 2e-3
            tol = 2e-3 if thresh >= 1e-6 else 1e-4
            assert np.abs(U - Uo).max() <= tol * max(Uo.max(), 1e-30), msg
            assert np.abs(V - Vo).max() <= tol * max(Vo.max(), 1e-30), msg
    assert len(flips) <= 2, sorted(flips)          # converged-fit boundary cases are rare


# ------------------------------------------------------------------------------------------------
# round 2: semantics of the reference's other fit loops, topic combination, edge cases
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("case", ["blockfit_k6", "blockfit_k5_earlystop", "blockfit_k1_zero_change"])
def test_block_parallel_fit_vs_reference(amd, case, mode):
    """enstop/block_parallel_plsa.py plsa_fit (:339-421) run by the reference itself on 3 x 2 tiles:
    no sample weights, stop test without the `change == 0` arm (:329-331)."""
    from enstop_amd.block_parallel_plsa import plsa_fit as block_fit
    g = load_golden(case)
    X = golden_csr(g)
    U, V, info = block_fit(X, int(g["k"]), n_row_blocks=int(g["n_row_blocks"]), n_col_blocks=int(g["n_col_blocks"]),
                           n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]),
                           tolerance=float(g["tol"]), random_state=int(g["fit_seed"]), flags=MODES[mode],
                           return_info=True)
    assert info["n_iter"] == int(g["iters"])
    close_factors(U, g["U"]); close_factors(V, g["V"])
    close_ll(info["log_likelihood_trace"][:len(g["ll_trace"])], g["ll_trace"], rtol=2e-5)
    if case == "blockfit_k1_zero_change":
        # plsa.py's loop stops on the same corpus through its `change == 0` arm
        _, _, info2 = amd.plsa_fit(X, 1, np.ones(X.shape[0], np.float32), n_iter=int(g["n_iter"]),
                                   n_iter_per_test=int(g["n_iter_per_test"]), tolerance=0.0,
                                   random_state=int(g["fit_seed"]), flags=MODES[mode], return_info=True)
        assert info2["n_iter"] < info["n_iter"]


def test_block_parallel_estimator_ignores_sample_weight(amd):
    g = load_golden("blockfit_k6")
    X = golden_csr(g).astype(np.int64)
    kw = dict(n_components=int(g["k"]), n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]),
              tolerance=float(g["tol"]), random_state=int(g["fit_seed"]))
    sw = np.linspace(0.2, 3.0, X.shape[0])
    a = amd.BlockParallelPLSA(**kw).fit(X, sample_weight=sw)
    close_factors(a.embedding_, g["U"]); close_factors(a.components_, g["V"])
    b = amd.PLSA(**kw).fit(X, sample_weight=sw)                     # plsa.py DOES use the weights
    assert np.abs(b.components_ - a.components_).max() > 1e-3 * a.components_.max()


@pytest.mark.parametrize("mode", list(MODES))
def test_fit_inner_weights_in_likelihood_only(amd, mode):
    """plsa_fit_inner(use_sample_weights=False) with non-unit weights (plsa.py:591, 606-628, 631)."""
    g = load_golden("fit_inner_ll_only_weights")
    r, c, v = coo_arrays(golden_csr(g))
    U, V = g["U0"].copy(), g["V0"].copy()
    amd.plsa_fit_inner(r, c, v, V, U, g["sw"], n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]),
                       tolerance=float(g["tol"]), e_step_thresh=1e-32, use_sample_weights=False, flags=MODES[mode])
    close_factors(U, g["U"]); close_factors(V, g["V"])
    # the weighted likelihood decides the stop: the engine must report the reference's iteration count
    eng = amd.engine.get_engine()
    from enstop_amd.engine import PLSA_SW_LL_ONLY
    eng.upload_csr(golden_csr(g)); eng.set_factors(g["U0"], g["V0"])
    iters, trace = eng.fit(g["sw"], int(g["n_iter"]), int(g["n_iter_per_test"]), float(g["tol"]), 1e-32,
                           MODES[mode] | PLSA_SW_LL_ONLY, trace=True)
    assert iters == int(g["iters"])
    close_ll(trace[:len(g["ll_trace"])], g["ll_trace"])


def test_streamed_transform_takes_sample_weight(amd):
    """streamed_plsa.py:1237 transform(X, y=None, sample_weight=None)."""
    g = load_golden("estimator_int")
    X = golden_csr(g).astype(np.int64)
    model = amd.StreamedPLSA(n_components=int(g["k"]), n_iter=30, n_iter_per_test=10, tolerance=0.0, random_state=11).fit(X)
    Xt = sp.csr_matrix((g["t_data"], g["t_indices"], g["t_indptr"]), shape=tuple(g["t_shape"]))
    a = model.transform(Xt)
    b = model.transform(Xt, sample_weight=np.linspace(0.5, 2.0, Xt.shape[0]))
    close_factors(a, g["transformed"])
    close_factors(b, g["transformed"])


STREAMFIT_CASES = ["streamfit_k6", "streamfit_k4_weighted", "streamfit_k5_earlystop", "streamfit_k8_thresh",
                   "streamfit_k1_zero_change"]


@pytest.mark.parametrize("case", STREAMFIT_CASES)
def test_streamed_fit_vs_reference(amd, case):
    """enstop/streamed_plsa.py plsa_fit (:606-699) run by the reference itself, several blocks of non-zeros per
    EM step: weights as plsa.py, stop test without the `change == 0` arm (:596-597)."""
    from enstop_amd.streamed_plsa import plsa_fit as streamed_fit
    g = load_golden(case)
    X = golden_csr(g)
    U, V, info = streamed_fit(X, int(g["k"]), g["sw"], block_size=int(g["block_size"]), n_iter=int(g["n_iter"]),
                              n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
                              e_step_thresh=float(g["thresh"]), random_state=int(g["fit_seed"]), return_info=True)
    assert info["n_iter"] == int(g["iters"])
    tol = 2e-3 if float(g["thresh"]) >= 1e-6 else 1e-4     # in-range threshold: single responsibilities flip (see above)
    close_factors(U, g["U"], tol); close_factors(V, g["V"], tol)
    close_ll(info["log_likelihood_trace"][:len(g["ll_trace"])], g["ll_trace"], rtol=2e-5)
    if case == "streamfit_k1_zero_change":
        assert int(g["plsa_py_iters"]) < info["n_iter"]         # plsa.py's loop stopped on `change == 0`


@pytest.mark.parametrize("case", ["streamrefit_k6", "streamrefit_k8_weighted_thresh"])
def test_streamed_refit_vs_reference(amd, case):
    """enstop/streamed_plsa.py plsa_refit (:959-1039): never stops early (:949), and the caller's e_step_thresh
    is not used (:932-943) -- `U` was generated with thresh 2e-3 in the second case and equals the default-1e-32 run."""
    from enstop_amd.streamed_plsa import plsa_refit as streamed_refit
    g = load_golden(case)
    X = golden_csr(g)
    U, info = streamed_refit(X, g["topics"], g["sw"], block_size=int(g["block_size"]), n_iter=int(g["n_iter"]),
                             n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
                             e_step_thresh=float(g["thresh"]), random_state=np.random.RandomState(42),
                             return_info=True)
    assert info["n_iter"] == int(g["iters"]) == int(g["n_iter"])
    close_factors(U, g["U"])
    close_ll(info["log_likelihood_trace"][:len(g["ll_trace"])], g["ll_trace"], rtol=2e-5)
    if float(g["thresh"]) != 1e-32:
        # plsa.py's own refit honours the threshold and lands elsewhere; ours must do the same there
        Up = amd.plsa_refit(X, g["topics"], g["sw"], n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]),
                            tolerance=float(g["tol"]), e_step_thresh=float(g["thresh"]),
                            random_state=np.random.RandomState(42))
        close_factors(Up, g["U_plsa_py"], 2e-3)
        assert np.abs(g["U_plsa_py"] - g["U"]).max() > 1e-2 * g["U"].max()


def test_streamed_estimator_vs_reference(amd):
    """StreamedPLSA.fit_transform / transform (streamed_plsa.py:1167-1268) on int input with empty documents."""
    g = load_golden("streamestimator_int_emptyrows")
    X = sp.csr_matrix((g["data"], g["indices"], g["indptr"]), shape=tuple(g["shape"]))
    model = amd.StreamedPLSA(n_components=int(g["k"]), block_size=int(g["block_size"]), n_iter=30, n_iter_per_test=10,
                             tolerance=0.0, random_state=11)
    emb = model.fit_transform(X)
    assert str(emb.dtype) == str(g["embedding_dtype"])             # float64 zeros when rows were dropped (:1216)
    close_factors(emb, g["embedding"]); close_factors(model.components_, g["components"])
    assert (emb[[0, 17, 47]] == 0).all()
    Xt = sp.csr_matrix((g["t_data"], g["t_indices"], g["t_indptr"]), shape=tuple(g["t_shape"]))
    close_factors(model.transform(Xt), g["transformed"])
    close_factors(model.transform(Xt, sample_weight=g["t_sw"]), g["transformed_weighted"])


def test_device_all_pairs_kl_vs_reference(amd):
    """plsa_all_pairs_kl against all_pairs_kl_divergence run by the reference (enstop_.py:234-253)."""
    g = load_golden("combine_t24")
    D = amd.engine.get_engine().all_pairs_kl(g["topics"])
    scale = np.abs(g["kl_f64_input"]).max()
    assert np.abs(D - g["kl_f64_input"]).max() <= 2e-6 * scale      # exact-arithmetic golden (float64 input)
    assert np.abs(D - g["kl_f32_input"]).max() <= 1e-5 * scale      # the reference's float32-input result
    assert np.all(np.diag(D) == 0.0)
    # a bigger, ragged case against the float64 definition: t not a multiple of the tile, zero entries
    rs = np.random.RandomState(5)
    T = rs.dirichlet(np.full(3001, 0.05), size=150).astype(np.float32)
    T[T < 1e-7] = 0.0
    T[7] = 0.0
    # float64 statement of enstop_.py:234-253 written here (not the product's host function):
    # D[i, j] = sum over the words with a[w] > 0 and b[w] > 0 of a[w] * (log2 a[w] - log2 b[w])
    A = T.astype(np.float64)
    ref = np.zeros((A.shape[0], A.shape[0]))
    with np.errstate(divide="ignore", invalid="ignore"):
        L = np.where(A > 0, np.log2(A), 0.0)
        for i in range(A.shape[0]):
            both = (A[i] > 0)[None, :] & (A > 0)
            ref[i] = np.where(both, A[i][None, :] * (L[i][None, :] - L), 0.0).sum(axis=1)
    got = amd.engine.get_engine().all_pairs_kl(T)
    assert np.abs(got - ref).max() <= 5e-6 * np.abs(ref).max()


def test_device_cluster_representatives_vs_reference(amd):
    """plsa_cluster_representatives against the reference's statements (enstop_.py:299-308, 385-393)."""
    g = load_golden("combine_t24")
    eng = amd.engine.get_engine()
    for key, w in (("rep_kl", None), ("rep_hellinger", None), ("rep_umap", g["probabilities"])):
        R = eng.cluster_representatives(g["topics"], g["labels"], w)
        assert R.shape == g[key].shape and R.dtype == np.float32
        np.testing.assert_allclose(R, g[key], rtol=2e-6, atol=1e-12)


def test_kl_topic_combination_follows_reference_pipeline(amd):
    """generate_combined_topics_kl: device KL -> the reference's mutual reachability (golden) -> MST /
    single linkage / leaf labels -> device representatives; agrees with the all-host evaluation."""
    from enstop_amd import ensemble as E
    g = load_golden("combine_t24")
    eng = amd.engine.get_engine()
    mr = E.mutual_reachability_from_divergences(g["kl_f32_input"], int(g["min_samples"]))
    np.testing.assert_array_equal(mr, g["mutual_reachability"])
    on_device = E.generate_combined_topics_kl(g["topics"], int(g["min_samples"]), 3, engine=eng)
    on_host = E.generate_combined_topics_kl(g["topics"], int(g["min_samples"]), 3)
    assert on_device.shape == on_host.shape and on_device.shape[0] >= 2
    np.testing.assert_allclose(on_device, on_host, rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("mode", list(MODES))
def test_zero_threshold_with_denormal_products(amd, oracle, mode):
    """e_step_thresh = 0 keeps every positive product, so a responsibility norm can be denormal; the
    reference divides (every quotient <= 1), a plain reciprocal would overflow (plsa.py:98-105)."""
    rs = np.random.RandomState(3)
    n, m, k = 40, 30, 8
    X = sp.random(n, m, density=0.3, format="csr", random_state=rs, dtype=np.float32)
    X.data = np.ceil(X.data * 3).astype(np.float32)
    U0 = rs.rand(n, k).astype(np.float32); V0 = rs.rand(k, m).astype(np.float32)
    U0[:10] *= np.float32(1e-24); V0[:, :8] *= np.float32(1e-20)    # products ~1e-44: denormal
    U0[10:] /= U0[10:].sum(1, keepdims=True)
    V0 /= V0.sum(1, keepdims=True)
    r, c, v = coo_arrays(X)
    ones = np.ones(n, np.float32)
    Uo, Vo = U0.copy(), V0.copy()
    with np.errstate(all="ignore"):
        oracle.plsa_fit_inner(r, c, v, Vo, Uo, ones, n_iter=3, n_iter_per_test=10, tolerance=0.0, e_step_thresh=0.0)
    U, V = U0.copy(), V0.copy()
    amd.plsa_fit_inner(r, c, v, V, U, ones, n_iter=3, n_iter_per_test=10, tolerance=0.0, e_step_thresh=0.0,
                       flags=MODES[mode])
    assert np.all(np.isfinite(U)) and np.all(np.isfinite(V))
    close_factors(U, Uo, tol=2e-4); close_factors(V, Vo, tol=2e-4)


# ------------------------------------------------------------------------------------------------
# the reference-generated fit at which the reference's float32 arithmetic is visibly inexact
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", list(MODES))
def test_big_fit_vs_reference_and_exact_arithmetic(amd, mode):
    """fit_k4_big: 1.5 M non-zeros, k = 4, two iterations, produced by the reference itself (make_golden.py big).
    tests/test_oracle_golden.py shows on the CPU that the strict oracle reproduces it bit for bit and that the
    all-float64 build of the same algorithm is > 1e-4 away from it.  Here: the HIP engine agrees with EXACT
    arithmetic to 1e-5 and its distance to the reference is the reference's own distance to exact arithmetic --
    above ~1e6 non-zeros "within 1e-4 of the reference" is not attainable by anything more accurate than it."""
    from oracle.plsa_oracle import Oracle
    g = load_golden("fit_k4_big")
    X = golden_csr(g)
    kw = dict(n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
              e_step_thresh=float(g["thresh"]), random_state=int(g["fit_seed"]))
    wide = Oracle(variant="wide")
    wide.set_threads(8)
    Uw, Vw, trace_w, _ = wide.plsa_fit(X, int(g["k"]), g["sw"], return_trace=True, **kw)
    U, V, info = amd.plsa_fit(X, int(g["k"]), g["sw"], flags=MODES[mode], return_info=True, **kw)
    assert info["n_iter"] == int(g["iters"])
    ref_vs_exact = peak_rel(g["V"], Vw)
    assert ref_vs_exact > 1e-4                                          # the reference's own error (CPU test)
    assert peak_rel(V, Vw) <= 1e-5 and peak_rel(U, Uw) <= 1e-5, (peak_rel(V, Vw), peak_rel(U, Uw))
    assert elem_rel(V, Vw) <= 1e-4 and elem_rel(U, Uw) <= 1e-4
    assert peak_rel(V, g["V"]) <= 1.5 * ref_vs_exact, (peak_rel(V, g["V"]), ref_vs_exact)
    assert peak_rel(U, g["U"]) <= 1.5 * max(peak_rel(g["U"], Uw), 1e-5)
    np.testing.assert_allclose(info["log_likelihood_trace"], trace_w, rtol=1e-6)
    # the reference's log-likelihood is ONE float32 running sum over 1.5 M terms (plsa.py:322, 375-384): 2e-3 off
    # the exact value here -- 200 times the north-star tolerance -- and HIP is that far from it, no further
    ll = info["log_likelihood_trace"].astype(np.float64)
    ref_ll_err = np.abs(g["ll_trace"].astype(np.float64) - trace_w).max() / np.abs(trace_w).max()
    assert ref_ll_err > 1e-4
    assert np.abs(ll - g["ll_trace"]).max() / np.abs(trace_w).max() <= 1.5 * ref_ll_err


def test_fuzz_slice(amd):
    """A bounded slice (200 cases, ~1 minute) of tests/fuzz_parity.py -- seeded random corpora with odd k, empty /
    heavy rows and columns, thresholds, sample weights, early stopping, refit, under five structure-knob sets and
    both schedules, each against the pinned oracle -- inside the collected suite."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "200", "3"],
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "mismatches 0" in out.stdout
    # round 6: every case also runs once in the reference's rounding and must equal the strict oracle BIT FOR BIT
    assert "200 of 200 fits bit-identical" in out.stdout, out.stdout[-500:]


@pytest.mark.parametrize("per_test", [2, 3, 5])
def test_speculation_across_likelihood_tests_keeps_every_stop_decision(amd, oracle, per_test):
    """The fused loop of a small corpus enqueues the iteration BEHIND a likelihood test before the host has the test's
    value (third set of factor buffers, plsa_fit).  Whatever the test then says -- stop at the very first test, at a later
    one, never -- iteration count, trace and factors must be those of the loop that waits (PLSA_SPECULATE=0), bit for
    bit, and the count must be the oracle's (the reference's plsa.py:630-638 loop)."""
    from enstop_amd.engine import reset_engines
    X = _corpus(500, 700, 0.05, seed=91, empty_rows=1)
    sw = np.ones(X.shape[0], np.float32)
    saved = dict(os.environ)
    try:
        for tol in (0.5, 2e-2, 3e-3, 1e-4, 0.0):             # stop at the first test ... never
            for n_iter in (3, 4, 17):
                kw = dict(n_iter=n_iter, n_iter_per_test=per_test, tolerance=tol, e_step_thresh=1e-16, random_state=7)
                got = {}
                for spec in ("1", "0"):
                    os.environ["PLSA_SPECULATE"] = spec
                    reset_engines()
                    got[spec] = amd.plsa_fit(X, 6, sw, flags=MODES["fused"], return_info=True, **kw)
                (U1, V1, i1), (U0, V0, i0) = got["1"], got["0"]
                assert i1["n_iter"] == i0["n_iter"], kw
                np.testing.assert_array_equal(i1["log_likelihood_trace"], i0["log_likelihood_trace"])
                np.testing.assert_array_equal(U1, U0)
                np.testing.assert_array_equal(V1, V0)
                _, _, _, iters = oracle.plsa_fit(X, 6, sw, return_trace=True, **kw)
                assert i1["n_iter"] == iters, kw
    finally:
        os.environ.clear(); os.environ.update(saved)
        reset_engines()


def test_schedule_knobs_do_not_change_results(amd):
    """Round-3 scheduling machinery changes WHEN work runs, never WHAT is computed: measured XCD boundaries of the
    column pass (PLSA_BALANCE), the event-linked pipelines of small corpora (PLSA_PIPELINE), hipGraph replay
    (PLSA_GRAPH), the address width of the factor-row gathers (PLSA_FORCE_WIDE), and the iteration enqueued ahead of a
    likelihood test's verdict on a third set of buffers (PLSA_SPECULATE) -- factors, iteration count and likelihood trace
    are bit-identical with each switched the other way.
    Two corpora: one large enough for the boundary tuning (nnz * k >= 1e8), one small enough for the pipelines."""
    from enstop_amd.engine import reset_engines
    rs = np.random.RandomState(4)
    big = sp.random(30000, 20000, density=0.005, format="csr", random_state=rs, dtype=np.float32)
    big.data = np.ceil(big.data * 4).astype(np.float32)
    small = sp.random(3000, 2500, density=0.02, format="csr", random_state=rs, dtype=np.float32)
    small.data = np.ceil(small.data * 4).astype(np.float32)
    cases = [(big, 64, dict(n_iter=7, n_iter_per_test=3, tolerance=0.0, random_state=2)),
             (small, 20, dict(n_iter=23, n_iter_per_test=4, tolerance=1e-7, random_state=5))]
    saved = dict(os.environ)
    try:
        ref = []
        reset_engines()
        for X, k, kw in cases:
            ref.append(amd.plsa_fit(X, k, np.ones(X.shape[0], np.float32), return_info=True, **kw))
        assert amd.engine.get_engine().balance_info()["timed_launches"] >= 0
        # PLSA_FORCE_WIDE: the 64-bit row addressing that factor tables of 4 GB or more take (32-bit byte offsets below)
        for knob, val in (("PLSA_BALANCE", "0"), ("PLSA_BALANCE", "1"), ("PLSA_PIPELINE", "0"), ("PLSA_GRAPH", "1"),
                          ("PLSA_FORCE_WIDE", "1"), ("PLSA_SPECULATE", "0"), ("PLSA_SPECULATE", "1")):
            os.environ[knob] = val
            reset_engines()                       # knobs are read when a context is created
            for (X, k, kw), (U0, V0, i0) in zip(cases, ref):
                U, V, info = amd.plsa_fit(X, k, np.ones(X.shape[0], np.float32), return_info=True, **kw)
                assert info["n_iter"] == i0["n_iter"], (knob, val)
                np.testing.assert_array_equal(info["log_likelihood_trace"], i0["log_likelihood_trace"])
                np.testing.assert_array_equal(U, U0, err_msg="%s=%s" % (knob, val))
                np.testing.assert_array_equal(V, V0, err_msg="%s=%s" % (knob, val))
            del os.environ[knob]
        # PLSA_ROW_XCD (round-5 experiment knob): XCD x walks the x-th eighth of the documents.  Rows are owned by one
        # group whatever the schedule -> factors bit-identical; the likelihood's per-workgroup partials are added in
        # another order -> equal to float64 rounding
        os.environ["PLSA_ROW_XCD"] = "1"
        reset_engines()
        for (X, k, kw), (U0, V0, i0) in zip(cases, ref):
            U, V, info = amd.plsa_fit(X, k, np.ones(X.shape[0], np.float32), return_info=True, **kw)
            assert info["n_iter"] == i0["n_iter"]
            np.testing.assert_allclose(info["log_likelihood_trace"], i0["log_likelihood_trace"], rtol=1e-6)
            np.testing.assert_array_equal(U, U0); np.testing.assert_array_equal(V, V0)
    finally:
        os.environ.clear(); os.environ.update(saved)
        reset_engines()
