"""PLSA_REFERENCE_SUMS / PLSA_REFERENCE_LL (`arithmetic="reference"` / `"reference_source"`): the HIP path with the
REFERENCE'S roundings -- every sum one float32 accumulator added in the order the reference's loops add (plsa.py:96-105,
182-194; enstop_amd/csrc/plsa_ref_kernels.hpp).  Needs a real MI355X: -m gpu.

The fixtures under tests/golden/ were produced by the reference's own source (tests/golden/make_golden.py), so in this
mode the comparison is not a tolerance: P(z|w,d), both factors and both norm vectors must be the reference's BITS.  Only the
log-likelihood keeps a tolerance -- its float32 logarithm differs in the last place between NumPy, glibc, numba and the
device.  The BASELINE-sized legs (config 1 against the numba-compiled reference, configs 2 / 3 against the strict oracle)
live in tests/test_parity_at_scale.py.
"""
import numpy as np
import pytest

from conftest import load_golden, golden_csr, coo_arrays

pytestmark = pytest.mark.gpu

KERNEL_CASES = ["kernels_k6", "kernels_k8_thresh", "kernels_k20", "kernels_k33"]
FIT_CASES = ["fit_k8_tol0", "fit_k5_earlystop", "fit_k4_weighted", "fit_k8_thresh",
             "fit_k6_tupleinit", "fit_k16_mid", "fit_k20_50it", "fit_k4_big"]


@pytest.fixture(scope="module")
def amd():
    import enstop_amd
    return enstop_amd


def same_bits(a, b, what):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    diff = a.view(np.uint32) != b.view(np.uint32)
    # +0.0 / -0.0 cannot occur (all sums are of non-negative terms), so the bit patterns must agree outright
    assert not diff.any(), "%s: %d of %d entries differ, first at %s: %r vs %r (max abs %.3e)" % (
        what, int(diff.sum()), diff.size, np.argwhere(diff)[0].tolist(), a[tuple(np.argwhere(diff)[0])],
        b[tuple(np.argwhere(diff)[0])], float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()))


@pytest.mark.parametrize("case", KERNEL_CASES)
def test_e_step_bits(amd, case):
    """plsa.py:91-105: the norm is ONE float32 sum over the topics in order, the quotient a true division."""
    g = load_golden(case)
    r, c, v = coo_arrays(golden_csr(g))
    P = np.full_like(g["P"], -7.0)
    amd.plsa_e_step(r, c, v, g["V"].copy(), g["U"].copy(), P, g["thresh"], arithmetic="reference")
    same_bits(P, g["P"], "P(z|w,d)")


@pytest.mark.parametrize("case", KERNEL_CASES)
def test_m_step_bits(amd, case):
    """plsa.py:182-204, 287-310, 801-814: factors AND both norm vectors, plain / weighted / refit."""
    g = load_golden(case)
    r, c, v = coo_arrays(golden_csr(g))
    n, k = g["U"].shape
    V, U = g["V"].copy(), g["U"].copy()
    a, b = np.zeros(k, np.float32), np.zeros(n, np.float32)
    amd.plsa_m_step(r, c, v, V, U, g["P"], a, b, arithmetic="reference")
    same_bits(a, g["norm_pwz"], "norm_pwz"); same_bits(b, g["norm_pdz"], "norm_pdz")
    same_bits(V, g["V_m"], "P(w|z)"); same_bits(U, g["U_m"], "P(z|d)")

    V, U = g["V"].copy(), g["U"].copy()
    amd.plsa_m_step_w_sample_weight(r, c, v, V, U, g["P"], g["sw"], a, b, arithmetic="reference")
    same_bits(a, g["norm_pwz_w"], "weighted norm_pwz"); same_bits(b, g["norm_pdz_w"], "weighted norm_pdz")
    same_bits(V, g["V_mw"], "weighted P(w|z)"); same_bits(U, g["U_mw"], "weighted P(z|d)")

    U = g["U"].copy()
    Vfixed = g["V"].copy()
    amd.plsa_refit_m_step(r, c, v, Vfixed, U, g["P"], np.ones(n, np.float32), b, arithmetic="reference")
    same_bits(U, g["U_refit"], "refit P(z|d)"); same_bits(b, g["norm_pdz_refit"], "refit norm_pdz")
    np.testing.assert_array_equal(Vfixed, g["V"])


@pytest.mark.parametrize("case", KERNEL_CASES)
def test_sequential_log_likelihood(amd, case):
    """plsa.py:372-384 as one float32 running sum: equal to the reference's value up to the last place of the float32
    logarithm (a few ulp of the total on these small problems)."""
    g = load_golden(case)
    r, c, v = coo_arrays(golden_csr(g))
    ones = np.ones(g["U"].shape[0], np.float32)
    for sw, key, VV, UU in ((ones, "ll_ones", g["V"], g["U"]), (g["sw"], "ll_sw", g["V"], g["U"]),
                            (ones, "ll_after_m", g["V_m"], g["U_m"])):
        got = amd.log_likelihood(r, c, v, VV, UU, sw, arithmetic="reference_source")
        assert got.dtype == np.float32
        if np.isfinite(g[key]):
            assert abs(float(got) - float(g[key])) <= 2e-6 * abs(float(g[key])), (key, got, g[key])
        else:
            assert float(got) == float(g[key]) or (np.isnan(got) and np.isnan(g[key]))


@pytest.mark.parametrize("arithmetic", ["reference", "reference_source"])
@pytest.mark.parametrize("case", FIT_CASES)
def test_fit_bits(amd, case, arithmetic):
    """The reference's own `plsa_fit` outputs (5 ... 71 iterations, weights, an in-range threshold, early stops, a 1.5 M
    non-zero corpus): iteration count equal, factors BIT FOR BIT -- with either log-likelihood (the stop decisions of these
    fixtures do not sit on the float32 sum's error)."""
    g = load_golden(case)
    X = golden_csr(g)
    init = (g["U_init"], g["V_init"]) if "U_init" in g else "random"
    for flags in (0, amd.PLSA_FUSED):            # PLSA_FUSED is ignored in this mode: one arithmetic, one kernel sequence
        U, V, info = amd.plsa_fit(X, int(g["k"]), g["sw"], init=init, n_iter=int(g["n_iter"]),
                                  n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
                                  e_step_thresh=float(g["thresh"]), random_state=int(g["fit_seed"]),
                                  flags=flags, return_info=True, arithmetic=arithmetic)
        assert info["n_iter"] == int(g["iters"])
        same_bits(U, g["U"], case + " P(z|d)")
        if "V" in g:
            same_bits(V, g["V"], case + " P(w|z)")
        else:                                    # big fixtures keep a column sample
            same_bits(V[:, g["V_cols"]], g["V_sample"], case + " P(w|z) sample")
        # the reference's log-likelihood is ONE float32 running sum (plsa.py:322): "reference_source" reproduces it to the
        # last place of the float32 logarithm at any size; the float64-accumulated one of "reference" is 2e-3 away from it on
        # the 1.5 M non-zeros of fit_k4_big (the reference's own error) and within 1e-5 on the small fixtures
        tr, ref = np.asarray(info["log_likelihood_trace"], np.float64), np.asarray(g["ll_trace"], np.float64)
        n_cmp = min(len(tr), len(ref))
        fin = np.isfinite(ref[:n_cmp])
        tol = 2e-6 if arithmetic == "reference_source" else 1e-5
        if arithmetic == "reference_source" or case != "fit_k4_big":
            assert np.all(np.abs(tr[:n_cmp][fin] - ref[:n_cmp][fin]) <= tol * np.abs(ref[:n_cmp][fin])), (tr, ref)


@pytest.mark.parametrize("case", ["refit_k6", "refit_k8_weighted"])
def test_refit_bits(amd, case):
    g = load_golden(case)
    X = golden_csr(g)
    for arithmetic in ("reference", "reference_source"):
        U, info = amd.plsa_refit(X, g["topics"], g["sw"], n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]),
                                 tolerance=float(g["tol"]), random_state=np.random.RandomState(42), return_info=True,
                                 arithmetic=arithmetic)
        assert info["n_iter"] == int(g["iters"])
        same_bits(U, g["U"], case)


def test_estimator_keyword_and_engine_state(amd):
    """`PLSA(arithmetic="reference")` reaches the kernels; a kernel-level call in the reference arithmetic leaves the shared
    engine in its default arithmetic (the next default fit is the engine's own, float64-normed one)."""
    g = load_golden("fit_k16_mid")
    X = golden_csr(g)
    r, c, v = coo_arrays(golden_csr(load_golden("kernels_k6")))
    gk = load_golden("kernels_k6")
    amd.plsa_e_step(r, c, v, gk["V"].copy(), gk["U"].copy(), np.zeros_like(gk["P"]), gk["thresh"], arithmetic="reference")
    kw = dict(n_iter=int(g["n_iter"]), n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
              e_step_thresh=float(g["thresh"]), random_state=int(g["fit_seed"]))
    Ud, Vd = amd.plsa_fit(X, int(g["k"]), g["sw"], flags=0, **kw)
    Ur, Vr = amd.plsa_fit(X, int(g["k"]), g["sw"], arithmetic="reference", **kw)
    same_bits(Ur, g["U"], "reference arithmetic")
    assert np.any(Ud.view(np.uint32) != Ur.view(np.uint32)), "the default fit ran in the reference arithmetic"
    est = amd.PLSA(n_components=int(g["k"]), arithmetic="reference", **kw).fit(X.astype(np.int64))
    same_bits(est.embedding_, g["U"], "estimator")
    same_bits(est.components_, g["V"], "estimator")


def test_reference_mode_refuses_sharding(amd):
    g = load_golden("fit_k16_mid")
    X = golden_csr(g)
    with amd.Engine() as eng:
        eng.upload_csr(X)
        eng.init_factors_device(int(g["k"]), 1)
        eng.set_arithmetic("reference")
        with pytest.raises(amd.DeviceError, match="doc-sharded"):
            eng.em_accumulate()
        eng.set_arithmetic(None)
        eng.em_accumulate()


def test_ensemble_members_bits(amd):
    """enstop_.py:84-115 (a bootstrapped member) and :220-231 (the serial ensemble sharing one RandomState) in the reference's
    rounding: the members' topic matrices are the reference's bits -- device bootstrap gather, device MT19937 initialisation and
    the reference-arithmetic EM kernels in one chain."""
    g = load_golden("member_k6")
    X = golden_csr(g)
    k = int(g["k"])
    kw = dict(n_iter=int(g["n_iter"]), n_iter_per_test=10, tolerance=0.0, e_step_thresh=float(g["thresh"]), arithmetic="reference")
    same_bits(amd.plsa_topics(X, k, random_state=np.random.RandomState(5), **kw), g["V_rs5"], "member, RandomState(5)")
    same_bits(amd.plsa_topics(X, k, random_state=9, **kw), g["V_int9"], "member, seed 9")
    same_bits(amd.plsa_topics(X, k, random_state=9, bootstrap=False, **kw), g["V_nobootstrap"], "member without bootstrap")
    stack = amd.ensemble_of_topics(X, k, n_runs=3, parallelism="none", random_state=np.random.RandomState(21), **kw)
    same_bits(stack, g["V_stack_rs21"], "serial ensemble of three")


@pytest.mark.parametrize("k", [1, 3, 65, 128, 130, 300, 520, 1000])
def test_wide_topic_counts_bits(amd, k):
    """Topic counts beyond one wave (k > 64: two to sixteen topics per lane in the chains and passes), k = 1 and an odd k,
    with document weights and e_step_thresh = 0 (denormal norms: the true division needs no rescue) -- against the strict
    oracle, bit for bit, kernel level and a short weighted fit."""
    import scipy.sparse as sp
    from oracle.plsa_oracle import Oracle
    o = Oracle(variant="strict")
    o.set_threads(4)
    o.set_ll_sequential(True)
    rs = np.random.RandomState(100 + k)
    n, m = 70, 90
    X = sp.random(n, m, density=0.15, format="csr", random_state=rs, dtype=np.float64)
    X.data = np.ceil(X.data * 4)
    X = X.astype(np.float32)
    r, c, v = coo_arrays(X)
    U0 = rs.rand(n, k); U0 /= U0.sum(1, keepdims=True)
    V0 = rs.rand(k, m); V0 /= V0.sum(1, keepdims=True)
    U0 = U0.astype(np.float32); V0 = V0.astype(np.float32)
    sw = (0.25 + rs.rand(n)).astype(np.float32)
    for thresh in (np.float32(0.0), np.float32(1e-32)):
        Po = np.zeros((X.nnz, k), np.float32)
        o.plsa_e_step(r, c, v, V0, U0, Po, thresh)
        P = np.zeros_like(Po)
        amd.plsa_e_step(r, c, v, V0.copy(), U0.copy(), P, thresh, arithmetic="reference")
        same_bits(P, Po, "P(z|w,d), thresh %g" % thresh)
    Vo, Uo = V0.copy(), U0.copy()
    nwo, ndo = np.zeros(k, np.float32), np.zeros(n, np.float32)
    o.plsa_m_step_w_sample_weight(r, c, v, Vo, Uo, Po, sw, nwo, ndo)
    V, U = V0.copy(), U0.copy()
    nw, nd = np.zeros(k, np.float32), np.zeros(n, np.float32)
    amd.plsa_m_step_w_sample_weight(r, c, v, V, U, Po, sw, nw, nd, arithmetic="reference")
    same_bits(nw, nwo, "norm_pwz"); same_bits(nd, ndo, "norm_pdz"); same_bits(V, Vo, "P(w|z)"); same_bits(U, Uo, "P(z|d)")
    ll_o = o.log_likelihood(r, c, v, Vo, Uo, sw)
    ll = amd.log_likelihood(r, c, v, V, U, sw, arithmetic="reference_source")
    assert abs(float(ll) - float(ll_o)) <= 2e-6 * abs(float(ll_o)), (ll, ll_o)
    kw = dict(n_iter=4, n_iter_per_test=2, tolerance=0.0, e_step_thresh=0.0, random_state=5)
    Uf_o, Vf_o, _, it_o = o.plsa_fit(X, k, sw, return_trace=True, **kw)
    Uf, Vf, info = amd.plsa_fit(X, k, sw, return_info=True, arithmetic="reference", **kw)
    assert info["n_iter"] == it_o
    same_bits(Uf, Uf_o, "fit P(z|d)"); same_bits(Vf, Vf_o, "fit P(w|z)")


STREAMFIT_CASES = ["streamfit_k6", "streamfit_k4_weighted", "streamfit_k5_earlystop", "streamfit_k8_thresh",
                   "streamfit_k1_zero_change"]


@pytest.mark.parametrize("case", STREAMFIT_CASES)
def test_streamed_module_fit_bits(amd, case):
    """enstop/streamed_plsa.py (blocks of non-zeros through double-buffered factors, :341-391) adds in plsa.py's order -- its
    fixtures, generated by running THAT module, are met bit for bit by the reference arithmetic under its stop rule (no
    `change == 0` arm, :596-597), in-range threshold and weights included."""
    from enstop_amd.streamed_plsa import plsa_fit as streamed_fit
    g = load_golden(case)
    X = golden_csr(g)
    U, V, info = streamed_fit(X, int(g["k"]), g["sw"], block_size=int(g["block_size"]), n_iter=int(g["n_iter"]),
                              n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
                              e_step_thresh=float(g["thresh"]), random_state=int(g["fit_seed"]), return_info=True,
                              arithmetic="reference_source")
    assert info["n_iter"] == int(g["iters"])
    same_bits(U, g["U"], case + " P(z|d)"); same_bits(V, g["V"], case + " P(w|z)")


@pytest.mark.parametrize("case", ["streamrefit_k6", "streamrefit_k8_weighted_thresh"])
def test_streamed_module_refit_bits(amd, case):
    from enstop_amd.streamed_plsa import plsa_refit as streamed_refit
    g = load_golden(case)
    X = golden_csr(g)
    U, info = streamed_refit(X, g["topics"], g["sw"], block_size=int(g["block_size"]), n_iter=int(g["n_iter"]),
                             n_iter_per_test=int(g["n_iter_per_test"]), tolerance=float(g["tol"]),
                             e_step_thresh=float(g["thresh"]), random_state=np.random.RandomState(42), return_info=True,
                             arithmetic="reference")
    assert info["n_iter"] == int(g["iters"])
    same_bits(U, g["U"], case)


@pytest.mark.parametrize("case", ["estimator_int", "estimator_float", "estimator_int_emptyrows"])
def test_estimator_bits(amd, case):
    """`PLSA.fit_transform` / `transform` as the reference's estimator runs them (plsa.py:1117-1220: validation, float input
    L1-row-normalised, empty documents dropped and restored as float64 zeros, transform = 50 refit iterations from seed 42):
    with `arithmetic="reference"` the fitted attributes and the transformed rows are the reference's bits."""
    import scipy.sparse as sp
    g = load_golden(case)
    shape = tuple(int(s) for s in g["shape"])
    X = sp.csr_matrix((g["data"], g["indices"], g["indptr"]), shape=shape)
    model = amd.PLSA(n_components=int(g["k"]), n_iter=30, n_iter_per_test=10, tolerance=0.0, random_state=11,
                     arithmetic="reference")
    emb = model.fit_transform(X)
    assert str(np.asarray(emb).dtype) == str(g["embedding_dtype"])
    assert np.array_equal(np.asarray(emb, np.float64), np.asarray(g["embedding"], np.float64)), \
        np.abs(np.asarray(emb, np.float64) - g["embedding"]).max()
    same_bits(model.components_, g["components"], case + " components_")
    Xt = sp.csr_matrix((g["t_data"], g["t_indices"], g["t_indptr"]), shape=tuple(int(s) for s in g["t_shape"]))
    same_bits(model.transform(Xt), g["transformed"], case + " transform")


@pytest.mark.parametrize("k", [8, 20, 70])
def test_norm_chain_from_parity_pairs_on_tie_heavy_input(amd, k, monkeypatch):
    """norm_pwz[z] (plsa.py:193, ONE float32 running sum over all non-zeros) evaluated from per-chunk (parity -> increment) pairs
    (k_ref_pair_*) against the serial chain (k_ref_norm_chain) and the oracle on input built to hit the hard cases: dyadic
    responsibilities and small integer counts (addends with few significant bits: exact ties in round-to-nearest-even all along
    the chain), document weights that are powers of two, stretches of zeros, one huge addend that jumps several binades, and
    enough non-zeros for ~15 binade crossings per topic.  Same bits three ways; the walk reports how many chunks it had to add
    addend by addend."""
    import scipy.sparse as sp
    from oracle.plsa_oracle import Oracle
    rs = np.random.RandomState(k)
    n, m = 4000, 300
    X = sp.random(n, m, density=0.08, format="csr", random_state=rs, dtype=np.float64)
    X.data = rs.randint(1, 5, size=X.nnz).astype(np.float64)
    X = X.astype(np.float32)
    X.data[X.nnz // 2] = 3.0e6                                   # one addend several binades above the running sums
    r, c, v = coo_arrays(X)
    P = (rs.randint(0, 9, size=(X.nnz, k)) / np.float32(16.0)).astype(np.float32)     # 0, 1/16, ..., 1/2: dyadic
    P[1000:3000] = 0.0                                           # a stretch of + 0.0 addends
    sw = (2.0 ** rs.randint(-2, 3, size=n)).astype(np.float32)
    U0 = np.full((n, k), 1.0 / k, np.float32); V0 = np.full((k, m), 1.0 / m, np.float32)
    o = Oracle(variant="strict")
    want = {}
    for weighted in (False, True):
        Vo, Uo = V0.copy(), U0.copy()
        nw, nd = np.zeros(k, np.float32), np.zeros(n, np.float32)
        if weighted:
            o.plsa_m_step_w_sample_weight(r, c, v, Vo, Uo, P, sw, nw, nd)
        else:
            o.plsa_m_step(r, c, v, Vo, Uo, P, nw, nd)
        want[weighted] = (nw, Vo, Uo)
    for mode in ("pairs", "serial"):
        monkeypatch.setenv("PLSA_REF_CHAIN", mode)
        with amd.Engine() as eng:                                # the knob is read when a context is created
            eng.upload_csr(X)
            eng.set_arithmetic("reference")
            for weighted in (False, True):
                eng.set_factors(U0, V0)
                eng.set_p(P)
                nw, nd = eng.m_step(sw if weighted else None)
                U, V = eng.get_factors()
                same_bits(nw, want[weighted][0], "norm_pwz, %s, weighted=%s" % (mode, weighted))
                same_bits(V, want[weighted][1], "P(w|z), %s" % mode); same_bits(U, want[weighted][2], "P(z|d), %s" % mode)
            info = eng.reference_chain_info()
            if mode == "pairs":
                groups = (k + 63) // 64                                  # one walking wave per 64 topics, each counts its chunks
                # (PAIR_L = 256 addends per chunk; ~15 binade crossings per topic and run, in a few hundred chunks)
                assert info["chunks"] == 2 * groups * ((X.nnz + 255) // 256) and 0 < info["slow_chunks"] < info["chunks"] // 2, info
            else:
                assert info["chunks"] == 0 and info["serial_chain_now"], info


def test_norm_chain_pairs_with_negative_and_non_finite_addends(amd, monkeypatch):
    """What the pair path must hand to plain additions: NEGATIVE addends (sample weights are any float32 to the reference,
    plsa.py:294: the running sum goes down, through zero, below it), an infinite one and a NaN.  With negative weights: pairs =
    serial = oracle bit for bit; with the non-finite ones pairs = serial (NaN == NaN: the payload of a NaN the host's adder
    makes is not the GPU's)."""
    import scipy.sparse as sp
    from oracle.plsa_oracle import Oracle
    rs = np.random.RandomState(5)
    n, m, k = 3000, 200, 12
    X = sp.random(n, m, density=0.1, format="csr", random_state=rs, dtype=np.float64)
    X.data = rs.randint(1, 4, size=X.nnz).astype(np.float64)
    X = X.astype(np.float32)
    r, c, v = coo_arrays(X)
    P = rs.rand(X.nnz, k).astype(np.float32)
    U0 = np.full((n, k), 1.0 / k, np.float32); V0 = np.full((k, m), 1.0 / m, np.float32)
    sw_neg = rs.randn(n).astype(np.float32)                       # about half of the documents weigh negative
    sw_neg[n // 2:] *= 3.0                                         # ... and the second half pulls the sums back through zero
    sw_bad = np.ones(n, np.float32); sw_bad[n // 3] = np.inf; sw_bad[2 * n // 3] = np.nan
    nw_o, nd_o = np.zeros(k, np.float32), np.zeros(n, np.float32)
    Vo, Uo = V0.copy(), U0.copy()
    Oracle(variant="strict").plsa_m_step_w_sample_weight(r, c, v, Vo, Uo, P, sw_neg, nw_o, nd_o)
    got = {}
    for mode in ("pairs", "serial"):
        monkeypatch.setenv("PLSA_REF_CHAIN", mode)
        with amd.Engine() as eng:
            eng.upload_csr(X)
            eng.set_arithmetic("reference")
            for name, sw in (("negative", sw_neg), ("non-finite", sw_bad)):
                eng.set_factors(U0, V0)
                eng.set_p(P)
                got[mode, name] = eng.m_step(sw)[0].copy()
    same_bits(got["pairs", "negative"], nw_o, "norm_pwz, negative weights, pairs against the oracle")
    same_bits(got["serial", "negative"], nw_o, "norm_pwz, negative weights, serial against the oracle")
    assert np.array_equal(got["pairs", "non-finite"], got["serial", "non-finite"], equal_nan=True)
    assert np.isnan(got["pairs", "non-finite"]).all()


def test_sequential_likelihood_from_parity_pairs(amd, monkeypatch):
    """plsa.py:322 / :384, ONE float32 running sum of x * log p over the non-zeros, evaluated as the chain of the NEGATED terms from
    parity pairs against the serial chain (k_ref_ll_chain) on the same terms: the same float32, bit for bit -- with normalised
    factors (every term <= 0), with topics that sum to more than one (p > 1: positive terms, chunks that must go the slow way), with
    negative sample weights (terms of both signs) and with a likelihood that is exactly zero."""
    import struct
    import scipy.sparse as sp
    rs = np.random.RandomState(11)
    n, m, k = 5000, 400, 16
    X = sp.random(n, m, density=0.06, format="csr", random_state=rs, dtype=np.float64)
    X.data = rs.randint(1, 6, size=X.nnz).astype(np.float64)
    X = X.astype(np.float32)
    U = rs.rand(n, k); U /= U.sum(1, keepdims=True)
    V = rs.rand(k, m); V /= V.sum(1, keepdims=True)
    U = U.astype(np.float32); V = V.astype(np.float32)
    V_big = (V * np.float32(m * 0.9)).astype(np.float32)           # p(w|d) ~ 0.9 on average: about half of the terms positive
    sw_neg = rs.randn(n).astype(np.float32)
    X1 = sp.csr_matrix(np.ones((4200, 1), np.float32))              # one word: p = 1, every term + 0.0
    cases = [("normalised", X, U, V, None), ("weighted", X, U, V, (0.5 + rs.rand(n)).astype(np.float32)),
             ("p > 1", X, U, V_big, None), ("negative weights", X, U, V, sw_neg),
             ("all zero", X1, np.ones((4200, 1), np.float32), np.ones((1, 1), np.float32), None)]
    got = {}
    for mode in ("pairs", "serial"):
        monkeypatch.setenv("PLSA_REF_CHAIN", mode)
        for name, Xc, Uc, Vc, sw in cases:
            with amd.Engine() as eng:
                eng.upload_csr(Xc)
                eng.set_arithmetic("reference_source")
                eng.set_factors(Uc, Vc)
                got[mode, name] = np.float32(eng.log_likelihood(sw))
    for name, *_ in cases:
        a, b = got["pairs", name], got["serial", name]
        assert struct.pack("f", a) == struct.pack("f", b), (name, a, b)
    assert got["pairs", "normalised"] < 0 and struct.pack("f", got["pairs", "all zero"]) == struct.pack("f", np.float32(0.0))


def test_norm_chain_goes_back_to_the_serial_chain_when_the_walk_keeps_failing(amd, monkeypatch):
    """PLSA_REF_CHAIN unset (auto): a corpus of ~20 chunks whose 64 topics cross a binade in almost every chunk -- the first walk
    reports most of its chunks on the slow way, the context then uses the serial chain; the bits are the oracle's every time."""
    import scipy.sparse as sp
    from oracle.plsa_oracle import Oracle
    monkeypatch.delenv("PLSA_REF_CHAIN", raising=False)
    rs = np.random.RandomState(3)
    n, m, k = 400, 120, 64
    X = sp.random(n, m, density=0.11, format="csr", random_state=rs, dtype=np.float64)
    X.data = rs.randint(1, 4, size=X.nnz).astype(np.float64)
    X = X.astype(np.float32)
    assert X.nnz >= 4096
    r, c, v = coo_arrays(X)
    P = rs.rand(X.nnz, k).astype(np.float32); P /= P.sum(1, keepdims=True)
    U0 = np.full((n, k), 1.0 / k, np.float32); V0 = np.full((k, m), 1.0 / m, np.float32)
    nw_o, nd_o = np.zeros(k, np.float32), np.zeros(n, np.float32)
    Vo, Uo = V0.copy(), U0.copy()
    Oracle(variant="strict").plsa_m_step(r, c, v, Vo, Uo, P, nw_o, nd_o)
    with amd.Engine() as eng:
        eng.upload_csr(X)
        eng.set_arithmetic("reference")
        for _ in range(4):
            eng.set_factors(U0, V0)
            eng.set_p(P)
            nw, _ = eng.m_step(None)
            eng.synchronize()
            same_bits(nw, nw_o, "norm_pwz")
        info = eng.reference_chain_info()
        assert info["serial_chain_now"] and 4 * info["slow_chunks"] > info["chunks"] > 0, info


@pytest.mark.parametrize("k", [8, 70, 200])
def test_long_columns_through_the_chain_kernel(amd, k, monkeypatch):
    """plsa.py:190 / :296: the vocabulary half of the M-step with the long columns handed to k_ref_norm_chain<GATHER> (one
    workgroup per column, the entries' rows of P found through their COO positions) -- here from 16 entries on, so that most
    columns of a small corpus go that way, one of them holding every document: Vacc, and with it P(w|z), bit for bit the oracle's,
    with and without sample weights."""
    import scipy.sparse as sp
    from oracle.plsa_oracle import Oracle
    monkeypatch.setenv("PLSA_REF_HEAVY_MIN", "16")
    rs = np.random.RandomState(100 + k)
    n, m = 1500, 90
    X = sp.random(n, m, density=0.05, format="lil", random_state=rs, dtype=np.float64)
    X[:, 7] = 1.0                                                  # a word in every document: 1 500 entries, 12 tiles of the chain
    X[::3, 11] = 2.0
    X = X.tocsr(); X.data = np.ceil(X.data * 3); X = X.astype(np.float32)
    r, c, v = coo_arrays(X)
    P = rs.rand(X.nnz, k).astype(np.float32); P /= P.sum(1, keepdims=True)
    sw = (0.25 + rs.rand(n)).astype(np.float32)
    U0 = np.full((n, k), 1.0 / k, np.float32); V0 = np.full((k, m), 1.0 / m, np.float32)
    o = Oracle(variant="strict")
    with amd.Engine() as eng:
        eng.upload_csr(X)
        eng.set_arithmetic("reference")
        for weights in (None, sw):
            Vo, Uo = V0.copy(), U0.copy()
            nw_o, nd_o = np.zeros(k, np.float32), np.zeros(n, np.float32)
            if weights is None:
                o.plsa_m_step(r, c, v, Vo, Uo, P, nw_o, nd_o)
            else:
                o.plsa_m_step_w_sample_weight(r, c, v, Vo, Uo, P, weights, nw_o, nd_o)
            eng.set_factors(U0, V0)
            eng.set_p(P)
            nw, nd = eng.m_step(weights)
            U, V = eng.get_factors()
            same_bits(nw, nw_o, "norm_pwz"); same_bits(V, Vo, "P(w|z)"); same_bits(U, Uo, "P(z|d)")


@pytest.mark.parametrize("k", [5, 64, 130])
def test_tiled_document_pass(amd, k, monkeypatch):
    """plsa.py:188-202 through k_ref_row_pass_tiled (the form large corpora take: a wave per 64 / NZ documents, norm_pdz one lane
    per document over an LDS tile), forced onto a small corpus with empty documents, one very long document and ragged lengths
    inside every tile: P(z|d) and norm_pdz bit for bit the oracle's, M-step and refit M-step, with and without weights."""
    import scipy.sparse as sp
    from oracle.plsa_oracle import Oracle
    monkeypatch.setenv("PLSA_REF_ROW_TILED", "1")
    rs = np.random.RandomState(300 + k)
    n, m = 777, 210
    X = sp.random(n, m, density=0.04, format="lil", random_state=rs, dtype=np.float64)
    X[5, :] = 1.0                                                  # one document with every word
    X[9, :] = 0.0; X[n - 1, :] = 0.0                               # empty documents, the last one included
    X = X.tocsr(); X.data = np.ceil(X.data * 3); X.eliminate_zeros(); X = X.astype(np.float32)
    r, c, v = coo_arrays(X)
    P = rs.rand(X.nnz, k).astype(np.float32); P /= P.sum(1, keepdims=True)
    sw = (0.25 + rs.rand(n)).astype(np.float32)
    U0 = np.full((n, k), 1.0 / k, np.float32); V0 = np.full((k, m), 1.0 / m, np.float32)
    o = Oracle(variant="strict")
    with amd.Engine() as eng:
        eng.upload_csr(X)
        eng.set_arithmetic("reference")
        for weights in (None, sw):
            for update_v in (True, False):
                Vo, Uo = V0.copy(), U0.copy()
                nw_o, nd_o = np.zeros(k, np.float32), np.zeros(n, np.float32)
                if not update_v:
                    o.plsa_refit_m_step(r, c, v, Vo, Uo, P, np.ones(n, np.float32) if weights is None else weights, nd_o)
                elif weights is None:
                    o.plsa_m_step(r, c, v, Vo, Uo, P, nw_o, nd_o)
                else:
                    o.plsa_m_step_w_sample_weight(r, c, v, Vo, Uo, P, weights, nw_o, nd_o)
                eng.set_factors(U0, V0)
                eng.set_p(P)
                nw, nd = eng.m_step(weights, update_v=update_v)
                U, V = eng.get_factors()
                same_bits(nd, nd_o, "norm_pdz"); same_bits(U, Uo, "P(z|d)"); same_bits(V, Vo, "P(w|z)")
