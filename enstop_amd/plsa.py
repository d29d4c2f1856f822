"""pLSA on MI355X behind the reference's own interface (enstop/plsa.py).

Same function names, argument order, defaults and return conventions as the reference module, so
existing callers switch by changing the import.  All numerical work runs in hand-written HIP
kernels through the C ABI (include/plsa_hip.h); there is no CPU code path -- importing works
anywhere, calling anything needs libplsa_hip.so and a gfx950 device.

  reference function (enstop/plsa.py)        here
  plsa_e_step                  :39          plsa_e_step            -> plsa_e_step (C ABI)
  plsa_m_step                  :124         plsa_m_step            -> plsa_m_step(sw=NULL)
  plsa_m_step_w_sample_weight  :221         plsa_m_step_w_sample_weight -> plsa_m_step(sw)
  log_likelihood               :329         log_likelihood         -> plsa_log_likelihood
  plsa_init                    :412         plsa_init (host NumPy, like the reference)
  plsa_fit_inner / plsa_fit    :517 / :643  plsa_fit_inner / plsa_fit -> plsa_fit (C ABI)
  plsa_refit_m_step            :746         plsa_refit_m_step      -> plsa_m_step(update_v=0)
  plsa_refit_inner / plsa_refit :820 / :923 plsa_refit_inner / plsa_refit -> plsa_refit (C ABI)
  PLSA                         :1000        PLSA
"""
import os

import numpy as np
from scipy.sparse import csr_matrix, issparse
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.utils import check_array, check_random_state

from .engine import PLSA_SW_LL_ONLY, PLSA_STOP_NO_ZERO_ARM, arithmetic_flags, default_flags, get_engine
from .utils import (_check_sample_weight, coherence, log_lift, mean_coherence, mean_log_lift, normalize,
                    standardize_input)


# ------------------------------------------------------------------------------------------------
# COO triplets (the reference kernels' argument form) -> CSR on the device
# ------------------------------------------------------------------------------------------------
def _coo_to_csr(X_rows, X_cols, X_vals, n, m):
    """Returns (csr, order): `order` is None when the triplets are already row-major sorted (what
    X.tocoo() of a canonical CSR gives, plsa.py:714), else the stable permutation applied."""
    rows = np.asarray(X_rows)
    order = None
    if rows.size > 1 and np.any(rows[1:] < rows[:-1]):
        order = np.argsort(rows, kind="stable")
        rows = rows[order]
        X_cols = np.asarray(X_cols)[order]
        X_vals = np.asarray(X_vals)[order]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows.astype(np.int64), minlength=n), out=indptr[1:])
    indptr = indptr.astype(np.int32)
    csr = csr_matrix((np.asarray(X_vals, np.float32), np.asarray(X_cols, np.int32), indptr), shape=(n, m))
    return csr, order


def _locked(fn):
    """Serialise calls that use the process-wide engine of a device (thread pools of callers)."""
    import functools
    import inspect
    sig = inspect.signature(fn)
    has_kwargs = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in sig.parameters.values())

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if has_kwargs and "device" not in sig.parameters:
            device = kwargs.get("device", None)
        else:
            device = sig.bind_partial(*args, **kwargs).arguments.get("device", None)
        with get_engine(device).lock:
            return fn(*args, **kwargs)
    return wrapper


class _KernelArithmetic:
    """Arithmetic of ONE kernel-level call on the shared engine (None = the engine's own or ENSTOP_AMD_ARITHMETIC;
    "reference": the reference's bits, which presupposes the reference's entry order -- triplets that had to be re-sorted
    are summed in row-major order); the engine is handed back in its default arithmetic."""

    def __init__(self, eng, arithmetic):
        self.eng = eng
        self.arithmetic = arithmetic if arithmetic is not None else (os.environ.get("ENSTOP_AMD_ARITHMETIC") or None)

    def __enter__(self):
        self.eng.set_arithmetic(self.arithmetic)

    def __exit__(self, *exc):
        self.eng.set_arithmetic(None)


def _stage(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, device=None, arithmetic=None):
    k, m = p_w_given_z.shape
    n = p_z_given_d.shape[0]
    csr, order = _coo_to_csr(X_rows, X_cols, X_vals, n, m)
    eng = get_engine(device)
    eng.upload_csr(csr)
    eng.set_factors(p_z_given_d, p_w_given_z)
    return eng, order


@_locked
def plsa_e_step(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, p_z_given_wd,
                probability_threshold=1e-32, device=None, arithmetic=None):
    """P(z|w,d) for every stored (d, w); fills and returns `p_z_given_wd` [nnz, k]."""
    eng, order = _stage(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, device, arithmetic)
    with _KernelArithmetic(eng, arithmetic):
        P = eng.e_step(probability_threshold)
    if order is None:
        p_z_given_wd[...] = P
    else:
        p_z_given_wd[order] = P
    return p_z_given_wd


@_locked
def _m_step(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, p_z_given_wd, sample_weight,
            norm_pwz, norm_pdz, update_v, device, arithmetic=None):
    eng, order = _stage(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, device, arithmetic)
    P = np.asarray(p_z_given_wd, np.float32)
    eng.set_p(P if order is None else P[order])
    with _KernelArithmetic(eng, arithmetic):
        npwz, npdz = eng.m_step(sample_weight, update_v=update_v)
    U, V = eng.get_factors(want_v=update_v)
    p_z_given_d[...] = U
    if update_v:
        p_w_given_z[...] = V
        if norm_pwz is not None:
            norm_pwz[...] = npwz
    if norm_pdz is not None:
        norm_pdz[...] = npdz
    return p_w_given_z, p_z_given_d


def plsa_m_step(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, p_z_given_wd, norm_pwz, norm_pdz,
                device=None, arithmetic=None):
    """New P(w|z), P(z|d) from P(z|w,d); overwrites both factor arrays in place and returns them."""
    return _m_step(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, p_z_given_wd, None, norm_pwz,
                   norm_pdz, True, device, arithmetic)


def plsa_m_step_w_sample_weight(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, p_z_given_wd,
                                sample_weight, norm_pwz, norm_pdz, device=None, arithmetic=None):
    """As plsa_m_step with per-document weights entering P(w|z) only (plsa.py:293-300)."""
    return _m_step(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, p_z_given_wd, sample_weight,
                   norm_pwz, norm_pdz, True, device, arithmetic)


def plsa_refit_m_step(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, p_z_given_wd, sample_weight,
                      norm_pdz, device=None, arithmetic=None):
    """M-step for P(z|d) only, topics frozen; `sample_weight` is accepted and unused exactly like
    the reference (plsa.py:801-814)."""
    return _m_step(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, p_z_given_wd, None, None,
                   norm_pdz, False, device, arithmetic)


@_locked
def log_likelihood(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, sample_weight, device=None, arithmetic=None):
    """sum x * log(sum_z P(w|z) P(z|d)) * sample_weight[d], returned as float32 like the reference."""
    eng, _ = _stage(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, device, arithmetic)
    with _KernelArithmetic(eng, arithmetic):
        return np.float32(eng.log_likelihood(sample_weight))


# ------------------------------------------------------------------------------------------------
# initialisation (host side, as in the reference)
# ------------------------------------------------------------------------------------------------
def _nndsvd(X, k):
    """Non-negative double SVD start (Boutsidis & Gallopoulos 2008) as plsa.py:458-491 spells it out: a rank-k
    randomized SVD of X (scikit-learn's PUBLIC `randomized_svd`, drawn from NumPy's global stream exactly like the
    reference's call -- it passes no random_state), the leading triplet taken as is, every further pair (u_j, v_j)
    replaced by the heavier of its positive / negative parts scaled by sqrt(s_j * |part_u| * |part_v|).
    Vectorised over j; no private scikit-learn module."""
    from sklearn.utils.extmath import randomized_svd
    U, S, Vh = randomized_svd(X, k)
    up, un = np.maximum(U, 0.0), np.maximum(-U, 0.0)             # [n, k]
    vp, vn = np.maximum(Vh, 0.0), np.maximum(-Vh, 0.0)           # [k, m]
    up_n, un_n = np.linalg.norm(up, axis=0), np.linalg.norm(un, axis=0)
    vp_n, vn_n = np.linalg.norm(vp, axis=1), np.linalg.norm(vn, axis=1)
    pos = up_n * vp_n > un_n * vn_n                              # plsa.py:478: strictly heavier positive part
    with np.errstate(divide="ignore", invalid="ignore"):         # an all-zero part gives nan exactly like the reference
        u = np.where(pos, up / up_n, un / un_n)
        v = np.where(pos[:, None], vp / vp_n[:, None], vn / vn_n[:, None])
        lbd = np.sqrt(S * np.where(pos, up_n * vp_n, un_n * vn_n))
    W = np.ascontiguousarray(u * lbd)
    H = np.ascontiguousarray(v * lbd[:, None])
    W[:, 0] = np.sqrt(S[0]) * np.abs(U[:, 0])                    # the leading triplet is non-negative (plsa.py:463-466)
    H[0, :] = np.sqrt(S[0]) * np.abs(Vh[0, :])
    return W, H


def plsa_init(X, k, init="random", rng=np.random):
    """Initial (P(z|d), P(w|z)) as float64, rows L1-normalised (plsa.py:412-513).

    "random" draws `rng.rand(k, m)` first and `rng.rand(n, k)` second -- the order matters for
    seed-for-seed agreement with the reference (plsa.py:455-456)."""
    n, m = X.shape
    if isinstance(init, str) and init == "random":
        p_w_given_z = rng.rand(k, m)
        p_z_given_d = rng.rand(n, k)
    elif isinstance(init, str) and init == "nndsvd":
        p_z_given_d, p_w_given_z = _nndsvd(X, k)
    elif isinstance(init, str) and init == "nmf":
        from sklearn.decomposition import non_negative_factorization
        W, H, _ = non_negative_factorization(X, n_components=k, init="nndsvd", solver="cd",
                                             beta_loss=2, tol=1e-2, max_iter=100)
        p_z_given_d, p_w_given_z = np.array(W, np.float64, order="C"), np.array(H, np.float64, order="C")
    elif isinstance(init, (tuple, list)):
        p_z_given_d, p_w_given_z = init
        p_z_given_d = np.array(p_z_given_d, dtype=np.float64, order="C")
        p_w_given_z = np.array(p_w_given_z, dtype=np.float64, order="C")
        if p_z_given_d.shape != (n, k) or p_w_given_z.shape != (k, m):
            raise ValueError("init factors have shapes {} and {}, expected {} and {}".format(
                p_z_given_d.shape, p_w_given_z.shape, (n, k), (k, m)))
    else:
        raise ValueError("Unrecognized init {}".format(init))
    normalize(p_w_given_z, axis=1)
    normalize(p_z_given_d, axis=1)
    return p_z_given_d, p_w_given_z


# ------------------------------------------------------------------------------------------------
# EM drivers
# ------------------------------------------------------------------------------------------------
@_locked
def plsa_fit_inner(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, sample_weight, n_iter=100,
                   n_iter_per_test=10, tolerance=0.001, e_step_thresh=1e-32,
                   use_sample_weights=False, device=None, flags=None, arithmetic=None):
    """EM loop on COO triplets; factor arrays are updated in place and returned (plsa.py:517-640)."""
    eng, _ = _stage(X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, device)
    sw = np.asarray(sample_weight, np.float32)
    flags = (default_flags() if flags is None else flags) | arithmetic_flags(arithmetic)
    # the weighted M-step only when requested (plsa.py:606-628); the log-likelihood always sees the
    # weights (plsa.py:591, 631): non-unit weights with use_sample_weights=False reach the engine
    # with the PLSA_SW_LL_ONLY flag
    if not np.any(sw != 1.0):
        sw = None
    elif not use_sample_weights:
        flags |= PLSA_SW_LL_ONLY
    eng.fit(sw, n_iter, n_iter_per_test, tolerance, e_step_thresh, flags)
    U, V = eng.get_factors()
    p_z_given_d[...] = U
    p_w_given_z[...] = V
    return p_z_given_d, p_w_given_z


def _fit_on_engine(eng, k, sample_weight, init, n_iter, n_iter_per_test, tolerance, e_step_thresh,
                   random_state, flags=None, trace=False, shape=None, X_for_init=None):
    """plsa_fit's body for a corpus already resident on `eng` (shared with the ensemble member)."""
    n, m, _ = eng.shape
    rng = check_random_state(random_state)
    if isinstance(init, str) and init == "device_random":
        # additive, non-reference option: counter-based RNG on the GPU (host MT19937 draws cost
        # ~1 s for 1M x 64 + 64 x 100k, four times the 50-iteration fit itself)
        eng.init_factors_device(k, int(rng.randint(0, 2 ** 31 - 1)))
        sw = None
        if sample_weight is not None and np.any(np.asarray(sample_weight) != 1.0):
            sw = np.asarray(sample_weight, np.float32)
        return eng.fit(sw, n_iter, n_iter_per_test, tolerance, e_step_thresh, flags, trace=trace)

    device_init = os.environ.get("ENSTOP_AMD_HOST_INIT", "auto")   # "1": host, "0": device, auto: by size
    use_device_init = device_init == "0" or (device_init == "auto" and k * (n + m) >= 262_144)
    if isinstance(init, str) and init == "random" and isinstance(rng, np.random.RandomState) and use_device_init:
        # same draws, same float64 normalisation, same float32 casts as plsa_init + plsa.py:709-710,
        # evaluated on the device from rng's own MT19937 state (bit-identical, rng is advanced).  Long
        # streams are cut into jump-ahead pieces (csrc/mt_jump.hpp): config 3's 141 M words take 8 ms
        # against 0.45 s for the host draws + normalisation + upload; tiny problems stay on the host
        eng.init_factors_numpy_stream(k, rng)
        sw = None
        if sample_weight is not None and np.any(np.asarray(sample_weight) != 1.0):
            sw = np.asarray(sample_weight, np.float32)
        return eng.fit(sw, n_iter, n_iter_per_test, tolerance, e_step_thresh, flags, trace=trace)

    class _Shape:                      # plsa_init only needs .shape for "random" / tuple inits
        pass
    Xs = X_for_init
    if Xs is None:
        Xs = _Shape()
        Xs.shape = (n, m)
    p_z_given_d, p_w_given_z = plsa_init(Xs, k, init=init, rng=rng)
    p_z_given_d = p_z_given_d.astype(np.float32, order="C")
    p_w_given_z = p_w_given_z.astype(np.float32, order="C")
    sw = None
    if sample_weight is not None and np.any(np.asarray(sample_weight) != 1.0):   # plsa.py:712
        sw = np.asarray(sample_weight, np.float32)
    eng.set_factors(p_z_given_d, p_w_given_z)
    iters, ll = eng.fit(sw, n_iter, n_iter_per_test, tolerance, e_step_thresh, flags, trace=trace)
    return iters, ll


@_locked
def plsa_fit(X, k, sample_weight, init="random", n_iter=100, n_iter_per_test=10, tolerance=0.001,
             e_step_thresh=1e-32, random_state=None, device=None, flags=None, return_info=False, arithmetic=None):
    """Fit pLSA with k topics to the sparse doc-term matrix X; returns (P(z|d) [n,k], P(w|z) [k,m]),
    both float32 (plsa.py:643-730).  Extra keyword arguments select the device, the kernel
    schedule (fused / materialised) and the arithmetic (`arithmetic="reference"`: the reference's float32 sums,
    rounding for rounding -- engine.arithmetic_flags); positional compatibility is unchanged."""
    if not issparse(X):
        X = csr_matrix(X)
    eng = get_engine(device)
    eng.upload_csr(X)
    if arithmetic is not None:
        flags = (default_flags() if flags is None else flags) | arithmetic_flags(arithmetic)
    iters, ll = _fit_on_engine(eng, k, sample_weight, init, n_iter, n_iter_per_test, tolerance,
                               e_step_thresh, random_state, flags, trace=return_info, X_for_init=X)
    p_z_given_d, p_w_given_z = eng.get_factors()
    if return_info:
        return p_z_given_d, p_w_given_z, dict(n_iter=iters, log_likelihood_trace=ll)
    return p_z_given_d, p_w_given_z


@_locked
def plsa_refit_inner(X_rows, X_cols, X_vals, topics, p_z_given_d, sample_weight, n_iter=50,
                     n_iter_per_test=10, tolerance=0.005, e_step_thresh=1e-32, device=None, flags=None, arithmetic=None):
    """EM on P(z|d) with the topics frozen (plsa.py:820-920); returns P(z|d)."""
    eng, _ = _stage(X_rows, X_cols, X_vals, topics, p_z_given_d, device)
    sw = np.asarray(sample_weight, np.float32)
    if arithmetic is not None:
        flags = (default_flags() if flags is None else flags) | arithmetic_flags(arithmetic)
    eng.refit(None if not np.any(sw != 1.0) else sw, n_iter, n_iter_per_test, tolerance, e_step_thresh, flags)
    U, _ = eng.get_factors(want_v=False)
    p_z_given_d[...] = U
    return p_z_given_d


@_locked
def plsa_refit(X, topics, sample_weight, n_iter=50, n_iter_per_test=10, tolerance=0.005,
               e_step_thresh=1e-32, random_state=None, device=None, flags=None, return_info=False, arithmetic=None):
    """Document vectors P(z|d) for X against fixed `topics` (plsa.py:923-997)."""
    if arithmetic is not None:
        flags = (default_flags() if flags is None else flags) | arithmetic_flags(arithmetic)
    if not issparse(X):
        X = csr_matrix(X)
    topics = np.asarray(topics)
    k = topics.shape[0]
    rng = check_random_state(random_state)
    eng = get_engine(device)
    eng.upload_csr(X)
    device_init = os.environ.get("ENSTOP_AMD_HOST_INIT", "auto")
    if isinstance(rng, np.random.RandomState) and (
            device_init == "0" or (device_init == "auto" and k * X.shape[0] >= 262_144)):
        # rng.rand(n, k), float64 row normalisation, float32 cast (plsa.py:979-981) on the device
        # from rng's own stream: bit-identical, 0.22 s of host draws saved at 1 M documents x 64
        eng.init_factors_numpy_stream(k, rng, topics=topics)
    else:
        p_z_given_d = rng.rand(X.shape[0], k)                # plsa.py:979
        normalize(p_z_given_d, axis=1)
        p_z_given_d = p_z_given_d.astype(np.float32)
        eng.set_factors(p_z_given_d, topics.astype(np.float32))
    sw = None
    if sample_weight is not None and np.any(np.asarray(sample_weight) != 1.0):
        sw = np.asarray(sample_weight, np.float32)
    iters, ll = eng.refit(sw, n_iter, n_iter_per_test, tolerance, e_step_thresh, flags, trace=return_info)
    U, _ = eng.get_factors(want_v=False)
    if return_info:
        return U, dict(n_iter=iters, log_likelihood_trace=ll)
    return U


# ------------------------------------------------------------------------------------------------
# estimator
# ------------------------------------------------------------------------------------------------
class _TopicMetricsMixin:
    """coherence() / log_lift() of the fitted topics, signatures as enstop/plsa.py:1222-1285."""

    def _metric(self, topic_num, n_words, single, mean):
        if not isinstance(topic_num, int) and topic_num is not None:
            raise ValueError("Topic number must be an integer or None.")
        if topic_num is None:
            return mean(self.components_, self.training_data_, n_words)
        if 0 <= topic_num < self.n_components:
            return single(self.components_, topic_num, self.training_data_, n_words)
        raise ValueError("Topic number must be in range 0 to {}".format(self.n_components))

    def coherence(self, topic_num=None, n_words=20):
        return self._metric(topic_num, n_words, coherence, mean_coherence)

    def log_lift(self, topic_num=None, n_words=20):
        return self._metric(topic_num, n_words, log_lift, mean_log_lift)


class PLSA(_TopicMetricsMixin, BaseEstimator, TransformerMixin):
    """Probabilistic Latent Semantic Analysis with the reference estimator's constructor, methods
    and fitted attributes (enstop/plsa.py:1000-1285): `components_` = P(w|z) [k, m],
    `embedding_` = P(z|d) [n, k], `training_data_`.  Additive: `n_iter_` (EM iterations run),
    and the `device` constructor keyword at the end of the signature.

    Conscious deviations from reference defects: float input detection works on
    current NumPy (the reference's `np.float` raises); `sample_weight` is restricted to the
    non-empty rows together with the data (the reference leaves it misaligned, plsa.py:1144-1164).
    """

    def __init__(self, n_components=10, init="random", n_iter=100, n_iter_per_test=10,
                 tolerance=0.001, e_step_thresh=1e-32, transform_random_seed=42, random_state=None,
                 device=None, arithmetic=None):
        self.n_components = n_components
        self.arithmetic = arithmetic
        self.init = init
        self.n_iter = n_iter
        self.n_iter_per_test = n_iter_per_test
        self.tolerance = tolerance
        self.e_step_thresh = e_step_thresh
        self.transform_random_seed = transform_random_seed
        self.random_state = random_state
        self.device = device

    _extra_flags = 0      # subclasses: stop-test variant of the reference module they stand in for

    def _flags(self):
        # arithmetic="reference": the reference's float32 sums, rounding for rounding (engine.arithmetic_flags)
        return default_flags() | self._extra_flags | arithmetic_flags(getattr(self, "arithmetic", None))

    def _fit_factors(self, X, sample_weight):
        return plsa_fit(X, self.n_components, sample_weight, self.init, self.n_iter, self.n_iter_per_test,
                        self.tolerance, self.e_step_thresh, self.random_state, device=self.device,
                        flags=self._flags(), return_info=True)

    def fit(self, X, y=None, sample_weight=None):
        self.fit_transform(X, sample_weight=sample_weight)
        return self

    def fit_transform(self, X, y=None, sample_weight=None):
        X = check_array(X, accept_sparse="csr")
        X = standardize_input(X)
        if not issparse(X):
            X = csr_matrix(X)
        sample_weight = _check_sample_weight(sample_weight, X, dtype=np.float32)
        if np.any(X.data < 0):
            raise ValueError("PLSA is only valid for matrices with non-negative entries")
        row_sums = np.asarray(X.sum(axis=1)).ravel()
        good_rows = row_sums != 0
        all_good = bool(np.all(good_rows))
        data_for_fitting = X if all_good else X[good_rows]
        weights = sample_weight if all_good else sample_weight[good_rows]
        U, V, info = self._fit_factors(data_for_fitting, weights)
        if all_good:
            self.embedding_ = U
        else:                                    # float64 zeros, like plsa.py:1174
            self.embedding_ = np.zeros((X.shape[0], self.n_components))
            self.embedding_[good_rows] = U
        self.components_ = V
        self.training_data_ = X
        self.n_iter_ = info["n_iter"]
        return self.embedding_

    def transform(self, X, y=None):
        X = check_array(X, accept_sparse="csr")
        random_state = check_random_state(self.transform_random_seed)
        sample_weight = _check_sample_weight(None, X, dtype=np.float32)
        # the reference converts to COO here (plsa.py:1208); the engine consumes the CSR arrays in the
        # same entry order directly (and, unlike a COO -> CSR round trip, never merges duplicates)
        X = csr_matrix(X) if not issparse(X) else X.tocsr()
        # plsa.py:1210-1218: fixed n_iter=50, n_iter_per_test=5, tolerance=0.001
        return plsa_refit(X, self.components_, sample_weight, n_iter=50, n_iter_per_test=5,
                          tolerance=0.001, random_state=random_state, device=self.device,
                          flags=self._flags())


class StreamedPLSA(PLSA):
    """Drop-in name for enstop.streamed_plsa.StreamedPLSA (streamed_plsa.py:1042-1337).  The reference
    class bounds memory by materialising P(z|w,d) for `block_size` non-zeros at a time
    (streamed_plsa.py:341-375); the fused HIP schedule never materialises it at all, so `block_size`
    is accepted for signature compatibility and has no effect.  Results equal `PLSA`'s."""

    def __init__(self, n_components=10, init="random", block_size=65536, n_iter=100, n_iter_per_test=10,
                 tolerance=0.001, e_step_thresh=1e-32, transform_random_seed=42, random_state=None,
                 device=None, arithmetic=None):
        super().__init__(n_components=n_components, init=init, n_iter=n_iter,
                         n_iter_per_test=n_iter_per_test, tolerance=tolerance,
                         e_step_thresh=e_step_thresh, transform_random_seed=transform_random_seed,
                         random_state=random_state, device=device, arithmetic=arithmetic)
        self.block_size = block_size

    # streamed_plsa.py:596-597: the stop test has no `change == 0` arm
    _extra_flags = PLSA_STOP_NO_ZERO_ARM

    def _flags(self):
        from .engine import PLSA_FUSED
        return PLSA_FUSED | self._extra_flags | arithmetic_flags(getattr(self, "arithmetic", None))

    def transform(self, X, y=None, sample_weight=None):
        """streamed_plsa.py:1237: unlike PLSA.transform this one takes `sample_weight` (it only enters
        the log-likelihood of plsa_refit, whose stop test never fires: same vectors either way)."""
        X = check_array(X, accept_sparse="csr")
        sample_weight = _check_sample_weight(sample_weight, X, dtype=np.float32)
        random_state = check_random_state(self.transform_random_seed)
        X = csr_matrix(X) if not issparse(X) else X.tocsr()
        return plsa_refit(X, self.components_, sample_weight, n_iter=50, n_iter_per_test=5,
                          tolerance=0.001, random_state=random_state, device=self.device,
                          flags=self._flags())


class BlockParallelPLSA(PLSA):
    """Drop-in name for enstop.block_parallel_plsa.BlockParallelPLSA (block_parallel_plsa.py:424-538).
    The reference tiles X into n_row_blocks x n_col_blocks padded COO tiles for numba threads (and
    silently wraps tile sizes above 65 535 rows through np.uint16, block_parallel_plsa.py:359-360);
    on the GPU the work decomposition is per document row / per column item, so the two tiling
    parameters are accepted as hints and ignored.  Same EM maths as `PLSA`, with the two semantic
    differences of the reference module reproduced (golden: tests/golden/blockfit_*.npz):
    `sample_weight` is validated but never reaches the fit (block_parallel_plsa.py:499, 516-527; the
    blockwise log-likelihood has no weights, :205-248), and the stop test has no `change == 0` arm
    (:329-331)."""

    _extra_flags = PLSA_STOP_NO_ZERO_ARM

    def _fit_factors(self, X, sample_weight):
        return super()._fit_factors(X, np.ones(X.shape[0], np.float32))

    def __init__(self, n_components=10, init="random", n_row_blocks=8, n_col_blocks=8, n_iter=100,
                 n_iter_per_test=10, tolerance=0.001, e_step_thresh=1e-32, transform_random_seed=42,
                 random_state=None, device=None, arithmetic=None):
        super().__init__(n_components=n_components, init=init, n_iter=n_iter,
                         n_iter_per_test=n_iter_per_test, tolerance=tolerance,
                         e_step_thresh=e_step_thresh, transform_random_seed=transform_random_seed,
                         random_state=random_state, device=device, arithmetic=arithmetic)
        self.n_row_blocks = n_row_blocks
        self.n_col_blocks = n_col_blocks


class GPUPLSA(BlockParallelPLSA):
    """Drop-in name for enstop.cuda_plsa.GPUPLSA (cuda_plsa.py:356-470), the reference's own accelerator
    estimator (numba-CUDA kernels over n_row_blocks x n_col_blocks tiles).  Nothing of that file is
    ported: this IS the MI355X engine under the same constructor, so code written against `GPUPLSA`
    keeps running; the tiling parameters are accepted and ignored."""


class DistributedPLSA(BlockParallelPLSA):
    """Drop-in name for enstop.distributed_plsa.DistributedPLSA (distributed_plsa.py:374-460).  The
    reference takes a dask array and sums per-tile partial factors with a dask graph
    (distributed_plsa.py:99-131).  Here the distribution unit is the GPU: with a communicator of more than
    one rank in force (`enstop_amd.distributed.init()`: RCCL through the C ABI, one process per GPU; or one the
    caller installed with `comm.install`), every rank passes the same X and the documents are sharded over the
    ranks -- one all-reduce of the P(w|z) accumulator per EM iteration (`sharded_plsa_fit`, DESIGN.md section
    6) -- and every rank receives the full result; in a single process it is `BlockParallelPLSA`.  Like the
    reference's loop (distributed_plsa.py:277-278, 450) it never sees sample weights and its stop test has
    no `change == 0` arm.  Dask arrays are materialised with `.compute()` first."""

    def fit_transform(self, X, y=None, sample_weight=None):
        if hasattr(X, "compute") and not issparse(X):
            X = X.compute()
        return super().fit_transform(X, y, sample_weight=sample_weight)

    def _fit_factors(self, X, sample_weight):
        from . import distributed
        _, world = distributed.rank_world()
        if world <= 1:
            return super()._fit_factors(X, sample_weight)
        from .sharded import sharded_plsa_fit
        return sharded_plsa_fit(X, self.n_components, None, self.init, self.n_iter,
                                self.n_iter_per_test, self.tolerance, self.e_step_thresh, self.random_state,
                                device=self.device, return_info=True, zero_arm=False)
