#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the produced
``tests/golden/*.npz`` files are committed and are the only thing that travels.
Nothing from the reference's source text is stored -- the fixtures hold inputs
and the outputs the reference computed for them.

How the reference is executed: numba is not importable in
this image, so ``numba`` is replaced by a no-op shim (``njit`` returns the
function unchanged, ``prange = range``) and the reference's own statements run
as plain Python/NumPy.  Under NumPy 2 (NEP 50) ``0.0 + np.float32`` stays
float32, so the accumulators are fp32 exactly like numba's typed locals and the
execution order is strictly sequential.  ``enstop/__init__.py`` is bypassed with
a synthetic package object because it imports dask/hdbscan/umap; for
``enstop_.py`` those three imports are satisfied with empty placeholder modules.
The topic-combination goldens (``gen_combine``) pin the reference's OWN
statements around the third-party calls: the all-pairs KL matrix, the mutual
reachability matrix handed to ``mst_linkage_core`` (captured at the call), and
the cluster representatives computed from labels / membership strengths that
the placeholder clusterers return as *given inputs*.  What hdbscan / umap
themselves would compute stays unpinned.

Usage:  python tests/golden/make_golden.py        (writes next to this file)
"""
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# --------------------------------------------------------------------------- shim
def install_shims():
    nb = types.ModuleType("numba")

    def _jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    class _Sink:
        def __getattr__(self, n):
            return self

        def __call__(self, *a, **k):
            return self

        def __getitem__(self, i):
            return self

    nb.njit = _jit
    nb.jit = _jit
    nb.prange = range
    nb.types = _Sink()
    nb.float32 = _Sink()
    cuda = types.ModuleType("numba.cuda")
    cuda.is_available = lambda: False
    nb.cuda = cuda
    sys.modules["numba"] = nb
    sys.modules["numba.cuda"] = cuda

    pkg = types.ModuleType("enstop")
    pkg.__path__ = [os.path.join(REF, "enstop")]
    sys.modules["enstop"] = pkg
    np.float = float  # enstop/utils.py:277 uses the removed alias

    # placeholders so `import enstop.enstop_` succeeds; never called below
    for name in ("dask", "hdbscan", "hdbscan._hdbscan_linkage", "hdbscan.hdbscan_",
                 "umap", "umap.distances"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["hdbscan._hdbscan_linkage"].mst_linkage_core = None
    sys.modules["hdbscan._hdbscan_linkage"].label = None
    sys.modules["hdbscan.hdbscan_"]._tree_to_labels = None
    sys.modules["umap.distances"].hellinger = None


install_shims()
import enstop.utils  # noqa: E402
import enstop.plsa as ref  # noqa: E402
import enstop.enstop_ as ref_ens  # noqa: E402


# --------------------------------------------------------------------------- data
def make_counts(n, m, density, seed, max_count=6, empty_rows=()):
    rs = np.random.RandomState(seed)
    X = sp.random(n, m, density=density, random_state=rs, format="csr", dtype=np.float64)
    X.data = np.ceil(X.data * max_count)
    X = X.tolil()
    for r in range(n):                      # no accidental empty rows / keep it canonical
        if X[r].nnz == 0 and r not in empty_rows:
            X[r, rs.randint(m)] = 1.0
    for r in empty_rows:
        X[r] = 0
    X = X.tocsr()
    X.eliminate_zeros()
    X.sort_indices()
    return X


def random_factors(n, m, k, seed):
    rs = np.random.RandomState(seed)
    V = rs.rand(k, m)
    U = rs.rand(n, k)
    V /= V.sum(axis=1, keepdims=True)
    U /= U.sum(axis=1, keepdims=True)
    return U.astype(np.float32), V.astype(np.float32)


class Recorder:
    """Wraps module-level functions of the reference to record the LL trace and
    the number of E-steps (= EM iterations actually executed)."""

    def __init__(self):
        self.ll = []
        self.n_e = 0
        self._ll = ref.log_likelihood
        self._e = ref.plsa_e_step

    def __enter__(self):
        def ll(*a):
            v = self._ll(*a)
            self.ll.append(np.float32(v))
            return v

        def e(*a):
            self.n_e += 1
            return self._e(*a)

        ref.log_likelihood = ll
        ref.plsa_e_step = e
        return self

    def __exit__(self, *exc):
        ref.log_likelihood = self._ll
        ref.plsa_e_step = self._e


def csr_parts(X):
    X = X.tocsr()
    return dict(indptr=X.indptr.astype(np.int32), indices=X.indices.astype(np.int32),
                data=X.data.astype(np.float32), shape=np.array(X.shape, dtype=np.int64))


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------- kernel-level goldens
def gen_kernels(name, n, m, k, density, seed, thresh, zero_doc=None):
    X = make_counts(n, m, density, seed)
    A = X.tocoo().astype(np.float32)
    U, V = random_factors(n, m, k, seed + 1)
    if zero_doc is not None:
        U[zero_doc] = 0.0                    # norm == 0 branch of the E-step (plsa.py:104)
    rs = np.random.RandomState(seed + 2)
    sw = (0.25 + 1.5 * rs.rand(n)).astype(np.float32)
    ones = np.ones(n, dtype=np.float32)

    P = np.full((A.nnz, k), -7.0, dtype=np.float32)     # poison: every cell must be written
    ref.plsa_e_step(A.row, A.col, A.data, V, U, P, np.float32(thresh))

    Vm, Um = V.copy(), U.copy()
    npwz = np.zeros(k, np.float32); npdz = np.zeros(n, np.float32)
    ref.plsa_m_step(A.row, A.col, A.data, Vm, Um, P, npwz, npdz)

    Vw, Uw = V.copy(), U.copy()
    npwz_w = np.zeros(k, np.float32); npdz_w = np.zeros(n, np.float32)
    ref.plsa_m_step_w_sample_weight(A.row, A.col, A.data, Vw, Uw, P, sw, npwz_w, npdz_w)

    Ur = U.copy()
    npdz_r = np.zeros(n, np.float32)
    ref.plsa_refit_m_step(A.row, A.col, A.data, V, Ur, P, ones, npdz_r)

    with np.errstate(divide="ignore"):
        ll1 = ref.log_likelihood(A.row, A.col, A.data, V, U, ones)
        llw = ref.log_likelihood(A.row, A.col, A.data, V, U, sw)
        ll_after = ref.log_likelihood(A.row, A.col, A.data, Vm, Um, ones)

    save(name, **csr_parts(X), k=np.int64(k), thresh=np.float32(thresh), U=U, V=V, sw=sw,
         P=P, V_m=Vm, U_m=Um, norm_pwz=npwz, norm_pdz=npdz,
         V_mw=Vw, U_mw=Uw, norm_pwz_w=npwz_w, norm_pdz_w=npdz_w,
         U_refit=Ur, norm_pdz_refit=npdz_r,
         ll_ones=np.float32(ll1), ll_sw=np.float32(llw), ll_after_m=np.float32(ll_after))


# --------------------------------------------------------------------------- driver-level goldens
def gen_fit(name, n, m, k, density, seed, n_iter, n_iter_per_test, tol, thresh=1e-32,
            weighted=False, tuple_init=False, fit_seed=7):
    X = make_counts(n, m, density, seed)
    rs = np.random.RandomState(seed + 3)
    sw = (0.5 + rs.rand(n)).astype(np.float32) if weighted else np.ones(n, np.float32)
    if tuple_init:
        Ui, Vi = random_factors(n, m, k, seed + 4)
        Ui = Ui.astype(np.float64) * 3.0        # un-normalised on purpose: plsa_init normalises
        Vi = Vi.astype(np.float64) * 0.5
        init = (Ui.copy(), Vi.copy())
    else:
        init = "random"
    # the initial factors exactly as plsa_fit derives them (plsa.py:707-710)
    if tuple_init:
        U0, V0 = ref.plsa_init(X, k, init=(Ui.copy(), Vi.copy()))
    else:
        U0, V0 = ref.plsa_init(X, k, init="random", rng=np.random.RandomState(fit_seed))
    U0 = U0.astype(np.float32, order="C"); V0 = V0.astype(np.float32, order="C")
    with Recorder() as rec, np.errstate(divide="ignore"):
        U, V = ref.plsa_fit(X, k, sw, init=init, n_iter=n_iter, n_iter_per_test=n_iter_per_test,
                            tolerance=tol, e_step_thresh=thresh, random_state=fit_seed)
    extra = {}
    if tuple_init:
        extra = dict(U_init=Ui, V_init=Vi)
    save(name, **csr_parts(X), k=np.int64(k), sw=sw, n_iter=np.int64(n_iter),
         n_iter_per_test=np.int64(n_iter_per_test), tol=np.float64(tol), thresh=np.float64(thresh),
         fit_seed=np.int64(fit_seed), U0=U0, V0=V0, U=U, V=V,
         ll_trace=np.array(rec.ll, np.float32), iters=np.int64(rec.n_e), **extra)
    print("   %s: iters=%d ll[0]=%.6g ll[-1]=%.6g" % (name, rec.n_e, rec.ll[0], rec.ll[-1]))


def gen_fit_big(name, n=5000, m=6000, k=4, density=0.05, seed=1, n_iter=2, fit_seed=7):
    """A plsa_fit large enough for the reference's float32 `norm_pwz[z] += s` (plsa.py:193: ONE running
    sum over all non-zeros per topic) to be visibly inexact: 1.5 M non-zeros, k = 4, two iterations --
    the strict and the all-float64 builds of oracle/plsa_oracle.c differ by 3e-4 of the largest P(w|z)
    entry here.  ~2 minutes of pure-Python reference time; CSR stored as int32 / uint8 to keep the file small."""
    rs = np.random.RandomState(seed)
    X = sp.random(n, m, density=density, random_state=rs, format="csr", dtype=np.float64)
    X.data = np.ceil(X.data * 6)
    X.sort_indices()
    assert (np.diff(X.indptr) > 0).all()
    sw = np.ones(n, np.float32)
    U0, V0 = ref.plsa_init(X, k, init="random", rng=np.random.RandomState(fit_seed))
    U0 = U0.astype(np.float32, order="C"); V0 = V0.astype(np.float32, order="C")
    with Recorder() as rec, np.errstate(divide="ignore"):
        U, V = ref.plsa_fit(X, k, sw, init="random", n_iter=n_iter, n_iter_per_test=10,
                            tolerance=0.0, e_step_thresh=1e-32, random_state=fit_seed)
    save(name, indptr=X.indptr.astype(np.int32), indices=X.indices.astype(np.int32),
         data_u8=X.data.astype(np.uint8), shape=np.array(X.shape, dtype=np.int64), k=np.int64(k), sw=sw,
         n_iter=np.int64(n_iter), n_iter_per_test=np.int64(10), tol=np.float64(0.0), thresh=np.float64(1e-32),
         fit_seed=np.int64(fit_seed), U0=U0, V0=V0, U=U, V=V,
         ll_trace=np.array(rec.ll, np.float32), iters=np.int64(rec.n_e))
    print("   %s: nnz=%d iters=%d ll=%s" % (name, X.nnz, rec.n_e, rec.ll))


def gen_refit(name, n, m, k, density, seed, n_iter, n_iter_per_test, tol, weighted=False):
    X = make_counts(n, m, density, seed)
    _, topics = random_factors(n, m, k, seed + 5)
    rs = np.random.RandomState(seed + 6)
    sw = (0.5 + rs.rand(n)).astype(np.float32) if weighted else np.ones(n, np.float32)
    with Recorder() as rec, np.errstate(divide="ignore"):
        U = ref.plsa_refit(X, topics, sw, n_iter=n_iter, n_iter_per_test=n_iter_per_test,
                           tolerance=tol, random_state=np.random.RandomState(42))
    save(name, **csr_parts(X), k=np.int64(k), topics=topics, sw=sw, n_iter=np.int64(n_iter),
         n_iter_per_test=np.int64(n_iter_per_test), tol=np.float64(tol), U=U,
         ll_trace=np.array(rec.ll, np.float32), iters=np.int64(rec.n_e))
    print("   %s: iters=%d" % (name, rec.n_e))


def gen_estimator(name, dtype, empty_rows, seed):
    n, m, k = 48, 64, 5
    X = make_counts(n, m, 0.15, seed, empty_rows=empty_rows)
    if dtype == "float":
        Xin = X.astype(np.float64)
        Xin.data = Xin.data * 0.37          # float input -> L1 row-normalised (utils.py:276-280)
    else:
        Xin = X.astype(np.int64)
    model = ref.PLSA(n_components=k, n_iter=30, n_iter_per_test=10, tolerance=0.0, random_state=11)
    with np.errstate(divide="ignore"):
        emb = model.fit_transform(Xin)
        Xt = make_counts(20, m, 0.2, seed + 9)
        tr = model.transform(Xt.astype(np.int64))
    parts = csr_parts(Xin)
    save(name, indptr=parts["indptr"], indices=parts["indices"],
         data=Xin.data.copy(), shape=parts["shape"], k=np.int64(k),
         embedding=np.asarray(emb), embedding_dtype=np.array(str(np.asarray(emb).dtype)),
         components=model.components_,
         t_indptr=Xt.indptr.astype(np.int32), t_indices=Xt.indices.astype(np.int32),
         t_data=Xt.data.astype(np.int64), t_shape=np.array(Xt.shape, np.int64), transformed=tr)


def gen_member(name, seed):
    """plsa_topics (enstop_.py:56-115): bootstrap + fit, and the serial ensemble stack."""
    n, m, k = 70, 90, 6
    X = make_counts(n, m, 0.12, seed)
    kw = dict(n_iter=20, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-16)
    with np.errstate(divide="ignore"):
        # (a) RandomState instance shared by bootstrap and init (stream continues)
        rs = np.random.RandomState(5)
        Va = ref_ens.plsa_topics(X, k, random_state=rs, **kw)
        idx_a = np.random.RandomState(5).randint(0, n, size=n)
        # (b) int seed: bootstrap and init both re-seed (enstop_.py:86 + plsa.py:707)
        Vb = ref_ens.plsa_topics(X, k, random_state=9, **kw)
        idx_b = np.random.RandomState(9).randint(0, n, size=n)
        # (c) bootstrap disabled
        Vc = ref_ens.plsa_topics(X, k, random_state=9, bootstrap=False, **kw)
        # (d) serial ensemble with one shared stream -> vstack (enstop_.py:220-231)
        rs = np.random.RandomState(21)
        Vd = ref_ens.ensemble_of_topics(X, k, model="plsa", n_jobs=1, n_runs=3,
                                        parallelism="none", random_state=rs, **kw)
    save(name, **csr_parts(X), k=np.int64(k), n_iter=np.int64(20), thresh=np.float64(1e-16),
         V_rs5=Va, idx_rs5=idx_a, V_int9=Vb, idx_int9=idx_b, V_nobootstrap=Vc, V_stack_rs21=Vd)


def gen_metrics(name, seed):
    """coherence / log_lift of enstop/utils.py on random topics + a small corpus (host-side metrics)."""
    import enstop.utils as ru
    n, m, k = 80, 60, 5
    X = make_counts(n, m, 0.15, seed)
    rs = np.random.RandomState(seed + 1)
    topics = rs.dirichlet(np.full(m, 0.3), size=k).astype(np.float32)
    coh = np.array([ru.coherence(topics, z, X, n_words=10) for z in range(k)])
    lift = np.array([ru.log_lift(topics, z, X, n_words=10) for z in range(k)])
    lift_all = np.array([ru.log_lift(topics, z, X) for z in range(k)])
    save(name, **csr_parts(X), topics=topics, coherence=coh, mean_coherence=np.float64(ru.mean_coherence(topics, X, n_words=10)),
         log_lift=lift, log_lift_allwords=lift_all, mean_log_lift=np.float64(ru.mean_log_lift(topics, X, n_words=10)))


def gen_fit_inner_ll_only(name, seed):
    """plsa_fit_inner called directly with non-unit weights and use_sample_weights=False (plsa.py:591,
    606-628, 631): unweighted M-step, weighted log-likelihood / stop test."""
    n, m, k = 50, 60, 5
    X = make_counts(n, m, 0.18, seed)
    A = X.tocoo().astype(np.float32)
    U0, V0 = random_factors(n, m, k, seed + 1)
    sw = (0.25 + 2.0 * np.random.RandomState(seed + 2).rand(n)).astype(np.float32)
    U, V = U0.copy(), V0.copy()
    with Recorder() as rec, np.errstate(divide="ignore"):
        ref.plsa_fit_inner(A.row, A.col, A.data, V, U, sw, n_iter=40, n_iter_per_test=5, tolerance=2e-3,
                           e_step_thresh=np.float32(1e-32), use_sample_weights=False)
    save(name, **csr_parts(X), k=np.int64(k), sw=sw, U0=U0, V0=V0, U=U, V=V, n_iter=np.int64(40),
         n_iter_per_test=np.int64(5), tol=np.float64(2e-3), ll_trace=np.array(rec.ll, np.float32),
         iters=np.int64(rec.n_e))
    print("   %s: iters=%d" % (name, rec.n_e))


def gen_blockfit(name, n, m, k, density, seed, n_iter, n_iter_per_test, tol, blocks=(3, 2), fit_seed=7):
    """enstop/block_parallel_plsa.py plsa_fit (:339-421): tiled EM, no sample weights, stop test
    without the `change == 0` arm (:329-331).  Iterations counted at plsa_em_step_by_blocks."""
    import enstop.block_parallel_plsa as bp
    X = make_counts(n, m, density, seed)
    count = {"em": 0, "ll": []}
    em, ll = bp.plsa_em_step_by_blocks, bp.log_likelihood_by_blocks

    def em_w(*a):
        count["em"] += 1
        return em(*a)

    def ll_w(*a):
        v = ll(*a)
        count["ll"].append(np.float32(v))
        return v
    bp.plsa_em_step_by_blocks, bp.log_likelihood_by_blocks = em_w, ll_w
    try:
        with np.errstate(divide="ignore", invalid="ignore"):
            U, V = bp.plsa_fit(X, k, n_row_blocks=blocks[0], n_col_blocks=blocks[1], n_iter=n_iter,
                               n_iter_per_test=n_iter_per_test, tolerance=tol, random_state=fit_seed)
    finally:
        bp.plsa_em_step_by_blocks, bp.log_likelihood_by_blocks = em, ll
    save(name, **csr_parts(X), k=np.int64(k), n_iter=np.int64(n_iter), n_iter_per_test=np.int64(n_iter_per_test),
         tol=np.float64(tol), fit_seed=np.int64(fit_seed), n_row_blocks=np.int64(blocks[0]),
         n_col_blocks=np.int64(blocks[1]), U=np.asarray(U, np.float32), V=np.asarray(V, np.float32),
         ll_trace=np.array(count["ll"], np.float32), iters=np.int64(count["em"]))
    print("   %s: iters=%d" % (name, count["em"]))


def gen_fit_cfg1(name="fit_cfg1_shape", corpus="gpurun_out/r04/cfg1_corpus.npz", k=20, n_iter=2, fit_seed=7,
                 n_cols=4000):
    """The reference itself at BASELINE config 1's exact shape: the 18 846 x 173 762 / 2.95 M-nnz corpus of
    `plsa_generate_synthetic(seed=0)` (downloaded once from the GPU box with tools/dump_synthetic_corpus.py:
    the generator runs in HBM), enstop/plsa.py plsa_fit, k = 20, two iterations, tolerance 0.  ~7 minutes
    of pure-Python reference time.  Stored: the corpus (so that CPU-only tests can run the oracle on it and
    the GPU test can check the generator still produces it), P(z|d) in full, the log-likelihood trace, the
    row sums of P(w|z) and `n_cols` of its columns (the 1000 most frequent words + a seeded random sample)."""
    root = os.path.dirname(os.path.dirname(HERE))
    c = np.load(os.path.join(root, corpus))
    n, m = (int(v) for v in c["shape"])
    X = sp.csr_matrix((c["data_u16"].astype(np.float64), c["indices"], c["indptr"]), shape=(n, m))
    assert X.has_sorted_indices and (np.diff(X.indptr) > 0).all()
    sw = np.ones(n, np.float32)
    import time
    t0 = time.time()
    with Recorder() as rec, np.errstate(divide="ignore"):
        U, V = ref.plsa_fit(X, k, sw, init="random", n_iter=n_iter, n_iter_per_test=10, tolerance=0.0,
                            e_step_thresh=1e-32, random_state=fit_seed)
    print("   reference run: %.0f s" % (time.time() - t0))
    freq = np.asarray((X > 0).sum(axis=0)).ravel()
    head = np.argsort(-freq, kind="stable")[:1000]
    rest = np.setdiff1d(np.arange(m), head)
    cols = np.sort(np.concatenate([head, np.random.RandomState(1).choice(rest, n_cols - 1000, replace=False)]))
    # int32 column deltas inside a row compress far better than the raw sorted ids
    idx = c["indices"].astype(np.int64)
    delta = np.diff(idx, prepend=0)
    delta[c["indptr"][:-1]] = idx[c["indptr"][:-1]]
    save(name, indptr=c["indptr"].astype(np.int32), indices_rowdelta=delta.astype(np.int32),
         data_u8=c["data_u16"].astype(np.uint8), shape=c["shape"], corpus_sha256=c["sha256"],
         corpus_seed=c["seed"], k=np.int64(k), n_iter=np.int64(n_iter), n_iter_per_test=np.int64(10),
         tol=np.float64(0.0), thresh=np.float64(1e-32), fit_seed=np.int64(fit_seed), U=U,
         V_cols=cols.astype(np.int32), V_sample=np.ascontiguousarray(V[:, cols]),
         V_rowsum64=V.astype(np.float64).sum(axis=1), V_max=V.max(axis=1),
         V_abs_checksum64=np.float64(np.abs(V.astype(np.float64)).sum()),
         ll_trace=np.array(rec.ll, np.float32), iters=np.int64(rec.n_e))
    np.save(os.path.join(root, "gpurun_out", "r04", "cfg1_reference_V_full.npy"), V)   # scratch, not committed
    print("   %s: nnz=%d iters=%d ll=%s" % (name, X.nnz, rec.n_e, rec.ll))


class StreamRecorder:
    """Counts EM steps / records the log-likelihood trace of enstop/streamed_plsa.py's loops (the module
    binds `log_likelihood` by name at import, so the wrappers are installed on the module itself)."""

    def __init__(self, sp_mod):
        self.m = sp_mod
        self.ll, self.n_em = [], 0
        self.saved = {}

    def __enter__(self):
        m = self.m

        def count(f):
            def g(*a, **k):
                self.n_em += 1
                return f(*a, **k)
            return g

        def ll_w(*a):
            v = self.saved["log_likelihood"](*a)
            self.ll.append(np.float32(v))
            return v
        for name in ("plsa_em_step", "plsa_em_step_w_sample_weights", "plsa_refit_em_step"):
            self.saved[name] = getattr(m, name)
            setattr(m, name, count(self.saved[name]))
        self.saved["log_likelihood"] = m.log_likelihood
        m.log_likelihood = ll_w
        return self

    def __exit__(self, *exc):
        for name, f in self.saved.items():
            setattr(self.m, name, f)


def gen_streamfit(name, n, m, k, density, seed, n_iter, n_iter_per_test, tol, block_size, thresh=1e-32,
                  weighted=False, fit_seed=7):
    """enstop/streamed_plsa.py plsa_fit (:606-699) -> plsa_fit_inner_blockwise (:469-603): E-step and partial
    M-step over blocks of `block_size` non-zeros (:349-375), sample weights as plsa.py (:304-320), stop test
    WITHOUT the `change == 0` arm (:596-597).  Iterations counted at plsa_em_step*."""
    import enstop.streamed_plsa as st
    X = make_counts(n, m, density, seed)
    rs = np.random.RandomState(seed + 3)
    sw = (0.5 + rs.rand(n)).astype(np.float32) if weighted else np.ones(n, np.float32)
    with StreamRecorder(st) as rec, np.errstate(divide="ignore", invalid="ignore"):
        U, V = st.plsa_fit(X, k, sw, init="random", block_size=block_size, n_iter=n_iter,
                           n_iter_per_test=n_iter_per_test, tolerance=tol, e_step_thresh=thresh,
                           random_state=fit_seed)
    # the same call through enstop/plsa.py: recorded so that the tests can state where the two modules
    # agree bit for bit (same summation order) and where they part (the `change == 0` arm)
    with Recorder() as rec2, np.errstate(divide="ignore", invalid="ignore"):
        U2, V2 = ref.plsa_fit(X, k, sw, init="random", n_iter=n_iter, n_iter_per_test=n_iter_per_test,
                              tolerance=tol, e_step_thresh=thresh, random_state=fit_seed)
    save(name, **csr_parts(X), k=np.int64(k), sw=sw, n_iter=np.int64(n_iter),
         n_iter_per_test=np.int64(n_iter_per_test), tol=np.float64(tol), thresh=np.float64(thresh),
         block_size=np.int64(block_size), fit_seed=np.int64(fit_seed), U=np.asarray(U, np.float32),
         V=np.asarray(V, np.float32), ll_trace=np.array(rec.ll, np.float32), iters=np.int64(rec.n_em),
         plsa_py_iters=np.int64(rec2.n_e), plsa_py_same_factors=np.bool_(
             np.array_equal(U, U2) and np.array_equal(V, V2)))
    print("   %s: nnz=%d blocks=%d iters=%d (plsa.py: %d, same factors: %s)" % (
        name, X.nnz, X.nnz // block_size + 1, rec.n_em, rec2.n_e,
        np.array_equal(U, U2) and np.array_equal(V, V2)))


def gen_streamrefit(name, n, m, k, density, seed, n_iter, n_iter_per_test, tol, block_size, thresh=1e-32,
                    weighted=False):
    """enstop/streamed_plsa.py plsa_refit (:959-1039) -> plsa_refit_inner_blockwise (:851-956).  Quirks the
    fixture pins: the loop never stops early (`if current_log_likelihood > 0`, :949) and `e_step_thresh` is
    NOT handed to plsa_refit_em_step (:932-943), which runs with its default 1e-32 whatever the caller
    passed -- `U_default_thresh` is the same call with e_step_thresh=1e-32."""
    import enstop.streamed_plsa as st
    X = make_counts(n, m, density, seed)
    _, topics = random_factors(n, m, k, seed + 5)
    rs = np.random.RandomState(seed + 6)
    sw = (0.5 + rs.rand(n)).astype(np.float32) if weighted else np.ones(n, np.float32)
    with StreamRecorder(st) as rec, np.errstate(divide="ignore", invalid="ignore"):
        U = st.plsa_refit(X, topics, sw, block_size=block_size, n_iter=n_iter, n_iter_per_test=n_iter_per_test,
                          tolerance=tol, e_step_thresh=thresh, random_state=np.random.RandomState(42))
    with np.errstate(divide="ignore", invalid="ignore"):
        Ud = st.plsa_refit(X, topics, sw, block_size=block_size, n_iter=n_iter, n_iter_per_test=n_iter_per_test,
                           tolerance=tol, e_step_thresh=1e-32, random_state=np.random.RandomState(42))
        # enstop/plsa.py's refit on the same call (it DOES honour e_step_thresh, plsa.py:893)
        Up = ref.plsa_refit(X, topics, sw, n_iter=n_iter, n_iter_per_test=n_iter_per_test, tolerance=tol,
                            e_step_thresh=thresh, random_state=np.random.RandomState(42))
    save(name, **csr_parts(X), k=np.int64(k), topics=topics, sw=sw, n_iter=np.int64(n_iter),
         n_iter_per_test=np.int64(n_iter_per_test), tol=np.float64(tol), thresh=np.float64(thresh),
         block_size=np.int64(block_size), U=np.asarray(U, np.float32), U_default_thresh=np.asarray(Ud, np.float32),
         U_plsa_py=np.asarray(Up, np.float32), ll_trace=np.array(rec.ll, np.float32), iters=np.int64(rec.n_em))
    print("   %s: iters=%d thresh ignored: %s, equals plsa.py: %s" % (
        name, rec.n_em, np.array_equal(U, Ud), np.array_equal(U, Up)))


def gen_streamestimator(name, seed, empty_rows=(0, 17, 47)):
    """StreamedPLSA.fit_transform / transform (streamed_plsa.py:1167-1268): int input with empty rows
    (float64 `embedding_`, :1216), weighted fit, transform with and without sample_weight (:1237)."""
    import enstop.streamed_plsa as st
    n, m, k = 48, 64, 5
    X = make_counts(n, m, 0.15, seed, empty_rows=empty_rows).astype(np.int64)
    model = st.StreamedPLSA(n_components=k, block_size=100, n_iter=30, n_iter_per_test=10, tolerance=0.0,
                            random_state=11)
    Xt = make_counts(20, m, 0.2, seed + 9).astype(np.int64)
    swt = np.linspace(0.5, 2.0, Xt.shape[0])
    with np.errstate(divide="ignore", invalid="ignore"):
        emb = model.fit_transform(X)
        tr = model.transform(Xt)
        tr_w = model.transform(Xt, sample_weight=swt)
    parts = csr_parts(X)
    save(name, indptr=parts["indptr"], indices=parts["indices"], data=X.data.copy(), shape=parts["shape"],
         k=np.int64(k), block_size=np.int64(100), embedding=np.asarray(emb),
         embedding_dtype=np.array(str(np.asarray(emb).dtype)), components=model.components_,
         t_indptr=Xt.indptr.astype(np.int32), t_indices=Xt.indices.astype(np.int32),
         t_data=Xt.data.astype(np.int64), t_shape=np.array(Xt.shape, np.int64), t_sw=swt,
         transformed=tr, transformed_weighted=tr_w)


def gen_nndsvd(name, seed):
    """plsa_init(init="nndsvd") (plsa.py:458-491, 510-511).  The reference calls randomized_svd(X, k) without a
    random_state: NumPy's global stream is seeded here (and in the test) so that both sides draw the same SVD."""
    n, m, k = 60, 80, 6
    X = make_counts(n, m, 0.15, seed)
    np.random.seed(4242)
    U, V = ref.plsa_init(X, k, init="nndsvd")
    save(name, **csr_parts(X), k=np.int64(k), numpy_seed=np.int64(4242), U=U, V=V)


def gen_combine(name, seed, t=24, m=150, min_samples=3):
    """Topic combination, enstop/enstop_.py:234-253 (KL), :283-296 (mutual reachability), :299-308 /
    :340-345 / :385-393 (cluster representatives).  Third-party calls are placeholders that record their
    argument and return given labels; every stored output is computed by the reference's statements."""
    rs = np.random.RandomState(seed)
    centres = rs.dirichlet(np.full(m, 0.08), size=5)
    topics = np.vstack([rs.dirichlet(centres[i % 5] * 60 + 0.02) for i in range(t)]).astype(np.float32)
    topics[topics < 1e-6] = 0.0                          # exact zeros: the `> 0` guards of kl_divergence
    topics[3, :] = 0.0; topics[3, :7] = 1.0 / 7.0        # a sparse row
    topics /= topics.sum(axis=1, keepdims=True)
    topics = topics.astype(np.float32)
    labels = (np.arange(t) % 5).astype(np.int64)
    labels[[1, 7]] = -1                                  # noise points
    probs = rs.rand(t)
    probs[labels == 4] = 0.0                             # a cluster whose weights are all zero
    probs[np.argmax(labels == 4)] = 0.5

    with np.errstate(divide="ignore", invalid="ignore"):
        kl32 = ref_ens.all_pairs_kl_divergence(topics)
        kl64 = ref_ens.all_pairs_kl_divergence(topics.astype(np.float64))
    captured = {}

    def fake_mst(mr):
        captured["mr"] = np.array(mr, copy=True)
        return np.zeros((mr.shape[0] - 1, 3))
    saved = (ref_ens.mst_linkage_core, ref_ens.label, ref_ens._tree_to_labels)
    ref_ens.mst_linkage_core = fake_mst
    ref_ens.label = lambda mst: mst
    ref_ens._tree_to_labels = lambda X, tree, **kw: (labels, None, None, None, None)
    try:
        with np.errstate(divide="ignore", invalid="ignore"):
            rep_kl = ref_ens.generate_combined_topics_kl(topics, min_samples=min_samples, min_cluster_size=3)
    finally:
        ref_ens.mst_linkage_core, ref_ens.label, ref_ens._tree_to_labels = saved

    class FakeHDBSCAN:
        def __init__(self, **kw):
            self.labels_, self.probabilities_ = labels, probs

        def fit_predict(self, D):
            captured["D_hell_shape"] = D.shape
            return labels

        def fit(self, E):
            return self

    class FakeUMAP:
        def __init__(self, **kw):
            pass

        def fit_transform(self, X):
            return np.zeros((X.shape[0], 2))
    ref_ens.hdbscan.HDBSCAN = FakeHDBSCAN
    ref_ens.umap.UMAP = FakeUMAP
    # umap.distances.hellinger is absent: the all-pairs Hellinger MATRIX is not pinned here, only what
    # the reference does with the labels afterwards
    ref_ens.hellinger = lambda a, b: 0.0
    rep_hell = ref_ens.generate_combined_topics_hellinger(topics, min_samples=min_samples, min_cluster_size=3)
    rep_umap = ref_ens.generate_combined_topics_hellinger_umap(topics, min_samples=min_samples, min_cluster_size=3)
    save(name, topics=topics, min_samples=np.int64(min_samples), kl_f32_input=np.asarray(kl32, np.float64),
         kl_f64_input=np.asarray(kl64, np.float64), mutual_reachability=captured["mr"],
         labels=labels, probabilities=probs, rep_kl=rep_kl, rep_hellinger=rep_hell, rep_umap=rep_umap)


def gen_stream_all():
    # block sizes far below nnz: several blocks per EM step, incl. nnz an exact multiple of the block
    gen_streamfit("streamfit_k6", n=70, m=90, k=6, density=0.12, seed=700, n_iter=25, n_iter_per_test=10, tol=0.0,
                  block_size=128)
    gen_streamfit("streamfit_k4_weighted", n=50, m=60, k=4, density=0.2, seed=220, n_iter=21, n_iter_per_test=5,
                  tol=0.0, block_size=100, weighted=True)
    gen_streamfit("streamfit_k5_earlystop", n=50, m=70, k=5, density=0.15, seed=210, n_iter=100,
                  n_iter_per_test=10, tol=1e-3, block_size=65536)
    gen_streamfit("streamfit_k8_thresh", n=40, m=64, k=8, density=0.2, seed=230, n_iter=15, n_iter_per_test=10,
                  tol=0.0, block_size=77, thresh=2e-3)
    # k = 1: the log-likelihood stops changing; plsa.py stops through `change == 0`, this loop runs on
    gen_streamfit("streamfit_k1_zero_change", n=30, m=40, k=1, density=0.2, seed=720, n_iter=25,
                  n_iter_per_test=5, tol=0.0, block_size=64)
    gen_streamrefit("streamrefit_k6", n=40, m=60, k=6, density=0.2, seed=300, n_iter=50, n_iter_per_test=5,
                    tol=0.001, block_size=96)
    gen_streamrefit("streamrefit_k8_weighted_thresh", n=30, m=50, k=8, density=0.2, seed=310, n_iter=20,
                    n_iter_per_test=10, tol=0.005, block_size=65536, thresh=2e-3, weighted=True)
    gen_streamestimator("streamestimator_int_emptyrows", seed=420)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":       # only the slow one
        gen_fit_big("fit_k4_big")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cfg1":      # the slowest one; needs the downloaded corpus
        gen_fit_cfg1()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "nndsvd":
        gen_nndsvd("init_nndsvd_k6", seed=1000)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stream":    # only the streamed_plsa.py fixtures
        gen_stream_all()
        sys.exit(0)
    gen_kernels("kernels_k6", n=40, m=50, k=6, density=0.15, seed=100, thresh=1e-32)
    gen_kernels("kernels_k8_thresh", n=36, m=44, k=8, density=0.2, seed=110, thresh=2.5e-3, zero_doc=3)
    gen_kernels("kernels_k20", n=64, m=200, k=20, density=0.06, seed=120, thresh=1e-16)
    gen_kernels("kernels_k33", n=20, m=30, k=33, density=0.25, seed=130, thresh=1e-32)

    gen_fit("fit_k8_tol0", n=60, m=90, k=8, density=0.15, seed=200, n_iter=25, n_iter_per_test=10, tol=0.0)
    gen_fit("fit_k5_earlystop", n=50, m=70, k=5, density=0.15, seed=210, n_iter=100, n_iter_per_test=10, tol=1e-3)
    gen_fit("fit_k4_weighted", n=50, m=60, k=4, density=0.2, seed=220, n_iter=21, n_iter_per_test=5, tol=0.0, weighted=True)
    gen_fit("fit_k8_thresh", n=40, m=64, k=8, density=0.2, seed=230, n_iter=15, n_iter_per_test=10, tol=0.0, thresh=2e-3)
    gen_fit("fit_k6_tupleinit", n=40, m=50, k=6, density=0.2, seed=240, n_iter=12, n_iter_per_test=4, tol=0.0, tuple_init=True)
    gen_fit("fit_k16_mid", n=300, m=400, k=16, density=0.05, seed=250, n_iter=30, n_iter_per_test=10, tol=0.0)
    gen_fit("fit_k20_50it", n=120, m=300, k=20, density=0.05, seed=260, n_iter=50, n_iter_per_test=10, tol=0.0)

    gen_refit("refit_k6", n=40, m=60, k=6, density=0.2, seed=300, n_iter=50, n_iter_per_test=5, tol=0.001)
    gen_refit("refit_k8_weighted", n=30, m=50, k=8, density=0.2, seed=310, n_iter=20, n_iter_per_test=10, tol=0.005, weighted=True)

    gen_estimator("estimator_int", "int", empty_rows=(), seed=400)
    gen_estimator("estimator_float", "float", empty_rows=(), seed=410)
    gen_estimator("estimator_int_emptyrows", "int", empty_rows=(0, 17, 47), seed=420)

    gen_member("member_k6", seed=500)

    gen_metrics("metrics", seed=600)

    gen_blockfit("blockfit_k6", n=70, m=90, k=6, density=0.12, seed=700, n_iter=25, n_iter_per_test=10, tol=0.0)
    gen_blockfit("blockfit_k5_earlystop", n=50, m=70, k=5, density=0.15, seed=210, n_iter=100, n_iter_per_test=10, tol=1e-3)
    # k = 1 converges in one step: the log-likelihood stops changing, plsa.py would stop through its
    # `change == 0` arm, the blockwise loop (tolerance 0) runs all iterations
    gen_blockfit("blockfit_k1_zero_change", n=30, m=40, k=1, density=0.2, seed=720, n_iter=25, n_iter_per_test=5, tol=0.0)

    gen_combine("combine_t24", seed=800)

    gen_stream_all()

    gen_nndsvd("init_nndsvd_k6", seed=1000)

    gen_fit_inner_ll_only("fit_inner_ll_only_weights", seed=900)

    gen_fit_big("fit_k4_big")
