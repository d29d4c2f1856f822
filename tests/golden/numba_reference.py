#!/opt/conda/bin/python3.9
"""The reference AS ITS USERS RUN IT -- enstop/plsa.py compiled by numba (fastmath, parallel prange) -- executed in the
build container (round 5).  Rounds 1-4 could only run the reference's source as plain Python under a no-op `numba` shim
(make_golden.py: sequential, no fastmath); SURVEY.md section 8c records numba as unusable here.  It is usable after all:
conda's python3.9 carries numba 0.54.1 / llvmlite 0.37, whose import fails on (a) a NumPy <= 1.20 gate, (b) the ufunc C
extension `_internal` (built for an older NumPy ABI; only @vectorize needs it) and (c) a few attributes NumPy removed since
(np.MachAr, np.bool, ...).  `numba_env()` below steps over the three; @njit with parallel=True / fastmath=True then
compiles and runs the reference unchanged.  BUILD CONTAINER ONLY (needs /root/reference and /opt/conda); what it writes
is data: inputs are the committed corpus of fit_cfg1_shape.npz, outputs are what the compiled reference computed.

    /opt/conda/bin/python3.9 tests/golden/numba_reference.py fixture     -> tests/golden/numba_cfg1.npz
    /opt/conda/bin/python3.9 tests/golden/numba_reference.py small       -> tests/golden/numba_small.npz
    /opt/conda/bin/python3.9 tests/golden/numba_reference.py blocks      -> tests/golden/numba_block_streamed.npz
    /opt/conda/bin/python3.9 tests/golden/numba_reference.py refit       -> tests/golden/numba_cfg1_refit.npz
    /opt/conda/bin/python3.9 tests/golden/numba_reference.py at_scale 2 gpurun_out/cfg2_corpus.npz         -> numba_cfg2.npz
    /opt/conda/bin/python3.9 tests/golden/numba_reference.py at_scale 3 gpurun_out/cfg3_150k_corpus.npz    -> numba_cfg3_sample.npz
    /opt/conda/bin/python3.9 tests/golden/numba_reference.py fuzz [N]    -> compiled reference vs oracle/plsa_oracle.c
    /opt/conda/bin/python3.9 tests/golden/numba_reference.py time        -> compiled reference vs the C port, same cores
"""
import json
import os
import sys
import time
import types
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def numba_env():
    warnings.filterwarnings("ignore")
    import numpy as np
    fake = types.ModuleType("numba.np.ufunc._internal")
    fake.PyUFunc_None, fake.PyUFunc_Zero, fake.PyUFunc_One, fake.PyUFunc_ReorderableNone = -1, 0, 1, -2

    class _DUFunc:
        def __init__(self, *a, **k):
            raise RuntimeError("ufunc support is stubbed out (only @vectorize needs it)")
    fake._DUFunc = _DUFunc
    fake.fromfunc = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    sys.modules["numba.np.ufunc._internal"] = fake

    class _MachAr:
        pass
    for name, val in (("MachAr", _MachAr), ("bool", bool), ("int", int), ("float", float), ("complex", complex),
                      ("object", object), ("str", str), ("long", int), ("unicode", str)):
        if name not in np.__dict__:
            setattr(np, name, val)
    real = np.__version__
    np.__version__ = "1.20.3"
    import numba
    np.__version__ = real
    pkg = types.ModuleType("enstop")                 # enstop/__init__.py imports dask / hdbscan / umap: not executed
    pkg.__path__ = ["/root/reference/enstop"]
    sys.modules["enstop"] = pkg
    import enstop.utils  # noqa: F401
    import enstop.plsa as ref
    return numba, ref


def load_cfg1():
    import numpy as np
    import scipy.sparse as sp
    g = np.load(os.path.join(HERE, "fit_cfg1_shape.npz"))
    indptr = g["indptr"].astype(np.int64)
    c = np.cumsum(g["indices_rowdelta"].astype(np.int64))
    before = np.concatenate([[0], c])[indptr[:-1]]
    indices = (c - np.repeat(before, np.diff(indptr))).astype(np.int32)
    X = sp.csr_matrix((g["data_u8"].astype(np.float32), indices, g["indptr"]), shape=tuple(int(v) for v in g["shape"]))
    return X, g


def oracle(variant, threads):
    from oracle.plsa_oracle import Oracle
    o = Oracle(variant=variant)
    o.set_threads(threads)
    return o


def peak_rel(a, b):
    import numpy as np
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


def fit_with_trace(ref, numba, X, k, sw, n_iter, n_iter_per_test, tolerance, thresh, seed):
    """plsa_fit of the compiled reference + the log-likelihoods it tested (recomputed with its own compiled
    log_likelihood on prefixes of the run: the reference does not return them) and the iteration it stopped at."""
    import numpy as np
    from sklearn.utils import check_random_state
    rng = check_random_state(seed)
    U0, V0 = ref.plsa_init(X, k, "random", rng=rng)
    U0 = U0.astype(np.float32, order="C"); V0 = V0.astype(np.float32, order="C")
    A = X.tocoo().astype(np.float32)
    r, c, v = A.row, A.col, A.data
    # the reference's own loop (plsa_fit_inner) is one compiled function: the stop iteration is read off by running it
    # with n_iter = 1, 2, ... only where a test happens -- cheaper: replicate its loop around its compiled kernels
    P = np.zeros((v.shape[0], k), np.float32)
    npw = np.zeros(k, np.float32); npd = np.zeros(X.shape[0], np.float32)
    U, V = U0.copy(), V0.copy()
    prev = ref.log_likelihood(r, c, v, V, U, sw)
    trace = [float(prev)]
    done = 0
    for i in range(n_iter):
        ref.plsa_e_step(r, c, v, V, U, P, np.float32(thresh))
        ref.plsa_m_step(r, c, v, V, U, P, npw, npd)
        done += 1
        if i % n_iter_per_test == 0:
            cur = ref.log_likelihood(r, c, v, V, U, sw)
            trace.append(float(cur))
            change = np.abs(cur - prev)
            if change == 0 or change / np.abs(cur) < tolerance:
                break
            prev = cur
    # cross-check against the reference's OWN compiled driver (same kernels, its own loop)
    U2, V2 = ref.plsa_fit_inner(r, c, v, V0.copy(), U0.copy(), sw, n_iter, n_iter_per_test, tolerance, np.float32(thresh), False)
    return dict(U=U, V=V, trace=np.array(trace, np.float32), iters=done, U_inner=U2, V_inner=V2, U0=U0, V0=V0, coo=(r, c, v))


def cmd_fixture():
    import numpy as np
    numba, ref = numba_env()
    X, g = load_cfg1()
    n, m = X.shape
    k = 20
    sw = np.ones(n, np.float32)
    out = {"numba_version": numba.__version__, "threads_available": numba.config.NUMBA_NUM_THREADS}
    report = {}
    # (1) BASELINE configs[0]: 50 EM iterations, tolerance 0 -- on all threads (what a user gets) and on one thread
    for threads in (numba.config.NUMBA_NUM_THREADS, 1):
        numba.set_num_threads(threads)
        t0 = time.time()
        res = fit_with_trace(ref, numba, X, k, sw, 50, 10, 0.0, 1e-32, 42)
        dt = time.time() - t0
        tag = "t%d" % threads
        assert res["iters"] == 50
        # its own driver gives what the replicated loop gives (same compiled kernels); prange may reorder the likelihood only
        report["fit50_%s" % tag] = {"seconds": round(dt, 2), "driver_vs_replicated_loop_U": peak_rel(res["U_inner"], res["U"]),
                                    "driver_vs_replicated_loop_V": peak_rel(res["V_inner"], res["V"])}
        if threads != 1:
            cols = g["V_cols"]
            out.update(U50=res["U"], V50_sample=res["V"][:, cols], V50_rowsum64=res["V"].astype(np.float64).sum(axis=1),
                       ll50=res["trace"], V_cols=cols, fit_seed=np.int64(42), k=np.int64(k), threads=np.int64(threads))
            U0, V0, (r, c, v) = res["U0"], res["V0"], res["coo"]
        else:
            out.update(U50_one_thread_vs_all=np.float64(peak_rel(res["U"], out["U50"])), ll50_one_thread=res["trace"])
    # (2) the default tolerance (PLSA(): n_iter 100, test every 10, tolerance 1e-3): where the compiled reference stops
    stops = {}
    for threads in (1, 2, 4, numba.config.NUMBA_NUM_THREADS):
        numba.set_num_threads(threads)
        res = fit_with_trace(ref, numba, X, k, sw, 100, 10, 1e-3, 1e-32, 42)
        stops[str(threads)] = int(res["iters"])
        out["ll_default_tol_t%d" % threads] = res["trace"]
    out["stop_iteration_threads"] = np.array(sorted(int(t) for t in stops), np.int64)
    out["stop_iteration"] = np.array([stops[str(t)] for t in sorted(int(t) for t in stops)], np.int64)
    report["default_tolerance_stop_iteration_by_threads"] = stops
    # (3) the oracle builds on the same problem, for the record kept next to the fixture
    for variant in ("strict", "n64", "wide"):
        o = oracle(variant, 8)
        Uo, Vo, tr, it = o.plsa_fit_inner(r, c, v, V0.copy(), U0.copy(), sw, n_iter=50, n_iter_per_test=10, tolerance=0.0,
                                          e_step_thresh=1e-32, return_trace=True)
        report["numba_reference_vs_oracle_%s" % variant] = {
            "U": peak_rel(out["U50"], Uo), "V": peak_rel(out["V50_sample"], Vo[:, out["V_cols"]]),
            "ll_rel": float(np.max(np.abs(out["ll50"].astype(np.float64) - tr) / np.abs(tr)))}
    np.savez_compressed(os.path.join(HERE, "numba_cfg1.npz"), **out)
    print(json.dumps(report, indent=1))
    json.dump(report, open(os.path.join(ROOT, "profiles", "r05_numba_reference_cfg1.json"), "w"), indent=1)


SMALL = ("fit_k8_tol0", "fit_k20_50it", "fit_k16_mid", "fit_k4_big", "fit_k4_weighted", "fit_k5_earlystop", "fit_k8_thresh")


def cmd_small():
    """The committed small fixtures (inputs + what the reference's SOURCE computes when run sequentially, make_golden.py)
    re-run through the COMPILED reference from the same initial factors -> tests/golden/numba_small.npz: its factors, the
    iteration it stops at, and how far compilation (fastmath, SIMD reductions) moved it from the sequential result."""
    import numpy as np
    import scipy.sparse as sp
    numba, ref = numba_env()
    numba.set_num_threads(numba.config.NUMBA_NUM_THREADS)
    out = {"numba_version": numba.__version__}
    for case in SMALL:
        g = np.load(os.path.join(HERE, case + ".npz"))
        data = g["data"] if "data" in g.files else g["data_u8"].astype(np.float32)
        X = sp.csr_matrix((data.astype(np.float32), g["indices"], g["indptr"]), shape=tuple(int(v) for v in g["shape"]))
        A = X.tocoo().astype(np.float32)
        r, c, v = A.row, A.col, A.data
        sw = g["sw"].astype(np.float32)
        use_sw = bool(np.any(sw != 1.0))                                   # plsa.py:712
        n_iter, per, tol, thresh = int(g["n_iter"]), int(g["n_iter_per_test"]), float(g["tol"]), np.float32(g["thresh"])
        U, V = ref.plsa_fit_inner(r, c, v, g["V0"].copy(), g["U0"].copy(), sw, n_iter, per, tol, thresh, use_sw)
        # where it stopped: the same compiled kernels driven one iteration at a time
        P = np.zeros((v.shape[0], int(g["k"])), np.float32)
        npw = np.zeros(int(g["k"]), np.float32); npd = np.zeros(X.shape[0], np.float32)
        U2, V2 = g["U0"].copy(), g["V0"].copy()
        prev = ref.log_likelihood(r, c, v, V2, U2, sw)
        done = 0
        for i in range(n_iter):
            ref.plsa_e_step(r, c, v, V2, U2, P, thresh)
            if use_sw:
                ref.plsa_m_step_w_sample_weight(r, c, v, V2, U2, P, sw, npw, npd)
            else:
                ref.plsa_m_step(r, c, v, V2, U2, P, npw, npd)
            done += 1
            if i % per == 0:
                cur = ref.log_likelihood(r, c, v, V2, U2, sw)
                change = np.abs(cur - prev)
                if change == 0 or change / np.abs(cur) < tol:
                    break
                prev = cur
        same = np.array_equal(U2, U) and np.array_equal(V2, V)
        # (an in-range threshold, fit_k8_thresh: the compiled DRIVER, with its kernels inlined under fastmath, rounds a
        #  borderline `v > thresh` differently from the same kernels compiled on their own -- the compiled reference is not
        #  bit-consistent with itself there; the driver's result is the one stored)
        assert same or tol == 0.0, case
        out[case + "__driver_vs_separately_compiled_kernels"] = np.float64(max(peak_rel(U2, U), peak_rel(V2, V)))
        out[case + "__U"], out[case + "__V"], out[case + "__iters"] = U, V, np.int64(done)
        out[case + "__dev_from_sequential_U"] = np.float64(peak_rel(U, g["U"]))
        out[case + "__dev_from_sequential_V"] = np.float64(peak_rel(V, g["V"]))
        print("%-18s iterations %3d (sequential: %3d)   compiled vs sequential source: U %.2e  V %.2e"
              % (case, done, int(g["iters"]), out[case + "__dev_from_sequential_U"], out[case + "__dev_from_sequential_V"]))
    np.savez_compressed(os.path.join(HERE, "numba_small.npz"), **out)


def cmd_fuzz(cases=300, seed=1):
    """compiled reference (fastmath, parallel) vs the strict oracle on random small problems: how far does COMPILATION
    move the reference from its own sequential semantics?"""
    import numpy as np
    import scipy.sparse as sp
    numba, ref = numba_env()
    o = oracle("strict", 1)
    rs = np.random.RandomState(seed)
    worst = dict(bit_identical=0)
    devs, devs_plain, devs_edge = [], [], []
    t0 = time.time()
    for case in range(cases):
        n = int(rs.randint(2, 400)); m = int(rs.randint(2, 500)); k = int(rs.choice([1, 2, 3, 5, 8, 13, 20, 32, 40]))
        X = sp.random(n, m, density=float(rs.choice([0.02, 0.1, 0.3])), format="csr", random_state=rs, dtype=np.float32)
        X.data = np.ceil(X.data * 5).astype(np.float32)
        if X.nnz == 0:
            continue
        sw = np.ones(n, np.float32)
        kw = dict(n_iter=int(rs.randint(1, 40)), n_iter_per_test=int(rs.randint(1, 12)), tolerance=float(rs.choice([0.0, 1e-4, 1e-2])),
                  e_step_thresh=float(rs.choice([1e-32, 1e-16, 1e-6])), random_state=int(rs.randint(1 << 30)))
        numba.set_num_threads(int(rs.choice([1, 8])))
        U, V = ref.plsa_fit(X, k, sw, **kw)
        Uo, Vo, tr, it = o.plsa_fit(X, k, sw, return_trace=True, **kw)
        dev = max(peak_rel(U, Uo), peak_rel(V, Vo))
        devs.append(dev)
        edge = kw["tolerance"] > 0 or kw["e_step_thresh"] > 1e-10     # a stop test or a threshold that can sit on the edge
        (devs_edge if edge else devs_plain).append(dev)
        worst["bit_identical"] += int(np.array_equal(U, Uo) and np.array_equal(V, Vo))
    pct = lambda a, q: float(np.percentile(a, q)) if len(a) else 0.0
    line = ("compiled (numba %s, fastmath, parallel) reference vs strict oracle: %d random problems, seed %d, %.0f s: "
            "peak-relative deviation of the factors median %.1e / 99 %% %.1e / max %.1e; tolerance 0 and threshold <= 1e-16 "
            "(%d cases): max %.1e; with a live stop test or a threshold of 1e-6 (%d cases): 99 %% %.1e / max %.1e; "
            "bit-identical in %d cases" % (numba.__version__, cases, seed, time.time() - t0, pct(devs, 50), pct(devs, 99),
                                           max(devs), len(devs_plain), max(devs_plain + [0.0]), len(devs_edge),
                                           pct(devs_edge, 99), max(devs_edge + [0.0]), worst["bit_identical"]))
    print(line)


def cmd_time():
    """EM iterations/s of the compiled reference and of the C port (the bench's cpu_baseline) on THE SAME cores and corpus."""
    import numpy as np
    import scipy.sparse as sp
    numba, ref = numba_env()
    cores = numba.config.NUMBA_NUM_THREADS
    numba.set_num_threads(cores)
    fast = oracle("fast", cores)
    rows = []

    def corpus(n, m, nnz, seed=0):                 # independent Zipf tokens, like plsa_generate_synthetic (host NumPy)
        rs = np.random.RandomState(seed)
        p = np.arange(1, m + 1, dtype=np.float64) ** -1.07
        cdf = np.cumsum(p / p.sum())
        tokens = int(nnz * 1.25)
        d = rs.randint(0, n, size=tokens)
        w = rs.permutation(m)[np.minimum(np.searchsorted(cdf, rs.rand(tokens)), m - 1)]
        X = sp.coo_matrix((np.ones(tokens, np.float32), (d, w)), shape=(n, m)).tocsr()
        X.sum_duplicates()
        return X
    X1, _ = load_cfg1()
    for name, X, k, iters in (("config 1 (the committed corpus)", X1, 20, 10),
                              ("config 2 shape (100k x 50k, ~10M nnz, host-made corpus)", corpus(100_000, 50_000, 10_000_000), 32, 3),
                              ("config 3 shape, first 250 000 documents (~25M nnz, host-made corpus)", corpus(250_000, 100_000, 25_000_000), 64, 2)):
        n = X.shape[0]
        sw = np.ones(n, np.float32)
        A = X.tocoo().astype(np.float32)
        r, c, v = A.row.astype(np.int32), A.col.astype(np.int32), A.data
        rng = np.random.RandomState(42)
        U0, V0 = ref.plsa_init(X, k, "random", rng=rng)
        U0 = U0.astype(np.float32); V0 = V0.astype(np.float32)
        ref.plsa_fit_inner(r, c, v, V0.copy(), U0.copy(), sw, 1, 10, 0.0, np.float32(1e-32), False)      # JIT / warm
        t0 = time.time()
        ref.plsa_fit_inner(r, c, v, V0.copy(), U0.copy(), sw, iters, 10, 0.0, np.float32(1e-32), False)
        t_ref = time.time() - t0
        t0 = time.time()
        fast.plsa_fit_inner(r, c, v, V0.copy(), U0.copy(), sw, n_iter=iters, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32)
        t_port = time.time() - t0
        rows.append({"workload": name, "nnz": int(X.nnz), "k": k, "iterations": iters, "cores": cores,
                     "numba_reference_iter_per_s": round(iters / t_ref, 4), "c_port_iter_per_s": round(iters / t_port, 4),
                     "port_over_reference": round(t_ref / t_port, 3)})
        print(json.dumps(rows[-1]), flush=True)
    json.dump(rows, open(os.path.join(ROOT, "profiles", "r05_numba_reference_vs_c_port_timing.json"), "w"), indent=1)


def blocks_corpus():
    """Mid-size corpus for the block-parallel / streamed modules: 12 000 documents x 6 000 words, ~60 distinct words per
    document drawn from a Zipf-like law, counts 1..4.  Built from NumPy's legacy RandomState only (stable across versions)."""
    import numpy as np
    import scipy.sparse as sp
    rs = np.random.RandomState(2026)
    n, m, per = 12000, 6000, 64
    w = (m * rs.rand(n * per) ** 3.0).astype(np.int64)                   # heavy head, long tail
    perm = rs.permutation(m)
    rows = np.repeat(np.arange(n), per)
    vals = rs.randint(1, 5, n * per).astype(np.float32)
    X = sp.csr_matrix((np.ones_like(vals), (rows, perm[w])), shape=(n, m))
    X.sum_duplicates()                                                   # distinct words per document
    X.sort_indices()
    X.data = vals[:X.nnz].copy()
    return X


def cmd_blocks():
    """enstop/block_parallel_plsa.py and enstop/streamed_plsa.py COMPILED BY NUMBA on a corpus large enough for their own
    tilings to matter (8 x 8 tiles of 1 500 x 750; 11 blocks of 65 536 non-zeros) -> tests/golden/numba_block_streamed.npz."""
    import numpy as np
    numba, ref = numba_env()
    import enstop.block_parallel_plsa as bp
    import enstop.streamed_plsa as st
    X = blocks_corpus()
    n = X.shape[0]
    k = 16
    kw = dict(n_iter=30, n_iter_per_test=10, tolerance=0.0, random_state=17)
    numba.set_num_threads(min(8, numba.config.NUMBA_NUM_THREADS))
    t0 = time.time()
    Ub, Vb = bp.plsa_fit(X, k, n_row_blocks=8, n_col_blocks=8, **kw)
    t_b = time.time() - t0
    sw = np.ones(n, np.float32)
    t0 = time.time()
    Us, Vs = st.plsa_fit(X, k, sw, block_size=65536, **kw)
    t_s = time.time() - t0
    sww = np.exp(np.random.RandomState(4).uniform(-1, 1, n)).astype(np.float32)
    Usw, Vsw = st.plsa_fit(X, k, sww, block_size=65536, **kw)
    Up, Vp = ref.plsa_fit(X, k, sw, **kw)
    held = X[::7]
    Ut = st.plsa_refit(held, Vs, np.ones(held.shape[0], np.float32), block_size=65536, n_iter=20, n_iter_per_test=5,
                       tolerance=0.0, random_state=42)
    out = os.path.join(HERE, "numba_block_streamed.npz")
    np.savez_compressed(out, indptr=X.indptr.astype(np.int32), indices=X.indices.astype(np.int32), data_u8=X.data.astype(np.uint8),
                        shape=np.array(X.shape), k=k, n_iter=30, n_iter_per_test=10, fit_seed=17,
                        U_block=Ub[::4], V_block=Vb, U_streamed=Us[::4], V_streamed=Vs, U_streamed_weighted=Usw[::4],
                        V_streamed_weighted=Vsw, sample_weight_seed=4, u_stride=4, held_stride=7, U_streamed_refit=Ut[::2],
                        streamed_equals_plsa_bitwise=bool(np.array_equal(Us, Up) and np.array_equal(Vs, Vp)),
                        block_vs_plsa=np.array([peak_rel(Ub, Up), peak_rel(Vb, Vp)]))
    rep = {"shape": [int(v) for v in X.shape], "nnz": int(X.nnz), "k": k, "threads": int(numba.get_num_threads()),
           "block_parallel_fit_s": round(t_b, 2), "streamed_fit_s": round(t_s, 2),
           "block_vs_plsa": [peak_rel(Ub, Up), peak_rel(Vb, Vp)], "streamed_vs_plsa": [peak_rel(Us, Up), peak_rel(Vs, Vp)],
           "fixture_bytes": os.path.getsize(out)}
    print(json.dumps(rep))
    with open(os.path.join(ROOT, "profiles", "r05_numba_reference_block_streamed.json"), "w") as f:
        json.dump(rep, f, indent=1)


def refit_topics(k, m):
    """A topic matrix both sides can rebuild bit for bit (no stored array): small integers, rows divided by their exact sums."""
    import numpy as np
    w = np.arange(m, dtype=np.int64)[None, :]
    z = np.arange(k, dtype=np.int64)[:, None]
    T = ((w * 7 + z * 131) % 97 + 1).astype(np.float64)
    return (T / T.sum(axis=1, keepdims=True)).astype(np.float32)      # integer sums: exact in float64 in any order


def cmd_refit():
    """`plsa_refit` of the reference COMPILED BY NUMBA at BASELINE config 1's exact shape -- PLSA.transform's own call
    (plsa.py:1210-1218: 50 iterations, a test every 5, tolerance 0.001, RandomState(42)) against fixed topics
    -> tests/golden/numba_cfg1_refit.npz (every second row of P(z|d))."""
    import numpy as np
    numba, ref = numba_env()
    X, g = load_cfg1()
    n, m = X.shape
    k = 20
    topics = refit_topics(k, m)
    sw = np.ones(n, np.float32)
    t0 = time.time()
    U = ref.plsa_refit(X, topics, sw, n_iter=50, n_iter_per_test=5, tolerance=0.001, e_step_thresh=1e-32, random_state=42)
    dt = time.time() - t0
    rep = {"seconds": round(dt, 2), "threads": int(numba.get_num_threads())}
    for variant in ("strict", "wide"):
        o = oracle(variant, 8)
        Uo = o.plsa_refit(X, topics, sw, n_iter=50, n_iter_per_test=5, tolerance=0.001, e_step_thresh=1e-32, random_state=42)
        rep["compiled_vs_oracle_" + variant] = peak_rel(U, Uo)
    np.savez_compressed(os.path.join(HERE, "numba_cfg1_refit.npz"), U_every_second_row=U[::2], k=np.int64(k),
                        topics_checksum=np.float64(topics.astype(np.float64).sum()), compiled_vs_strict=np.float64(rep["compiled_vs_oracle_strict"]))
    print(json.dumps(rep))
    json.dump(rep, open(os.path.join(ROOT, "profiles", "r05_numba_reference_cfg1_refit.json"), "w"), indent=1)


def load_dump(path):
    """a corpus downloaded from the GPU box by tools/dump_synthetic_corpus.py (the engine's own generator; its sha256 travels
    with it and is what the GPU test checks the regenerated corpus against)"""
    import hashlib
    import numpy as np
    import scipy.sparse as sp
    g = np.load(path)
    indptr = g["indptr"].astype(np.int64)
    c = np.cumsum(g["indices_rowdelta"].astype(np.int64))
    before = np.concatenate([[0], c])[indptr[:-1]]
    indices = (c - np.repeat(before, np.diff(indptr))).astype(np.int32)
    data = (g["data_u8"] if "data_u8" in g.files else g["data_u16"]).astype(np.float32)
    X = sp.csr_matrix((data, indices, g["indptr"]), shape=tuple(int(v) for v in g["shape"]))
    h = hashlib.sha256()
    for arr in (X.indptr.astype(np.int32), X.indices.astype(np.int32), X.data.astype(np.float32)):
        h.update(np.ascontiguousarray(arr).tobytes())
    assert h.hexdigest() == str(g["sha256"]), "the dump does not decode to the matrix it was made from"
    return X, g


AT_SCALE = {2: dict(k=32, n_iter=3, name="numba_cfg2", rows=0), 3: dict(k=64, n_iter=2, name="numba_cfg3_sample", rows=150_000)}


def cmd_at_scale(cfg_id, dump):
    """RESULT fixtures of the reference COMPILED BY NUMBA above config 1 (round 6): BASELINE config 2 whole (100 k x 50 k, 10 M
    non-zeros, k = 32, 3 iterations) and config 3's first 150 000 documents (15 M non-zeros over the full 100 k vocabulary,
    k = 64, 2 iterations) -- the engine's own synthetic corpora, downloaded once (tools/dump_synthetic_corpus.py), fitted by the
    compiled reference from RandomState(42) with a likelihood test after every iteration.  Stored: the corpus' sha256 (the
    GPU test regenerates the corpus on the device and checks it), a row sample of P(z|d), a column sample of P(w|z), its
    float64 row sums, the tested log-likelihoods, and for the record how far the compiled run sits from the strict / exact
    oracles.  <= 3 MB each."""
    import numpy as np
    numba, ref = numba_env()
    spec = AT_SCALE[cfg_id]
    X, g = load_dump(dump)
    assert int(g["rows"]) == spec["rows"]
    n, m = X.shape
    k, n_iter = spec["k"], spec["n_iter"]
    sw = np.ones(n, np.float32)
    numba.set_num_threads(numba.config.NUMBA_NUM_THREADS)
    t0 = time.time()
    res = fit_with_trace(ref, numba, X, k, sw, n_iter, 1, 0.0, 1e-32, 42)
    dt = time.time() - t0
    assert res["iters"] == n_iter
    rs = np.random.RandomState(7)
    rows = np.sort(rs.choice(n, 5000, replace=False)).astype(np.int32)
    cols = np.sort(rs.choice(m, 4000, replace=False)).astype(np.int32)
    report = {"config": cfg_id, "shape": [n, m], "nnz": int(X.nnz), "k": k, "n_iter": n_iter, "numba_version": numba.__version__,
              "threads": int(numba.config.NUMBA_NUM_THREADS), "seconds_incl_compilation": round(dt, 1),
              "driver_vs_replicated_loop": {"U": peak_rel(res["U_inner"], res["U"]), "V": peak_rel(res["V_inner"], res["V"])}}
    r, c, v = res["coo"]
    for variant in ("strict", "wide"):
        o = oracle(variant, 8)
        Uo, Vo, tr, it = o.plsa_fit_inner(r, c, v, res["V0"].copy(), res["U0"].copy(), sw, n_iter=n_iter, n_iter_per_test=1,
                                          tolerance=0.0, e_step_thresh=1e-32, return_trace=True)
        report["compiled_reference_vs_oracle_%s" % variant] = {
            "U": peak_rel(res["U"], Uo), "V": peak_rel(res["V"], Vo),
            "ll_rel": float(np.max(np.abs(res["trace"].astype(np.float64) - tr) / np.abs(tr)))}
    out = dict(numba_version=np.array(numba.__version__), corpus_sha256=np.array(str(g["sha256"])), corpus_seed=np.int64(g["seed"]),
               corpus_rows=np.int64(spec["rows"]), shape=np.array([n, m], np.int64), nnz=np.int64(X.nnz), k=np.int64(k),
               n_iter=np.int64(n_iter), fit_seed=np.int64(42), U_rows=rows, U_sample=res["U"][rows], V_cols=cols,
               V_sample=res["V"][:, cols], V_rowsum64=res["V"].astype(np.float64).sum(axis=1), ll_trace=res["trace"],
               vs_strict_U=np.float64(report["compiled_reference_vs_oracle_strict"]["U"]),
               vs_strict_V=np.float64(report["compiled_reference_vs_oracle_strict"]["V"]),
               vs_wide_U=np.float64(report["compiled_reference_vs_oracle_wide"]["U"]),
               vs_wide_V=np.float64(report["compiled_reference_vs_oracle_wide"]["V"]),
               vs_wide_ll=np.float64(report["compiled_reference_vs_oracle_wide"]["ll_rel"]))
    path = os.path.join(HERE, spec["name"] + ".npz")
    np.savez_compressed(path, **out)
    report["fixture_bytes"] = os.path.getsize(path)
    print(json.dumps(report, indent=1))
    json.dump(report, open(os.path.join(ROOT, "profiles", "r06_%s.json" % spec["name"]), "w"), indent=1)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "fixture"
    if what == "fixture":
        cmd_fixture()
    elif what == "fuzz":
        cmd_fuzz(int(sys.argv[2]) if len(sys.argv) > 2 else 300, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    elif what == "time":
        cmd_time()
    elif what == "small":
        cmd_small()
    elif what == "blocks":
        cmd_blocks()
    elif what == "refit":
        cmd_refit()
    elif what == "at_scale":
        cmd_at_scale(int(sys.argv[2]), sys.argv[3])
