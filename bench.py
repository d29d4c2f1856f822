#!/usr/bin/env python3
"""bench.py -- EM iterations/sec of the pLSA hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--schedule fused|materialised] [--topics 64]
    python bench.py --gpus 8 --config 4        BASELINE configs[3]: the 32-member ensemble on the 20NG shape over 8 GPUs

A "step" is one EM iteration (E-step + M-step, plus the log-likelihood test at the reference's
schedule, plsa.py:630) over the whole synthetic corpus, driven through the C ABI (plsa_fit).
N = 1 : one fit on the corpus.   N > 1 : the ensemble path -- one process per GPU, rank r fits
bootstrap member r of the same corpus (device-side row gather), no collective on the EM path, one RCCL
all-gather of the topic matrices at the end (inside the timed region) -- the exchange the product itself
uses: the member's topics go into the engine's device stack and enstop_amd.distributed's gather
(plsa_comm_allgather_stack: one grouped ncclAllGather from the C ABI + one copy to pinned host memory; no
PyTorch in the process).  `value` is the whole-job aggregate: (N * K) EM iterations / max-over-ranks wall time.

`ensemble` (every N): a MEASURED ensemble through the product's own call -- enstop_amd.ensemble_of_topics on the
host copy of the same corpus, four members per rank (after one untimed warm-up member), 50 EM iterations each: upload, bootstrap row gather, CSC /
item build (+ boundary tuning), MT19937 initialisation on the device, fit, stack, gather -- wall clock between
two barriers, max over ranks -> `ensemble.fits_per_min`, with each rank's fit / gather / RCCL-init seconds.

Launching N > 1: either an external launcher that sets RANK / LOCAL_RANK / WORLD_SIZE (the driver's
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`), or plain
`python bench.py --gpus N`, which spawns the N ranks itself.  Either way the ranks meet through a file
that carries the 128-byte RCCL unique id (enstop_amd/comm.py).  If fewer than N GPUs are visible, or RCCL
cannot be brought up, the run FAILS (non-zero exit status, no JSON line); `--exchange files` is a test
mode for single-GPU boxes (ranks share the GPU, host-file exchange) and says so in the JSON.

Inputs are generated in HBM before the timed region (plsa_generate_synthetic); factors are
initialised on the host exactly as plsa_init does and uploaded before the timed region.

`value` is the MEDIAN of three back-to-back timed regions of exactly K iterations each (every region bracketed by
barrier + synchronize, max over ranks); all three are listed in `timed_regions` (round 3: one of eight fresh
processes ended 6 % low after an unlucky boundary tuning -- one region must not become the driver's number).

Extra objects in the JSON line:
  roofline      the per-nnz materialising E-step kernel (plsa.py:39-107; the kernel the north_star
                roofline target names): algorithmic bytes (SURVEY.md section 8d / DESIGN.md) / average
                launch duration, HIP events on the engine's stream over a second timed leg that runs
                the reference's own kernel sequence (E-step -> M-step -> LL) on the same data
  roofline_dominant_fused   same figures for the dominant kernel of the main (fused) timed region
  cpu_baseline  the CPU port (oracle/plsa_oracle.c, -O3 -ffast-math, OpenMP, reference thread
                structure) timed on the WHOLE corpus (2 EM iterations, "sampled": false; --cpu-baseline-sampled: a
                bounded row sample, extrapolated), rank 0 / N = 1 only; next to it `whole_config2`: the same port on
                the WHOLE of BASELINE configs[1]
  other_configs (N = 1, --config 3) every other BASELINE.json configuration in compact form, each with steps /
                ms_per_step: config1, config2, config5 (EM iterations/s + k_e_step roofline fraction),
                ensemble_20ng_shape (configs[3]: 32 members through enstop_amd.ensemble_of_topics, fits/min),
                ensemble_topics_estimator_20ng_shape (configs[3] through EnsembleTopics itself, planted topics: wall + outcome),
                config3_topical (config 3's shape on a corpus WITH co-occurrence structure)
  hot_kernels   per hot kernel: VGPRs / waves per SIMD / LDS (hipcc's own remarks, captured when the library was
                built: enstop_amd/kernel_resources.json) and traffic_ratio = counter bytes / algorithmic bytes
  roofline.traffic   HBM-side bytes per launch from rocprofv3 PMC passes run by this script itself
                (FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled per MI355X_MICROARCH.md)
                when rocprofv3 is available; otherwise the committed profile's figure, labelled as such
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs (index = position in `configs`, 1-based like SURVEY.md section 8)
CONFIGS = {
    1: dict(n=18_846, m=173_762, nnz=2_950_000, k=20, name="20NG-shaped synthetic CSR 18846 x 173762, 2.95M nnz, k=20"),
    2: dict(n=100_000, m=50_000, nnz=10_000_000, k=32, name="synthetic CSR 100k docs x 50k vocab, 10M nnz, k=32"),
    3: dict(n=1_000_000, m=100_000, nnz=100_000_000, k=64, name="synthetic CSR 1M docs x 100k vocab, 100M nnz, k=64"),
    # configs[3]: "EnsembleTopics n_components=20, n_runs=32 on 20-Newsgroups, sharded across 8 GPUs" -- config 1's corpus;
    # with --config 4 the measured ensemble leg fits 32 members in all (32 / N per rank)
    4: dict(n=18_846, m=173_762, nnz=2_950_000, k=20, ensemble_runs=32,
            name="20NG-shaped synthetic CSR 18846 x 173762, 2.95M nnz, k=20; ensemble of 32 bootstrapped fits"),
    5: dict(n=5_000_000, m=200_000, nnz=500_000_000, k=128, name="synthetic CSR 5M docs x 200k vocab, 500M nnz, k=128"),
}
TOPICAL = dict(topics=64, alpha=0.1, background=0.25)      # --topics: documents as Dirichlet mixtures of latent topics
# the 20-Newsgroups stand-in of configs 1 / 4 (round 6): the dataset cannot be obtained here, and real text has co-occurrence
# structure that independent Zipf tokens lack (their gathers hit the L2 1.6x more often) -- twenty latent topics, like the
# twenty newsgroups.  Config 3's headline stays on SURVEY.md 8d's generator (independent tokens) with `config3_topical` beside it.
TOPICAL_20NG = dict(topics=20, alpha=0.1, background=0.25)


def corpus_kind(kw):
    if not kw or not kw.get("topics"):
        return "independent Zipf(1.07) tokens (SURVEY.md 8d generator)"
    return "topical: Dirichlet(%g) mixtures of %d latent topics, %g background (plsa_generate_synthetic_topics)" % (
        kw.get("alpha", 0.1), kw["topics"], kw.get("background", 0.25))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
FITS_ITERS = 50                # EM iterations per ensemble member when quoting fits/min
# How the timed C port relates to the reference itself (measured once, in the build container, where the numba-compiled
# reference can run -- tests/golden/numba_reference.py time; it cannot travel to the GPU box): same 8 cores, same corpora
PORT_CALIBRATION = ("the numba-compiled reference (enstop/plsa.py, fastmath + parallel) runs at 0.68-0.79x the rate of this C "
                    "port on the same 8 cores of the build container (config 1: 2.03 vs 2.59 iterations/s; config-2 shape "
                    "0.52 vs 0.76; config-3 shape on 250 000 documents 0.066 vs 0.090): "
                    "profiles/r05_numba_reference_vs_c_port_timing.json -- the port is the FASTER side, the GPU / CPU "
                    "ratio quoted from it is conservative")


def algorithmic_bytes(kind, n, m, nnz, k):
    """Compulsory HBM traffic per launch, each array counted once."""
    if kind == "e_step":        # indptr + indices + P write + U,V read
        return 4 * (n + 1) + 4 * nnz + 4 * k * nnz + 4 * k * (n + m)
    if kind == "m_step_p":      # indptr + indices + data + P read + U,V write
        return 4 * (n + 1) + 8 * nnz + 4 * k * nnz + 4 * k * (n + m)
    if kind == "loglik":
        return 4 * (n + 1) + 8 * nnz + 4 * k * (n + m)
    if kind == "fused":         # CSR once, factors read once and written once
        return 4 * (n + 1) + 8 * nnz + 8 * k * (n + m)
    if kind == "fused_col":     # CSC copy once, factors read once, V written once
        return 4 * (m + 1) + 8 * nnz + 4 * k * (n + m) + 4 * k * m
    raise KeyError(kind)


KERNEL_KIND = {
    "k_e_step": "e_step", "k_row_pass<P>": "m_step_p", "k_col_pass<P>": "m_step_p", "k_loglik": "loglik",
    "k_row_pass<fused>": "fused", "k_row_pass<fused,LL>": "fused", "k_col_pass<fused>": "fused_col",
}


def init_factors(n, m, k, seed):
    """plsa_init(random) + float32 casts, plsa.py:455-456, 510-511, 709-710."""
    from enstop_amd.plsa import plsa_init

    class S:
        shape = (n, m)
    U, V = plsa_init(S, k, init="random", rng=np.random.RandomState(seed))
    return U.astype(np.float32), V.astype(np.float32)


def cpu_baseline(eng, cfg, k, budget_cells=1.2e9, iters=3, whole=False):
    """Reference-structure CPU port (the only place of this script that touches oracle/), all host cores.
    whole=False: on the first rows of the corpus resident on `eng`, bounded to ~budget_cells cells per
    iteration, scaled by the nnz fraction ("sampled": true).  whole=True: on the whole corpus, no scaling."""
    from oracle.plsa_oracle import Oracle
    o = Oracle(fast=True)
    cores = os.cpu_count() or 1
    o.set_threads(cores)
    A = eng.download_active_csr()
    n, m = A.shape
    nnz_full = A.nnz
    target_nnz = int(budget_cells / k)
    rows = n if whole else int(min(n, max(1000, np.searchsorted(A.indptr, target_nnz))))
    S = A[:rows]
    Ac = S.tocoo()
    r, c, v = (np.ascontiguousarray(Ac.row, np.int32), np.ascontiguousarray(Ac.col, np.int32),
               np.ascontiguousarray(Ac.data, np.float32))
    U, V = init_factors(rows, m, k, 42)
    sw = np.ones(rows, np.float32)
    t0 = time.perf_counter()
    _, _, _, done = o.plsa_fit_inner(r, c, v, V, U, sw, n_iter=iters, n_iter_per_test=10, tolerance=0.0,
                                     e_step_thresh=1e-32, use_sample_weights=False, return_trace=True)
    dt = time.perf_counter() - t0
    if whole:
        return {"value": round(done / dt, 4), "unit": "iter/s", "cores": cores, "kind": "port", "sampled": False,
                "port_vs_numba_reference": PORT_CALIBRATION,
                "sample": "whole corpus (%d docs, %d nnz), %d EM iterations in %.2f s" % (n, A.nnz, done, dt),
                "gcell_per_s": round(A.nnz * k * done / dt / 1e9, 4)}
    frac = S.nnz / nnz_full
    return {
        "value": round(done / dt * frac, 5), "unit": "iter/s (full-corpus equivalent)", "cores": cores,
        "kind": "port", "sampled": True,
        "sample": "first %d docs (%d nnz = %.3f of the corpus, full %d-word vocabulary), %d EM iterations in %.2f s; "
                  "iterations/s on the sample x nnz fraction" % (rows, S.nnz, frac, m, done, dt),
        "sample_iter_per_s": round(done / dt, 4),
        "gcell_per_s": round(S.nnz * k * done / dt / 1e9, 4),
    }


def quick_config(eng, cfg_id, steps, warmup, seed, with_cpu=False, e_step=True, min_steps=200, device_init=False,
                 corpus_kw=None, pass_times=False):
    """Compact measurement of another BASELINE.json config on the same device (N = 1 only): fused EM
    iterations/s and the materialising E-step's roofline fraction, same method as the main legs.
    device_init: the factors come from the device MT19937 initialisation (the same RandomState(42) stream, bit-identical
    to the host's: config 5's 665 M draws take the host ~10 s).  corpus_kw: generator options (topical corpora).
    pass_times: per-kernel HIP-event averages of the two fused passes over 10 further iterations."""
    from enstop_amd.engine import PLSA_FUSED
    cfg = CONFIGS[cfg_id]
    n, m, k = cfg["n"], cfg["m"], cfg["k"]
    eng.release_scratch()
    nnz = eng.generate_synthetic(n, m, cfg["nnz"], zipf_s=1.07, seed=seed, **(corpus_kw or {}))
    if device_init:
        def reset():
            eng.init_factors_numpy_stream(k, np.random.RandomState(42))
    else:
        U0, V0 = init_factors(n, m, k, 42)

        def reset():
            eng.set_factors(U0, V0)
    reset()
    eng.fit(None, n_iter=warmup, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=PLSA_FUSED)
    eng.synchronize()
    t0 = time.perf_counter()
    steps = max(steps, min_steps)  # a 0.17 ms iteration: 50 of them would be dominated by the call's fixed costs
    it, _ = eng.fit(None, n_iter=steps, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=PLSA_FUSED)
    eng.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": cfg["name"] + ("" if not corpus_kw else " -- TOPICAL corpus %r" % (corpus_kw,)), "corpus": corpus_kind(corpus_kw),
           "nnz": nnz, "k": k,
           "steps": it, "value": round(it / dt, 2), "unit": "iter/s", "ms_per_step": round(dt / it * 1e3, 4),
           "gcell_per_s": round(nnz * k * it / dt / 1e9, 2)}
    if pass_times:
        eng.timing(True)
        eng.timing_reset()
        eng.fit(None, n_iter=10, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=PLSA_FUSED)
        out["fused_pass_avg_ms"] = {kk: round(v[1] / v[0], 4) for kk, v in eng.timing_report().items() if "pass" in kk}
        eng.timing(False)
    if e_step:
        reset()
        eng.timing(True)
        eng.e_step(1e-32, want_host_copy=False)
        eng.timing_reset()
        for _ in range(5):
            eng.e_step(1e-32, want_host_copy=False)
        ms, cnt = eng.timing_get("k_e_step")
        eng.timing(False)
        b = algorithmic_bytes("e_step", n, m, nnz, k)
        out["e_step"] = {"avg_launch_ms": round(ms / cnt, 5), "launches": cnt, "algorithmic_GB": round(b / 1e9, 3),
                         "achieved_GBps": round(b / 1e9 / (ms / cnt / 1e3), 1),
                         "frac": round(b / 1e9 / (ms / cnt / 1e3) / HBM_PEAK_GBS, 4)}
        eng.release_scratch()
    if with_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline(eng, cfg, k, whole=True)
        except Exception as e:
            out["cpu_baseline"] = "failed: %r" % (e,)
    return out


def ensemble_20ng_shape(eng, seed, n_runs=32, n_jobs=4):
    """BASELINE configs[3] on one GPU: EnsembleTopics' member fan-out -- enstop_amd.ensemble_of_topics(X, 20, n_runs=32)
    on the 20NG-shaped corpus, 50 EM iterations per member (BASELINE configs[0]), up to four members in flight
    (n_jobs: the reference's thread pool, enstop_.py:209-217).  Wall clock of the whole call, upload included."""
    import enstop_amd
    cfg = CONFIGS[4]
    eng.release_scratch()
    eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], zipf_s=1.07, seed=seed, **TOPICAL_20NG)
    X = eng.download_active_csr()
    kw = dict(n_iter=FITS_ITERS, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, random_state=seed + 7, n_jobs=n_jobs)
    enstop_amd.ensemble_of_topics(X, cfg["k"], n_runs=n_jobs, **kw)          # warm-up: member contexts and their buffers
    walls = []
    for _ in range(3):
        t0 = time.perf_counter()
        stack = enstop_amd.ensemble_of_topics(X, cfg["k"], n_runs=n_runs, **kw)
        walls.append(time.perf_counter() - t0)
    assert stack.shape == (n_runs * cfg["k"], cfg["m"]) and np.all(np.isfinite(stack))
    assert np.abs(stack.sum(axis=1, dtype=np.float64) - 1.0).max() < 1e-3
    wall = sorted(walls)[1]
    return {"workload": cfg["name"], "corpus": corpus_kind(TOPICAL_20NG), "nnz": int(X.nnz), "k": cfg["k"], "fits": n_runs,
            "iters_per_member": FITS_ITERS,
            "n_jobs": n_jobs, "steps": n_runs * FITS_ITERS, "wall_s": round(wall, 4), "walls_s": [round(w, 4) for w in walls],
            "ms_per_step": round(wall / (n_runs * FITS_ITERS) * 1e3, 5), "ms_per_fit": round(wall / n_runs * 1e3, 3),
            "value": round(n_runs / wall * 60.0, 1), "unit": "fits/min",
            "path": "enstop_amd.ensemble_of_topics(X_host, 20, n_runs=32, n_iter=50, tolerance=0, n_jobs=4): upload + per member "
                    "(device bootstrap gather, CSC / item build, MT19937 init on the device, fit, D2D into the stack) + one copy "
                    "of the stack to the host; median of three calls"}


def ensemble_topics_estimator_20ng_shape(eng, seed, n_starts=32):
    """BASELINE configs[3] through the ESTIMATOR: EnsembleTopics(n_components=20, n_starts=32).fit_transform on a corpus of the
    20NG shape with 20 planted topics (so the outcome can be checked, not only timed): the 32 bootstrapped 50-iteration fits,
    the all-pairs Hellinger matrix of the 640 topics, the HDBSCAN* leaf clusters, their representatives, the refit of every
    document (enstop_.py:417-584).  Wall clock of the whole call; median of three."""
    import enstop_amd
    from scipy.optimize import linear_sum_assignment
    cfg = CONFIGS[4]
    eng.release_scratch()
    eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], seed=seed + 5, topics=cfg["k"], alpha=0.05, background=0.1)
    X = eng.download_active_csr().astype(np.int64)
    labels = eng.synthetic_dominant_topics()
    walls, found, acc = [], None, None
    for _ in range(4):                                       # the first call warms the member contexts up
        et = enstop_amd.EnsembleTopics(n_components=cfg["k"], n_starts=n_starts, topic_combination="hellinger",
                                       n_iter=FITS_ITERS, n_jobs=4, random_state=seed + 1)
        t0 = time.perf_counter()
        emb = et.fit_transform(X)
        walls.append(time.perf_counter() - t0)
        found = int(et.n_components_)
        C = np.zeros((found, cfg["k"]), np.int64)
        np.add.at(C, (emb.argmax(axis=1), labels), 1)
        r_, c_ = linear_sum_assignment(-C)
        acc = float(C[r_, c_].sum()) / len(labels)
    wall = sorted(walls[1:])[1]
    return {"workload": "EnsembleTopics(n_components=20, n_starts=32, topic_combination='hellinger', n_iter=50) on a 20NG-shaped "
                        "corpus with 20 planted topics (%d x %d, %d nnz)" % (X.shape[0], X.shape[1], X.nnz),
            "fits": n_starts, "steps": n_starts * FITS_ITERS, "wall_s": round(wall, 4), "walls_s": [round(w, 4) for w in walls[1:]],
            "ms_per_step": round(wall / (n_starts * FITS_ITERS) * 1e3, 5), "value": round(n_starts / wall * 60.0, 1),
            "unit": "member fits/min, topic combination and refit of all documents included",
            "topics_found": found, "topics_planted": cfg["k"], "documents_on_their_planted_topic": round(acc, 4)}


def ensemble_leg(eng, comm, k, world, rank, args, rccl_init_s):
    """MEASURED ensemble throughput through the product's own call (enstop_amd.ensemble_of_topics): every member
    pays what a member of the reference pays (enstop_.py:84-115) -- bootstrap resample, structure build,
    initialisation, 50 EM iterations -- plus upload of the corpus and the all-gather of the stack."""
    import enstop_amd
    from enstop_amd import enstop_ as product
    eng.bootstrap(None)
    X = eng.download_active_csr()                      # host copy of the corpus: what a caller of the API holds
    members = args.members_per_rank
    if members <= 0:                                   # default: 4 per rank; --config 4: BASELINE's 32 runs over the ranks
        members = max(1, CONFIGS[args.config].get("ensemble_runs", 4 * world) // world)
    n_runs = members * world
    kw = dict(n_runs=n_runs, n_iter=FITS_ITERS, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32,
              random_state=args.seed + 7, n_jobs=4)
    # untimed warm-up of the ensemble path itself (one member per rank, two iterations): the member stack on the device, the
    # page-locked landing buffer of the gather, the page-locked slots of the staged upload -- one-time allocations of a process
    # (~40 ms at config 3) that rounds 4-5 billed to the first two members
    enstop_amd.ensemble_of_topics(X, k, **dict(kw, n_runs=world, n_iter=2))
    eng.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    stack = enstop_amd.ensemble_of_topics(X, k, **kw)
    eng.synchronize()
    comm.barrier()
    dt = time.perf_counter() - t0
    assert stack.shape == (n_runs * k, X.shape[1]) and np.all(np.isfinite(stack))
    sums = stack.sum(axis=1)
    assert np.abs(sums - 1.0).max() < 1e-3, "a member's topics do not sum to one"
    tm = dict(product.last_ensemble_timing)
    mine = np.array([dt, tm.get("fit_s", 0.0), tm.get("gather_s", 0.0), rccl_init_s], np.float64)
    per_rank = comm.allgather_array(mine)              # [world, 4]
    wall = float(per_rank[:, 0].max())
    return {"fits": n_runs, "members_per_rank": members, "iters_per_member": FITS_ITERS,
            "wall_s": round(wall, 4), "fits_per_min": round(n_runs / wall * 60.0, 2),
            "ms_per_member": round(wall / members * 1e3, 1),
            "warm_up": "one untimed member per rank (2 iterations) through the same call: one-time buffers of the process",
            "per_rank": [{"rank": r, "wall_s": round(float(per_rank[r, 0]), 4), "fit_s": round(float(per_rank[r, 1]), 4),
                          "gather_s": round(float(per_rank[r, 2]), 4), "rccl_init_s": round(float(per_rank[r, 3]), 4)}
                         for r in range(world)],
            "path": "enstop_amd.ensemble_of_topics(X_host, k, n_runs=%d, n_iter=%d, tolerance=0): upload + per member "
                    "(device bootstrap gather, CSC / item build, MT19937 init on the device, fit, D2D into the stack) + "
                    "distributed.gather_stack" % (n_runs, FITS_ITERS)}


def spawn_ranks(args):
    """`python bench.py --gpus N` without an external launcher: start the N ranks ourselves (one process
    per GPU), give them a private rendezvous file for the RCCL unique id, relay rank 0's JSON line."""
    tmpdir = tempfile.mkdtemp(prefix="plsa_bench_")
    base = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT="0",
                PLSA_LAUNCH_NONCE=os.path.basename(tmpdir),
                PLSA_COMM_ID_FILE=os.path.join(tmpdir, "rccl.id"), PLSA_BENCH_FILES_DIR=os.path.join(tmpdir, "x"),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(args.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        deadline = time.time() + float(os.environ.get("PLSA_BENCH_SPAWN_TIMEOUT", "3000"))
        pending = list(procs)
        while pending:
            for p_ in list(pending):
                code = p_.poll()
                if code is not None:
                    pending.remove(p_)
                    if code != 0 and rc == 0:
                        rc = code
                        for q in pending:           # one rank died: the others would wait in a collective forever --
                            q.terminate()           # SIGTERM: each prints the stage it was at (comm.install_termination_reporter)
            if time.time() > deadline:
                rc = rc or 124
                for q in pending:
                    q.kill()
                break
            time.sleep(0.05)
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)
    return rc


def corpus_options(args):
    """--topics N > 0: topical corpus with N latent topics; 0: independent tokens; -1 (default): the 20-topic stand-in for the
    20-Newsgroups configurations (1, 4), independent tokens -- SURVEY.md 8d's generator -- for the others."""
    if args.topics > 0:
        return dict(TOPICAL, topics=args.topics)
    if args.topics < 0 and args.config in (1, 4):
        return dict(TOPICAL_20NG)
    return {}


def pmc_child(args):
    """Short run for the counter passes (`rocprofv3 --pmc ... -- python bench.py --pmc-child`): the same
    corpus and factors, two launches of the materialising E-step and four fused EM iterations."""
    from enstop_amd.engine import Engine, PLSA_FUSED
    cfg = CONFIGS[args.config]
    eng = Engine(0)
    eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], zipf_s=1.07, seed=args.seed, **corpus_options(args))
    U0, V0 = init_factors(cfg["n"], cfg["m"], cfg["k"], 42)
    eng.set_factors(U0, V0)
    # four iterations: the first two document passes carry a log-likelihood (k_row_pass<fused,LL>), the others do not
    eng.fit(None, n_iter=4, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=PLSA_FUSED)
    for _ in range(2):
        eng.e_step(1e-32, want_host_copy=False)
    eng.synchronize()
    eng.close()


def short_kernel_name(k):
    """rocprofv3 kernel name -> the engine's timing name (k_e_step_rows / k_e_step -> k_e_step, ...).
    Names look like `void plsa::k_col_pass<plsa::Shape<16, 1, true>, false>(...)`: the flags that matter
    are the template arguments BEHIND the Shape."""
    import re
    head = k.replace("void ", "").split("(")[0]
    if "k_e_step" in head:
        return "k_e_step"
    # flags BEHIND the Shape: k_row_pass<S, FROM_P, WANT_LL[, TINY]>, k_col_pass<S, FROM_P, TIMED[, TINY]>
    mm = re.search(r"(k_row_pass|k_col_pass)<.*>\s*,\s*(true|false)(?:\s*,\s*(true|false))?(?:\s*,\s*(?:true|false))?\s*>\s*$", head)
    if not mm:
        return None
    if mm.group(1) == "k_col_pass":
        return "k_col_pass<P>" if mm.group(2) == "true" else "k_col_pass<fused>"
    if mm.group(2) == "true":
        return "k_row_pass<P>"
    return "k_row_pass<fused,LL>" if mm.group(3) == "true" else "k_row_pass<fused>"


def measure_traffic(args):
    """HBM-side bytes per launch of the hot kernels, measured now: two rocprofv3 passes (FETCH_SIZE and
    WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots') over --pmc-child.
    Returns {kernel: {fetch_size_kib_raw, write_size_kib_raw, hbm_bytes_per_launch}} or None."""
    exe = shutil.which("rocprofv3")
    if exe is None or os.environ.get("PLSA_BENCH_NO_PMC", "0") == "1":
        return None
    raw = {}
    tmpdir = tempfile.mkdtemp(prefix="plsa_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmpdir, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "b", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--config", str(args.config),
                   "--seed", str(args.seed), "--topics", str(args.topics), "--no-pmc"]
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=float(os.environ.get("PLSA_BENCH_PMC_TIMEOUT", "150")), check=True)
            agg = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = short_kernel_name(row.get("Kernel_Name", ""))
                    if name is None or row.get("Counter_Name") != counter:
                        continue
                    n_, v_ = agg.get(name, (0, 0.0))
                    agg[name] = (n_ + 1, v_ + float(row["Counter_Value"]))
            for name, (n_, v_) in agg.items():
                raw.setdefault(name, {})[counter] = v_ / n_
    except Exception as e:                                # never cost the measurement
        print("bench.py: PMC passes failed (%r); traffic falls back to the committed profile" % (e,), file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)
    table = {}
    for name, d in raw.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            # FETCH_SIZE / WRITE_SIZE are reported in KiB; gfx950 tallies a 128-B read request as 64 B
            table[name] = {"fetch_size_kib_raw": d["FETCH_SIZE"], "write_size_kib_raw": d["WRITE_SIZE"],
                           "hbm_bytes_per_launch": int((2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)}
    return table or None


def hot_kernel_table(k, kernels, traffic, source):
    """Per hot kernel of this run: registers / occupancy / LDS as hipcc reported them when the library was built
    (enstop_amd/kernel_resources.json, written by enstop_amd/build.py), launch time, compulsory bytes, counter
    traffic and traffic_ratio = counter bytes / compulsory bytes (well above 1 = gathered rows re-fetched)."""
    res = {}
    try:
        res = json.load(open(os.path.join(ROOT, "enstop_amd", "kernel_resources.json")))
    except Exception:
        pass
    shape = {20: "k=20", 32: "k=32", 64: "k=64", 128: "k=128"}.get(k)
    suffix = {"k_e_step": ", false>", "k_row_pass<fused>": ", false, false, false>", "k_row_pass<fused,LL>": ", false, true, false>",
              "k_col_pass<fused>": ", false, false, false>", "k_row_pass<P>": ", true, false, false>",
              "k_col_pass<P>": ", true, false, false>"}
    out = {}
    for name, e in kernels.items():
        if name not in suffix or "algorithmic_GB" not in e:
            continue
        row = {"avg_ms": e["avg_ms"], "algorithmic_GB": e["algorithmic_GB"], "frac_of_hbm_peak_on_algorithmic_bytes":
               round(e["GBps"] / HBM_PEAK_GBS, 4)}
        t = traffic.get(name)
        if t:
            row["traffic_GB"] = round(t["hbm_bytes_per_launch"] / 1e9, 3)
            row["traffic_ratio"] = round(t["hbm_bytes_per_launch"] / 1e9 / e["algorithmic_GB"], 2)
            row["frac_of_hbm_peak_on_traffic"] = round(t["hbm_bytes_per_launch"] / 1e9 / (e["avg_ms"] / 1e3) / HBM_PEAK_GBS, 4)
            row["traffic_source"] = "this run" if source != "from_committed_profile" else "committed profile"
        base = name.split("<")[0]
        for r in res.get("kernels", []):
            if r["shape"] != shape or not r["kernel"].startswith(base):
                continue
            if not r["kernel"].endswith(suffix[name]):
                continue
            row.setdefault("instantiations", []).append(
                {"kernel": r["kernel"], "vgprs": r["vgprs"], "waves_per_simd": r["waves_per_simd"],
                 "lds_bytes": r["lds_bytes"], "scratch_bytes_per_lane": r["scratch"]})
        if base == "k_row_pass" and any("Shape<8, 2" in i["kernel"] for i in row.get("instantiations", [])):
            # k = 64: the document pass is launched in the 8 x 2 lane shape (plsa_hip.hip::set_shape)
            row["instantiations"] = [i for i in row["instantiations"] if "Shape<8, 2" in i["kernel"]]
        out[name] = row
    return out

def _pick(d, *keys):
    return {k_: d[k_] for k_ in keys if isinstance(d, dict) and k_ in d}


def reference_arithmetic_leg(eng, seed):
    """The opt-in PARITY mode (PLSA_REFERENCE_SUMS: every sum of the E- and M-step the reference's float32 chain, DESIGN.md
    section 2.1) at the two BASELINE sizes the reference itself is run at: EM iterations/s, the likelihood every ten iterations
    like the reference's default.  Not `value`: the bits it returns are the subject of tests/test_reference_arithmetic.py."""
    from enstop_amd.engine import PLSA_REFERENCE_SUMS
    out = {"unit": "iter/s", "arithmetic": "reference (float32 sums in the reference's order)"}
    for cfg_id, kw in ((1, dict(TOPICAL_20NG)), (2, {})):
        cfg = CONFIGS[cfg_id]
        eng.release_scratch()
        eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], zipf_s=1.07, seed=seed, **kw)
        eng.init_factors_numpy_stream(cfg["k"], np.random.RandomState(42))
        fit = dict(n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=PLSA_REFERENCE_SUMS)
        eng.fit(None, n_iter=10, **fit)
        eng.synchronize()
        t0 = time.perf_counter()
        it, _ = eng.fit(None, n_iter=30, **fit)
        eng.synchronize()
        dt = time.perf_counter() - t0
        key = "value" if cfg_id == 1 else "config2_value"
        out[key] = round(it / dt, 1)
        out["ms_per_step" if cfg_id == 1 else "config2_ms_per_step"] = round(dt / it * 1e3, 3)
    out["chain"] = eng.reference_chain_info()
    return out


def compact_line(out, full_path=None):
    """The ONE JSON line of a default run: the contract keys, `roofline` (north-star kernel + `timed_loop` = the dominant
    kernel of the loop `value` times), `cpu_baseline`, every other BASELINE configuration and the ensemble legs as one
    short object each -- sized to fit the 2000-character tail the driver keeps.  Everything else (per-kernel tables,
    register counts, counter traffic, schedules) is in the full object (`full`, --full-json / --full)."""
    line = _pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data")
    c = out.get("config", {})
    line["config"] = {"workload": CONFIG_SHORT.get((c.get("n_docs"), c.get("k")), c.get("workload")),
                      "corpus": "topical" if str(c.get("corpus", "")).startswith("topical") else "independent Zipf tokens (SURVEY 8d)",
                      "nnz": c.get("nnz"), "k": c.get("k"), "schedule": c.get("schedule"),
                      "parallelism": str(c.get("parallelism", ""))[:60]}
    r = out.get("roofline") or {}
    line["roofline"] = _pick(r, "kernel", "leg", "bound", "achieved", "peak", "unit", "frac", "traffic")
    line["roofline"]["avg_ms"] = r.get("avg_launch_ms")
    if "timed_loop" in r:
        line["roofline"]["timed_loop"] = r["timed_loop"]
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, "value", "unit", "cores", "kind", "sampled")
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:90]
    line["timed_regions"] = [t.get("value") for t in out.get("timed_regions", [])]
    line["rccl_ranks"] = out.get("rccl_ranks", 0)
    line["exchange"] = str(out.get("exchange", ""))[:60]
    oc = out.get("other_configs")
    if isinstance(oc, dict):
        short = line["other_configs"] = {}
        for name, v in oc.items():
            if not isinstance(v, dict):
                short[name] = str(v)[:60]
                continue
            e = _pick(v, "value", "steps", "ms_per_step")
            if name == "reference_arithmetic":
                e = _pick(v, "value", "config2_value")
            if "iter/s" not in str(v.get("unit", "iter/s")):
                e["unit"] = str(v.get("unit"))[:15]
            if isinstance(v.get("e_step"), dict):
                e["e_step_frac"] = v["e_step"].get("frac")
            if isinstance(v.get("cpu_baseline"), dict):
                e["cpu_port_iter_s"] = v["cpu_baseline"].get("value")
            if str(v.get("corpus", "")).startswith("topical"):
                e["corpus"] = "topical"
            for extra in ("wall_s", "topics_found", "documents_on_their_planted_topic", "config2_value"):
                if extra in v:
                    e[extra] = v[extra]
            short[name] = e
    en = out.get("ensemble")
    if isinstance(en, dict):
        line["ensemble"] = _pick(en, "fits", "iters_per_member", "wall_s", "fits_per_min", "ms_per_member")
    elif en is not None:
        line["ensemble"] = str(en)[:80]
    if full_path:
        line["full"] = full_path
    return line


CONFIG_SHORT = {(18_846, 20): "20NG-shaped 18846x173762, 2.95M nnz", (100_000, 32): "100k x 50k, 10M nnz",
                (1_000_000, 64): "1M x 100k, 100M nnz", (5_000_000, 128): "5M x 200k, 500M nnz"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--schedule", default=os.environ.get("PLSA_BENCH_SCHEDULE", "fused"),
                    choices=["fused", "materialised"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="(default since round 5; kept for old command lines)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 counter passes")
    ap.add_argument("--no-ensemble", action="store_true", help="skip the measured ensemble leg")
    ap.add_argument("--members-per-rank", type=int, default=0,
                    help="members each rank fits in the measured ensemble leg (0 = 4, or 32 / N with --config 4)")
    ap.add_argument("--topics", type=int, default=-1,
                    help="generate a TOPICAL corpus of the config's shape (documents as Dirichlet mixtures of this many latent "
                         "topics; plsa_generate_synthetic_topics) instead of independent Zipf tokens; labelled in `config`.  "
                         "Default -1: 20 topics for the 20-Newsgroups configurations (1, 4), independent tokens (0) otherwise")
    ap.add_argument("--full", action="store_true",
                    help="print the FULL result object as the JSON line (tools); default: the compact line (the driver keeps a "
                         "2000-character tail) and the full object in --full-json")
    ap.add_argument("--full-json", default="", help="where the full result object goes (default gpurun_out/bench_full_cfgC_nN.json)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the compact legs of the other BASELINE configs")
    ap.add_argument("--cpu-baseline-sampled", action="store_true",
                    help="time the CPU port on a bounded row sample (~15 s) instead of the whole corpus (config 3: ~45 s, 26 GB)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "files"],
                    help="files: TEST MODE for boxes with fewer GPUs than ranks (host-file exchange, shared GPUs)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    # Only the JSON line may reach stdout: route file descriptor 1 to stderr for the whole run (RCCL and
    # the HIP runtime print banners from C) and write the result to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    from enstop_amd import _lib, comm as plsa_comm
    from enstop_amd.engine import Engine, PLSA_FUSED
    import ctypes
    if world > 1:
        plsa_comm.install_termination_reporter(rank, world)
    cnt = ctypes.c_int(0)
    if _lib.load().plsa_device_count(ctypes.byref(cnt)) or cnt.value < 1:
        print("bench.py: no HIP device visible", file=sys.stderr)
        sys.exit(3)
    n_dev = cnt.value
    if world > 1 and args.exchange == "rccl" and n_dev < world and n_dev != 1:
        print("bench.py: %d ranks but only %d GPUs visible" % (world, n_dev), file=sys.stderr)
        sys.exit(3)
    if world > 1 and args.exchange == "rccl" and n_dev == 1 and local_rank > 0 and "HIP_VISIBLE_DEVICES" not in os.environ \
            and "ROCR_VISIBLE_DEVICES" not in os.environ:
        print("bench.py: %d ranks need %d GPUs, 1 visible (use --exchange files to test on a shared GPU)"
              % (world, world), file=sys.stderr)
        sys.exit(3)
    device = local_rank if local_rank < n_dev else local_rank % n_dev
    plsa_comm._STATE["device"] = device
    fail_at = os.environ.get("PLSA_BENCH_FAIL_AT", "")          # test hook "rank:stage": that rank dies there (exit 9)

    def stage(name):
        plsa_comm.set_stage(name)
        if fail_at == "%d:%s" % (rank, name):
            print("bench.py: rank %d exits at stage '%s' (PLSA_BENCH_FAIL_AT)" % (rank, name), file=sys.stderr)
            os._exit(9)

    cfg = CONFIGS[args.config]
    n, m, k = cfg["n"], cfg["m"], cfg["k"]
    flags = (PLSA_FUSED if args.schedule == "fused" else 0)

    os.environ.setdefault("ENSTOP_AMD_DEVICE", str(device))
    from enstop_amd.engine import get_engine
    eng = get_engine(device)          # the process-wide engine of this GPU: the one the product's own calls use
    info = eng.device_info()
    comm = plsa_comm.SingleComm()
    rccl_init_s = 0.0
    if world > 1:
        if args.exchange == "rccl":
            # any failure here (duplicate GPU, bootstrap, ...) raises: non-zero exit status, no JSON line
            t_init = time.perf_counter()
            comm = plsa_comm.init_from_env(eng)
            rccl_init_s = time.perf_counter() - t_init
        else:
            comm = plsa_comm.install(plsa_comm.FileComm(os.environ.get("PLSA_BENCH_FILES_DIR", "/tmp/plsa_bench_x"),
                                                        rank, world))
    t_gen = time.perf_counter()
    nnz = eng.generate_synthetic(n, m, cfg["nnz"], zipf_s=1.07, seed=args.seed, **corpus_options(args))
    if world > 1:  # ensemble member `rank`: bootstrap rows on the device (enstop_.py:87-88)
        idx = np.random.RandomState(args.seed + 1000 + rank).randint(0, n, size=n)
        eng.bootstrap(idx)
    n_act, m_act, nnz_act = eng.shape
    t_gen = time.perf_counter() - t_gen
    U0, V0 = init_factors(n_act, m, k, 42 + rank)
    eng.set_factors(U0, V0)

    def barrier():
        eng.synchronize()
        comm.barrier()

    from enstop_amd import distributed as plsa_dist
    stack_base = eng.stack_reserve(1, k, m) if world > 1 else None
    # (the gathered stack is read through a view of the engine's page-locked buffer: distributed.gather_stack(view=True),
    #  the product's fastest host copy -- no page faults of a fresh result array inside the timed region)

    def gather_components():
        """the np.vstack of enstop_.py:231 through the product's own exchange: this member's topics go into the
        engine's device stack, enstop_amd.distributed.gather_stack brings every rank's stack together"""
        if world == 1:
            return None
        eng.copy_components_to_device(stack_base)
        return plsa_dist.gather_stack(eng, world, k, m, view=True).reshape(world, k, m)     # view of the page-locked buffer

    # ---- warmup (untimed): W EM iterations + the collective --------------------------------------
    stage("warm-up gather")
    if args.warmup > 0:
        it, _ = eng.fit(None, n_iter=args.warmup, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=flags)
        assert it == args.warmup
    gather_components()

    # ---- timed regions: three times exactly K EM iterations, each bracketed by barrier + synchronize; the
    # MEDIAN region is reported (its per-kernel events, its wall time), all three are listed
    stage("timed")
    regions = []
    stack = None
    for _region in range(3):
        eng.set_factors(U0, V0)
        eng.timing(True)
        eng.timing_reset()
        barrier()
        t0 = time.perf_counter()
        it, _ = eng.fit(None, n_iter=args.steps, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=flags)
        t_fit = time.perf_counter() - t0
        stack = gather_components()
        t_gather = time.perf_counter() - t0 - t_fit
        barrier()
        dt_r = time.perf_counter() - t0
        assert it == args.steps, "early stop inside the timed region (%d of %d)" % (it, args.steps)
        rep_r = eng.timing_report()
        eng.timing(False)
        dt_r = float(comm.allreduce_f64([dt_r], "max")[0])
        split = comm.allreduce_f64([t_fit, t_gather], "max")
        regions.append((dt_r, rep_r, float(split[0]), float(split[1])))
    # one more region of the same K iterations WITHOUT the per-kernel HIP events (two event records per launch cost
    # ~1.5 % at config 3 and ~25 % at the 20NG shape, whose iteration is seven launches of 6-60 us): reported next to
    # `value`, never instead of it
    eng.set_factors(U0, V0)
    barrier()
    t0 = time.perf_counter()
    it, _ = eng.fit(None, n_iter=args.steps, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=flags)
    gather_components()
    barrier()
    dt_plain = float(comm.allreduce_f64([time.perf_counter() - t0], "max")[0])
    assert it == args.steps
    order = sorted(range(3), key=lambda i: regions[i][0])
    dt, report, dt_fit, dt_gather = regions[order[1]]
    schedule = eng.balance_info()            # measured XCD boundaries of the column pass (results do not depend on them)
    if stack is not None and rank == 0:
        assert stack.shape == (world, k, m) and np.all(np.isfinite(stack))
        _, V_mine = eng.get_factors(want_u=False)
        assert np.array_equal(stack[0], V_mine), "all-gather slot 0 is not rank 0's topic matrix"

    nnz_total = float(comm.allreduce_f64([float(nnz_act)], "sum")[0])

    # ---- per-kernel roofline figures from the HIP events of the timed region ----------------------
    kernels = {}
    for name, (cnt_, ms) in report.items():
        kind = KERNEL_KIND.get(name)
        entry = {"launches": cnt_, "avg_ms": round(ms / cnt_, 5), "total_ms": round(ms, 4)}
        if kind:
            b = algorithmic_bytes(kind, n_act, m, nnz_act, k)
            entry["algorithmic_GB"] = round(b / 1e9, 4)
            entry["GBps"] = round(b / 1e9 / (ms / cnt_ / 1e3), 1)
        kernels[name] = entry
    dom = max((kv for kv in kernels.items() if "GBps" in kv[1]), key=lambda kv: kv[1]["total_ms"])

    def roof(name, e):
        return {"kernel": name, "bound": "hbm", "achieved": e["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(e["GBps"] / HBM_PEAK_GBS, 4), "traffic": None,
                "algorithmic_GB_per_launch": e["algorithmic_GB"], "avg_launch_ms": e["avg_ms"],
                "launches": e["launches"]}

    # ---- second timed leg: the reference's own kernel sequence (E-step -> M-step -> LL test) through
    # the materialised nnz x k array.  It carries the kernel the north_star roofline target is
    # stated on (k_e_step) and is reported next to the fused schedule, never instead of it.
    mat = None
    e_entry = kernels.get("k_e_step")
    if e_entry is None:
        k_mat = min(args.steps, 10)
        eng.set_factors(U0, V0)
        eng.fit(None, n_iter=2, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=0)   # warm-up
        eng.timing(True)
        eng.timing_reset()
        barrier()
        t1 = time.perf_counter()
        it2, _ = eng.fit(None, n_iter=k_mat, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=0)
        barrier()
        dt2 = time.perf_counter() - t1
        rep2 = eng.timing_report()
        eng.timing(False)
        kernels_mat = {}
        for name, (cnt_, ms) in rep2.items():
            kind = KERNEL_KIND.get(name)
            entry = {"launches": cnt_, "avg_ms": round(ms / cnt_, 5), "total_ms": round(ms, 4)}
            if kind:
                bts = algorithmic_bytes(kind, n_act, m, nnz_act, k)
                entry["algorithmic_GB"] = round(bts / 1e9, 4)
                entry["GBps"] = round(bts / 1e9 / (ms / cnt_ / 1e3), 1)
            kernels_mat[name] = entry
        e_entry = kernels_mat["k_e_step"]
        mat = {"schedule": "materialised (reference kernel sequence)", "steps": it2, "p_placement": eng.placement_info(),
               "value": round(it2 / dt2, 4), "ms_per_step": round(dt2 / it2 * 1e3, 4), "kernels": kernels_mat}

    n_gpus = world
    exchange = "none" if world == 1 else ("RCCL (C ABI, ncclAllGather)" if isinstance(comm, plsa_comm.RcclComm)
                                          else "host files -- TEST MODE, ranks share %d GPU(s)" % n_dev)
    out = {
        "metric": "EM iterations/sec", "value": round(n_gpus * args.steps / dt, 4), "unit": "iter/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "value_is": "median of 3 back-to-back timed regions of exactly %d iterations (barrier + synchronize around each, "
                    "max over ranks)" % args.steps,
        "region_without_per_kernel_events": {"value": round(n_gpus * args.steps / dt_plain, 4),
                                             "ms_per_step": round(dt_plain / args.steps * 1e3, 4),
                                             "note": "a fourth region of the same K iterations with the engine's per-kernel "
                                                     "HIP events off; `value` and the per-kernel figures come from the "
                                                     "three regions that record them"},
        "timed_region_split_s": {"iterations": round(dt_fit, 5), "topic_gather_to_host": round(dt_gather, 5),
                                 "note": "max over ranks of the two parts of the median region; the gather (D2D into the stack, "
                                         "all-gather when N > 1, one copy to the host array) is INSIDE the timed region"},
        "timed_regions": [{"value": round(n_gpus * args.steps / r[0], 4), "ms_per_step": round(r[0] / args.steps * 1e3, 4)}
                          for r in regions],
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["name"] + ("" if not corpus_options(args) else " -- TOPICAL corpus %r" % (corpus_options(args),)),
                   "corpus": corpus_kind(corpus_options(args)),
                   "n_docs": n, "n_vocab": m, "nnz": nnz, "k": k,
                   "schedule": args.schedule,
                   "parallelism": "single fit" if n_gpus == 1 else
                   "ensemble: one bootstrap member per GPU x%d, all-gather of topics: %s" % (n_gpus, exchange),
                   "ll_test_every": 10, "tolerance": 0.0, "e_step_thresh": 1e-32},
        "rccl_ranks": world if isinstance(comm, plsa_comm.RcclComm) else 0,
        "exchange": exchange,
        "gcell_per_s": round(nnz_total * k * args.steps / dt / 1e9, 3),
        # derived from the iteration rate alone (no bootstrap / structure build / init / gather): the MEASURED
        # figure is ensemble.fits_per_min below
        "ensemble_fits_per_min_from_iteration_rate": round(n_gpus * args.steps / dt / FITS_ITERS * 60.0, 3),
        # north_star's roofline kernel: the per-nnz materialising E-step (HBM-bound)
        "roofline_note": "`roofline`: k_e_step, MATERIALISED schedule (second leg, `materialised_leg`; `value` never launches "
                         "it). `roofline_dominant_fused`: dominant kernel of the FUSED schedule that produces `value`, against "
                         "its compulsory bytes; its counter traffic / ratio: `hot_kernels`.",
        "roofline": roof("k_e_step", e_entry),
        # dominant kernel of the (fused) timed region against its own compulsory bytes; it is
        # gather/VALU-bound, not an HBM-roofline claim
        "roofline_dominant_fused": roof(*dom),
        "kernels": kernels,
        "column_pass_schedule": schedule,
        "materialised_leg": mat,
        "device": info["name"] or "AMD Instinct MI355X", "arch": info["arch"], "generate_s": round(t_gen, 2),
    }
    if rank == 0:
        if n_gpus == 1 and not args.no_cpu_baseline:
            # the WHOLE corpus by default (config 3: 2 iterations, ~45 s of host time, 26 GB of host memory: nothing is
            # extrapolated); --cpu-baseline-sampled: the bounded row sample of rounds 1-4 (~15 s, "sampled": true)
            try:
                out["cpu_baseline"] = cpu_baseline(eng, cfg, k) if args.cpu_baseline_sampled else \
                    cpu_baseline(eng, cfg, k, iters=2, whole=True)
            except Exception as e:       # the baseline must never cost the GPU measurement
                out["cpu_baseline"] = {"value": None, "unit": "iter/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
    # ---- measured ensemble: the product's own call, four members per rank (see the module docstring) -------------
    if not args.no_ensemble:
        stage("ensemble")
        try:
            out["ensemble"] = ensemble_leg(eng, comm, k, world, rank, args, rccl_init_s)
            out["ensemble_fits_per_min"] = out["ensemble"]["fits_per_min"]
        except Exception as e:
            if world > 1:
                plsa_comm.report_failure(e, rank, world, eng)
                raise                                     # a rank that fails here would leave the others in a collective
            out["ensemble"] = "failed: %r" % (e,)
    if rank == 0 and n_gpus == 1 and args.config == 3 and not corpus_options(args) and not args.no_other_configs:
        # every other BASELINE.json configuration on the same device in compact form (same method as the main legs), so
        # that each is a number the driver's own run observes: configs[0] / [1] / [4] as EM iterations/s + the
        # materialising E-step's roofline fraction, configs[3] (the 32-member ensemble on the 20NG shape) as fits/min
        # through the product's own call; config 2 with the CPU port on the WHOLE of its corpus; and config 3's shape with
        # TOPICAL structure (the corpus above has no co-occurrence: DESIGN.md section 5)
        other = out["other_configs"] = {}

        def leg(name, fn):
            t_leg = time.perf_counter()
            try:
                other[name] = fn()
                if isinstance(other[name], dict):
                    other[name]["leg_wall_s"] = round(time.perf_counter() - t_leg, 1)
            except Exception as e:
                other[name] = "failed: %r" % (e,)
        leg("config2", lambda: quick_config(eng, 2, args.steps, args.warmup, args.seed, with_cpu=not args.no_cpu_baseline))
        c2 = other["config2"]
        if isinstance(out.get("cpu_baseline"), dict) and isinstance(c2, dict) and isinstance(c2.get("cpu_baseline"), dict):
            out["cpu_baseline"]["whole_config2"] = c2["cpu_baseline"]
        leg("config1", lambda: quick_config(eng, 1, args.steps, args.warmup, args.seed, min_steps=1000, corpus_kw=dict(TOPICAL_20NG)))
        leg("reference_arithmetic", lambda: reference_arithmetic_leg(eng, args.seed))
        leg("ensemble_20ng_shape", lambda: ensemble_20ng_shape(eng, args.seed))
        leg("ensemble_topics_estimator_20ng_shape", lambda: ensemble_topics_estimator_20ng_shape(eng, args.seed))
        leg("config3_topical", lambda: quick_config(eng, 3, args.steps, args.warmup, args.seed, e_step=False, min_steps=50,
                                                    corpus_kw=dict(TOPICAL), pass_times=True))
        from enstop_amd.engine import reset_engines as _reset
        _reset()                                   # config 5's materialised leg needs ~270 of the 288 GB
        eng = get_engine(device)
        leg("config5", lambda: quick_config(eng, 5, args.steps, args.warmup, args.seed, min_steps=20, device_init=True))
    if world > 1:
        comm.barrier()
        plsa_comm.shutdown()
    from enstop_amd.engine import reset_engines
    reset_engines()
    # ---- HBM-side traffic of the roofline kernels: PMC passes in this run, else the committed profile ----
    if rank == 0:
        table = None
        if n_gpus == 1 and not args.no_pmc:
            table = measure_traffic(args)
        source = "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over " \
                 "`bench.py --pmc-child`; (2 x FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled per MI355X_MICROARCH.md"
        if table is None:
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    table = json.load(open(pmc)).get("config%d" % args.config, {})
                    source = "from_committed_profile"
                except Exception:
                    table = None
        for key in ("roofline", "roofline_dominant_fused"):
            rec = (table or {}).get(out[key]["kernel"])
            if rec:
                out[key]["traffic"] = rec["hbm_bytes_per_launch"]
                out[key]["traffic_source"] = source if source != "from_committed_profile" else \
                    "from_committed_profile: " + rec.get("source", "profiles/pmc_traffic.json")
        # the driver-visible roofline object names BOTH kernels: the north-star one (k_e_step: not launched by the loop `value`
        # times -- a separate, materialised leg) and, as `timed_loop`, the dominant kernel of the loop that IS timed
        out["roofline"]["leg"] = "materialised (separate from value)" if mat is not None else "the timed loop itself (--schedule materialised)"
        d_ = out["roofline_dominant_fused"]
        tl = {"kernel": d_["kernel"], "avg_ms": d_["avg_launch_ms"], "algorithmic_GB": d_["algorithmic_GB_per_launch"],
              "frac": d_["frac"], "traffic_GB": None, "traffic_ratio": None, "frac_on_traffic": None,
              "share_of_ms_per_step": round(d_["avg_launch_ms"] / out["ms_per_step"], 3)}
        if d_.get("traffic"):
            tl["traffic_GB"] = round(d_["traffic"] / 1e9, 3)
            tl["traffic_ratio"] = round(d_["traffic"] / 1e9 / d_["algorithmic_GB_per_launch"], 2)
            tl["frac_on_traffic"] = round(d_["traffic"] / 1e9 / (d_["avg_launch_ms"] / 1e3) / HBM_PEAK_GBS, 4)
        out["roofline"]["timed_loop"] = tl
        if table and source != "from_committed_profile":
            out["pmc_traffic"] = table
        out["hot_kernels"] = hot_kernel_table(k, dict(kernels, **((mat or {}).get("kernels", {}))), table or {}, source)
        sys.stdout.flush()
        full_path = args.full_json or os.path.join(ROOT, "gpurun_out", "bench_full_cfg%d_n%d.json" % (args.config, n_gpus))
        try:
            os.makedirs(os.path.dirname(full_path) or ".", exist_ok=True)
            with open(full_path, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:
            print("bench.py: full result object not written (%r)" % (e,), file=sys.stderr)
            full_path = None
        line = out if args.full else compact_line(out, full_path and os.path.relpath(full_path, ROOT))
        os.write(json_fd, (json.dumps(line, separators=(",", ":")) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as exc:        # one line per rank that says where a multi-GPU run died; then the traceback
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            try:
                from enstop_amd import comm as _c
                _c.report_failure(exc)
            except Exception:
                pass
        raise
