"""Ensemble of bootstrapped pLSA fits on MI355X behind the reference's interface
(enstop/enstop_.py:56-115 plsa_topics, :164-231 ensemble_of_topics).

The reference fans the members out over dask/joblib *threads* of one process.  Here a member is one
fit on one GPU: within a process the members assigned to it run back to back on its device (the
corpus is uploaded once, each bootstrap resample is a device-side row gather); across GPUs the
launcher starts one process per device (torchrun); every member's topics go into the process' device stack and the
stacks are all-gathered over RCCL (`distributed.gather_stack`).  Members are dealt round-robin, run r -> rank r mod world_size, and
each member draws from its own RandomState stream so the stack does not depend on the device count.
"""
import numpy as np
from scipy.sparse import issparse, csr_matrix
from sklearn.utils import check_random_state

import os
import threading

from .engine import get_engine, get_member_engines
from .plsa import _fit_on_engine, _locked

# A corpus this small leaves most of the GPU idle during one fit (0.15 ms per EM iteration at the 20NG
# shape, a handful of short dependent kernels): members are then fitted concurrently, like the reference's
# thread pool of nogil fits (enstop_.py:209-217).  Measured on MI355X, 32 members x 50 iterations on the
# 20NG-shaped corpus: 4960 fits/min with one member at a time, 7050 with two, 7470 with three, flat beyond
# (profiles/r02_ensemble_concurrent_members_cfg1.jsonl); config 2 (10 M nnz) +22 %; nothing to gain once a
# single fit saturates the memory system.  Round 4: the member contexts need their own hardware queues (libplsa_hip.so asks
# for 8 at load time, see include/plsa_hip.h): 7 370-7 520 -> 8 070-8 290 fits/min with four members in flight; six or eight
# in flight are slower (profiles/r04_hw_queues_api_jobs_6_8.txt), hence the cap (ENSTOP_AMD_CONCURRENT_MEMBERS_MAX: experiments).
CONCURRENT_MEMBERS_MAX = int(os.environ.get("ENSTOP_AMD_CONCURRENT_MEMBERS_MAX", "4"))
CONCURRENT_MEMBERS_CELLS = float(os.environ.get("ENSTOP_AMD_CONCURRENT_MEMBERS_CELLS", "2e9"))   # nnz * k below which members run concurrently


def _member_flags(kwargs):
    """`flags=` and / or `arithmetic=` of plsa_topics / ensemble_of_topics -> the flags of the members' fits (None: defaults)"""
    from .engine import arithmetic_flags, default_flags
    flags, arithmetic = kwargs.get("flags", None), kwargs.get("arithmetic", None)
    if arithmetic is None:
        return flags
    return (default_flags() if flags is None else flags) | arithmetic_flags(arithmetic)


def concurrent_members(nnz, k, n_jobs):
    if n_jobs is None or n_jobs < 1 or nnz * float(k) >= CONCURRENT_MEMBERS_CELLS:
        return 1
    return int(min(n_jobs, CONCURRENT_MEMBERS_MAX))


last_ensemble_timing = {}      # seconds spent by the last multi-member call of this process (bench.py reports it)


def _fit_members(A, k, runs, seeds, member_kw, device, n_jobs, world=1, n_runs=None):
    """Fits the given runs of an ensemble on this process' GPU and leaves their topic matrices in the device
    stack of the process-wide engine (run r -> slot r // world); run r draws from RandomState(seeds[r]) whichever
    engine or thread fits it, so the result does not depend on `n_jobs`.  Returns the engine that holds the stack."""
    jobs = min(concurrent_members(A.nnz, k, n_jobs), max(len(runs), 1))
    engines = get_member_engines(device, jobs)
    for e in engines[1:]:
        e.upload_csr(A)                  # (the first one holds the corpus already; uploading inside the worker threads
                                         #  instead measured no better: 9 130 against 8 950 fits/min at the 20NG shape)
    m = A.shape[1]
    # every rank reserves the same number of slots (the all-gather is slot by slot), filled or not
    slots = max(1, max((r // world for r in runs), default=0) + 1) if n_runs is None else max(1, (n_runs + world - 1) // world)
    base = engines[0].stack_reserve(slots, k, m)
    errors = []

    def work(j):
        try:
            for r in runs[j::jobs]:
                _member_on_engine(engines[j], k, random_state=np.random.RandomState(seeds[r]),
                                  stack_dst=base + 4 * (r // world) * k * m, **member_kw)
        except BaseException as e:       # re-raised in the caller's thread
            errors.append(e)
    if jobs == 1:
        work(0)
    else:
        threads = [threading.Thread(target=work, args=(j,)) for j in range(jobs)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    return engines[0]


def _member_on_engine(eng, k, bootstrap=True, random_state=None, init="random", n_iter=100,
                      n_iter_per_test=10, tolerance=0.001, e_step_thresh=1e-16, flags=None, stack_dst=None):
    """One ensemble member on a corpus already uploaded to `eng`; returns P(w|z) [k, m] -- or, with
    `stack_dst` (a device address), leaves it there and returns None."""
    n_base = eng.base_rows
    if bootstrap:
        rng = check_random_state(random_state)                       # enstop_.py:86
        idx = rng.randint(0, n_base, size=n_base)                    # enstop_.py:87
        eng.bootstrap(idx)                                           # enstop_.py:88 (on device)
    else:
        eng.bootstrap(None)
    # sample_weight is all ones (enstop_.py:91); random_state is passed on unchanged, so a
    # RandomState instance continues its stream into the factor initialisation while an int
    # re-seeds it (enstop_.py:101,113 + plsa.py:707)
    _fit_on_engine(eng, k, None, init, n_iter, n_iter_per_test, tolerance, e_step_thresh,
                   random_state, flags)
    if stack_dst is not None:
        eng.copy_components_to_device(stack_dst)
        return None
    _, V = eng.get_factors(want_u=False)
    return V


@_locked
def plsa_topics(X, k, **kwargs):
    """Bootstrap-resample the documents of X and fit pLSA; returns the (k, n_words) topic matrix.
    Keyword arguments as the reference: bootstrap, random_state, init, n_iter, n_iter_per_test,
    tolerance, e_step_thresh (default 1e-16 here, enstop_.py:99,111); plus device, flags, arithmetic ("reference": the
    reference's float32 sums, rounding for rounding -- engine.arithmetic_flags)."""
    A = X.tocsr() if issparse(X) else csr_matrix(X)
    eng = get_engine(kwargs.get("device", None))
    eng.upload_csr(A)
    return _member_on_engine(
        eng, k, bootstrap=kwargs.get("bootstrap", True), random_state=kwargs.get("random_state", None),
        init=kwargs.get("init", "random"), n_iter=kwargs.get("n_iter", 100),
        n_iter_per_test=kwargs.get("n_iter_per_test", 10), tolerance=kwargs.get("tolerance", 0.001),
        e_step_thresh=kwargs.get("e_step_thresh", 1e-16), flags=_member_flags(kwargs))


def nmf_topics(X, k, **kwargs):
    """enstop_.py:118-161: bootstrap-resample the documents, fit scikit-learn's NMF on the HOST, return
    the L1-row-normalised components.  NMF is a different model from pLSA and not part of the GPU hot
    path (SURVEY.md section 2 lists it out of scope); it is delegated to scikit-learn exactly as the
    reference does so that `model="nmf"` keeps working.  `alpha` maps to `alpha_W` (scikit-learn
    renamed the parameter in 1.0)."""
    from sklearn.decomposition import NMF
    from .utils import normalize
    A = X.tocsr() if issparse(X) else csr_matrix(X)
    if kwargs.get("bootstrap", True):
        rng = check_random_state(kwargs.get("random_state", None))
        A = A[rng.randint(0, A.shape[0], size=A.shape[0])]
    nmf = NMF(n_components=k, init=kwargs.get("init", "nndsvd"), beta_loss=kwargs.get("beta_loss", 1),
              alpha_W=kwargs.get("alpha", 0.0), solver=kwargs.get("solver", "mu"),
              random_state=kwargs.get("random_state", None)).fit(A)
    topics = np.array(nmf.components_, dtype=np.float64, order="C")
    normalize(topics, axis=1)
    return topics


def _ensemble_of_nmf_topics(X, k, n_runs, **kwargs):
    """model="nmf" branch of ensemble_of_topics (enstop_.py:199-231), serial on the host."""
    kw = {key: kwargs[key] for key in ("bootstrap", "random_state", "init", "beta_loss", "alpha", "solver")
          if key in kwargs}
    return np.vstack([nmf_topics(X, k, **kw) for _ in range(n_runs)])


def ensemble_of_topics(X, k, model="plsa", n_jobs=4, n_runs=16, parallelism="dask", **kwargs):
    """All topics of `n_runs` bootstrapped fits stacked to (n_runs * k, n_words), enstop_.py:164-231.
    model="plsa": on the GPU(s), see `_ensemble_of_plsa_topics`; model="nmf": scikit-learn on the host."""
    if model == "nmf":
        return _ensemble_of_nmf_topics(X, k, n_runs, **kwargs)
    if model != "plsa":
        raise ValueError('Model must be one of "plsa" or "nmf"')
    return _ensemble_of_plsa_topics(X, k, n_jobs, n_runs, parallelism, **kwargs)


@_locked
def _ensemble_of_plsa_topics(X, k, n_jobs=4, n_runs=16, parallelism="dask", **kwargs):
    """All topics of `n_runs` bootstrapped fits stacked to (n_runs * k, n_words), enstop_.py:164-231.

    `parallelism`:
      "none"            members run serially on this process' GPU sharing `random_state` exactly
                        like the reference's serial branch (enstop_.py:220-223).
      "dask" / "joblib" accepted for drop-in compatibility; the thread fan-out they name is
                        replaced by the one-GPU-per-process model: after
                        `enstop_amd.distributed.init()` under a launcher that sets RANK / WORLD_SIZE
                        (torchrun, `bench.py --gpus N`) run r executes on rank r % world_size and
                        the stack is all-gathered over RCCL.  `n_jobs`: members fitted CONCURRENTLY on this
                        process' GPU (at most 4 contexts, and 1 once a single fit fills the GPU:
                        nnz * k >= 2e9); the stack does not depend on it.
    Per-run streams: with an int (or None) `random_state` run r uses seed `random_state + r`
    (reference: every thread re-seeds with the same int, producing identical members,
    enstop_.py:86 -- a documented defect, not reproduced).
    """
    if parallelism not in ("dask", "joblib", "none"):
        raise ValueError("Unrecognized parallelism {}; should be one of {}".format(
            parallelism, ("dask", "joblib", "none")))
    A = X.tocsr() if issparse(X) else csr_matrix(X)
    eng = get_engine(kwargs.get("device", None))
    eng.upload_csr(A)
    member_kw = dict(bootstrap=kwargs.get("bootstrap", True), init=kwargs.get("init", "random"),
                     n_iter=kwargs.get("n_iter", 100), n_iter_per_test=kwargs.get("n_iter_per_test", 10),
                     tolerance=kwargs.get("tolerance", 0.001),
                     e_step_thresh=kwargs.get("e_step_thresh", 1e-16), flags=_member_flags(kwargs))
    random_state = kwargs.get("random_state", None)

    if parallelism == "none":
        topics = [_member_on_engine(eng, k, random_state=random_state, **member_kw) for _ in range(n_runs)]
        return np.vstack(topics)

    from . import distributed
    rank, world = distributed.rank_world()
    if isinstance(random_state, np.random.RandomState):
        # one shared stream cannot be split deterministically across processes: derive per-run seeds
        base_seed = int(random_state.randint(0, 2 ** 31 - 1))
    elif random_state is None:
        base_seed = int(np.random.randint(0, 2 ** 31 - 1)) if world == 1 else distributed.broadcast_seed()
    else:
        base_seed = int(random_state)
    runs = list(range(rank, n_runs, world))
    import time
    t0 = time.perf_counter()
    # the result array is allocated now and its pages are touched by a side thread while the members are fitted (the fill
    # releases the GIL): the one device-to-host copy of the stack then lands in resident pages -- a fresh 424 MB array
    # costs that copy 17.6 ms in page faults against 7.9 ms (profiles/r04_member_stack_to_host_paths.md)
    out, toucher = None, None
    if n_runs % world == 0:
        out = np.empty((n_runs * k, A.shape[1]), np.float32)
        toucher = threading.Thread(target=out.fill, args=(0.0,), daemon=True)
        toucher.start()
    stack_eng = _fit_members(A, k, runs, {r: base_seed + r for r in runs}, member_kw, kwargs.get("device", None),
                             n_jobs, world, n_runs)
    t1 = time.perf_counter()
    if toucher is not None:
        toucher.join()
    out = distributed.gather_stack(stack_eng, n_runs, k, A.shape[1], out=out)
    last_ensemble_timing.update(fit_s=t1 - t0, gather_s=time.perf_counter() - t1, members=len(runs), rank=rank, world=world)
    return out
