"""Doc-sharded single pLSA fit across GPUs (the MI355X-native counterpart of
enstop/distributed_plsa.py, whose dask graph sums per-tile partial factors with
`da.dstack(...).sum(axis=-1)`, distributed_plsa.py:116-131).

Every rank owns a contiguous row range of X (balanced by nnz), its rows of P(z|d) and a full copy of
P(w|z).  One EM iteration is: local fused E+M pass -> ONE all-reduce(sum) of the un-normalised P(w|z)
accumulator [m, kp] floats (RCCL over xGMI; 25.6 MB at config 3, ring time ~0.3 ms) -> identical
normalisation on every rank.  The log-likelihood test is a scalar all-reduce every `n_iter_per_test`
iterations.  The loop reproduces plsa_fit_inner's stop semantics (plsa.py:630-638) with the same
one-pass-late decision as the single-GPU fused driver.

The loop is three ABI calls per iteration (`plsa_em_accumulate`, `plsa_allreduce_accumulator` -- RCCL on the
engine's stream -- or any external all-reduce of the buffer `plsa_accumulator_device` exposes,
`plsa_em_finish`), driven by `sharded_em` below; that form runs on the RCCL communicator (`distributed.init()`),
on any communicator the caller installs (`comm.install`), and inside one process over several engines on one device (tests: N
shards without N GPUs).  The same loop also exists entirely inside the C ABI -- `plsa_fit(..., PLSA_SHARDED)`:
every collective of the communicator on the context's one stream, no host synchronisation between likelihood
tests -- behind ENSTOP_AMD_SHARDED_INLOOP=1: opt-in until a run on two real GPUs has compared it with the
split form (a one-rank communicator is all a single-GPU box can host).
"""
import os

import numpy as np
from sklearn.utils import check_random_state

from .engine import Engine
from .plsa import plsa_init


def row_ranges_by_nnz(indptr, parts):
    """Contiguous row ranges with ~equal nnz, every range non-empty whenever there are at least `parts`
    rows (a corpus with fewer rows than ranks, or one row holding most of the non-zeros, would otherwise
    leave a rank without rows: its engine cannot hold an empty matrix and its peers would wait for it in
    the all-reduce).  With fewer rows than parts some ranges are necessarily empty; `sharded_plsa_fit`
    then raises the same ValueError on every rank before any exchange."""
    n = len(indptr) - 1
    nnz = int(indptr[-1])
    cuts = [0]
    for r in range(1, parts):
        c = int(np.searchsorted(indptr, nnz * r / parts))
        lo = min(cuts[-1] + 1, n)                      # strictly increasing ...
        hi = max(lo, n - (parts - r))                  # ... and leave one row for each later part
        cuts.append(min(max(c, lo), hi))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(parts)]


class LocalComm:
    """All 'ranks' are engines of this process: sums go through host memory (test double)."""
    rank, world, name = 0, 1, "local"

    def allreduce_accumulators(self, engines):
        total = None
        for e in engines:
            a = e.accumulator_get().astype(np.float64)
            total = a if total is None else total + a
        total = total.astype(np.float32)
        for e in engines:
            e.accumulator_set(total)

    def allreduce_scalar(self, values):
        return float(np.sum(values))


class RankComm:
    """One shard per process: the exchanges go through the communicator in force (comm.current():
    RCCL through the C ABI, or one the caller installed)."""

    def __init__(self, comm):
        self.comm = comm
        self.rank, self.world, self.name = comm.rank, comm.world, comm.name

    def allreduce_accumulators(self, engines):
        (eng,) = engines
        self.comm.allreduce_accumulator(eng)

    def allreduce_scalar(self, values):
        return float(self.comm.allreduce_f64([float(np.sum(values))])[0])


def sharded_em(engines, comm, sample_weights=None, n_iter=100, n_iter_per_test=10, tolerance=0.001,
               e_step_thresh=1e-32, trace_last=False, zero_arm=True, materialised=False):
    """plsa_fit_inner (plsa.py:583-640) over row shards.  `engines`: the shards local to this process,
    each with its rows uploaded and factors set (same P(w|z) everywhere).  Returns (iterations,
    float32 log-likelihood trace)."""
    sws = sample_weights or [None] * len(engines)
    for e, sw in zip(engines, sws):          # one upload per fit, not one copy + host wait per iteration
        e.set_sample_weight(sw)
    try:
        return _sharded_em_loop(engines, comm, n_iter, n_iter_per_test, tolerance, e_step_thresh, trace_last, zero_arm,
                                materialised)
    finally:
        for e in engines:
            e.set_sample_weight(None)


def _sharded_em_loop(engines, comm, n_iter, n_iter_per_test, tolerance, e_step_thresh, trace_last, zero_arm,
                     materialised=False):
    sws = [None] * len(engines)              # the resident weights apply
    trace = []
    prev = np.float32(comm.allreduce_scalar([e.log_likelihood(sw) for e, sw in zip(engines, sws)]))
    trace.append(prev)                                             # plsa.py:591
    pending, iters, stopped = False, 0, False
    for i in range(n_iter):
        parts = [e.em_accumulate(sw, e_step_thresh, want_ll=pending, materialised=materialised) for e, sw in zip(engines, sws)]
        comm.allreduce_accumulators(engines)
        if pending:
            cur = np.float32(comm.allreduce_scalar(parts))
            trace.append(cur)
            with np.errstate(invalid="ignore", divide="ignore"):
                change = np.abs(cur - prev)
                # plsa.py:634-636; zero_arm=False: distributed_plsa.py:277-278 has no `change == 0` arm
                if (zero_arm and change == 0) or float(change / np.abs(cur)) < tolerance:
                    stopped = True
                    break                                          # this pass is discarded
            prev = cur
        for e in engines:
            e.em_finish()
        iters += 1
        pending = (i % n_iter_per_test == 0)
    if pending and not stopped and trace_last:
        # the reference also evaluates the test of the last iteration (plsa.py:630-633); it cannot
        # change the result any more and is only computed to complete the trace
        trace.append(np.float32(comm.allreduce_scalar([e.log_likelihood(sw) for e, sw in zip(engines, sws)])))
    return iters, np.array(trace, np.float32)


def sharded_plsa_fit(X, k, sample_weight=None, init="random", n_iter=100, n_iter_per_test=10,
                     tolerance=0.001, e_step_thresh=1e-32, random_state=None, device=None,
                     local_shards=None, return_info=False, flags=None, zero_arm=True):
    """pLSA fit of X with the documents sharded over the ranks of the communicator in force (one process
    per GPU; `distributed.init()` or `comm.install`) or, when `local_shards` is given, over
    that many engines of this process (single-GPU emulation used by the tests).  Every rank passes the
    same X and arguments; returns the full (P(z|d), P(w|z)) on every rank.  Same initial factors as
    `plsa_fit` for the same seed.

    `local_shards=R` with `flags` that do not contain PLSA_FUSED runs the MATERIALISED schedule tiled by doc blocks
    (block_parallel_plsa.py:373-403): R contexts on one device, one shared P(z|w,d) buffer of the largest block's size.

    Every communicator drives the accumulate / all-reduce / finish split; with the RCCL communicator and
    ENSTOP_AMD_SHARDED_INLOOP=1 the whole loop runs inside the C ABI instead (`plsa_fit` with PLSA_SHARDED)."""
    from . import comm as _comm
    from .engine import PLSA_FUSED, PLSA_SHARDED, PLSA_STOP_NO_ZERO_ARM
    X = X.tocsr()
    n, m = X.shape
    rng = check_random_state(random_state)
    U0, V0 = plsa_init(X, k, init=init, rng=rng)                    # global stream, sliced per shard
    U0 = U0.astype(np.float32, order="C"); V0 = V0.astype(np.float32, order="C")
    sw_all = None
    if sample_weight is not None and np.any(np.asarray(sample_weight) != 1.0):
        sw_all = np.asarray(sample_weight, np.float32)

    native = on_comm_engine = False
    if local_shards:
        ranges = row_ranges_by_nnz(X.indptr, local_shards)
        engines = [Engine(device) for _ in ranges]
        comm = LocalComm()
        mine = list(range(local_shards))
    else:
        c = _comm.current()
        ranges = row_ranges_by_nnz(X.indptr, c.world)
        # ENSTOP_AMD_SHARDED_INLOOP=1: the whole loop inside the C ABI (plsa_fit with PLSA_SHARDED).  Opt-in until a
        # run on two real GPUs has pinned it against the split path below; the default with RCCL is the same three
        # ABI calls per iteration every other communicator uses (accumulate / all-reduce / finish)
        native = isinstance(c, _comm.RcclComm) and os.environ.get("ENSTOP_AMD_SHARDED_INLOOP", "0") == "1"
        on_comm_engine = isinstance(c, _comm.RcclComm)
        # the RCCL communicator belongs to one engine: the sharded fit runs on that engine
        engines = [c.eng if on_comm_engine else Engine(device)]
        comm = RankComm(c)
        mine = [c.rank]
    if any(b <= a for a, b in ranges):
        raise ValueError("sharded_plsa_fit: %d documents cannot be split over %d shards" % (n, len(ranges)))
    # local shards with the MATERIALISED schedule (flags without PLSA_FUSED): the doc-block tiling of
    # block_parallel_plsa.py:373-403 -- every block's P(z|w,d) is computed and consumed inside its own step, and the blocks
    # share ONE buffer sized for the largest of them (config 5: 8 blocks of 32 GB instead of 256 GB)
    tiled = bool(local_shards) and flags is not None and not (flags & PLSA_FUSED)
    lender = -1
    try:
        sws = []
        for e, r in zip(engines, mine):
            a, b = ranges[r]
            e.upload_csr(X[a:b])
            e.set_factors(U0[a:b], V0)
            sws.append(None if sw_all is None else sw_all[a:b])
        if tiled:
            need = [e.p_bytes() for e in engines]
            lender = int(np.argmax(need))
            shared = engines[lender].p_reserve(max(need))
            for i, e in enumerate(engines):
                if i != lender:
                    e.p_borrow(shared, max(need))
        if native:
            fl = (PLSA_FUSED if flags is None else flags) | PLSA_SHARDED | (0 if zero_arm else PLSA_STOP_NO_ZERO_ARM)
            iters, trace = engines[0].fit(sws[0], n_iter, n_iter_per_test, tolerance, e_step_thresh, fl,
                                          trace=return_info)
        else:
            iters, trace = sharded_em(engines, comm, sws, n_iter, n_iter_per_test, tolerance, e_step_thresh,
                                      trace_last=return_info, zero_arm=zero_arm, materialised=tiled)
        U = np.zeros((n, k), np.float32)
        V = None
        for e, r in zip(engines, mine):
            a, b = ranges[r]
            Ur, V = e.get_factors()
            U[a:b] = Ur
        if not local_shards and len(ranges) > 1:                  # assemble P(z|d) on every rank
            parts = comm.comm.allgather_array(_pad_rows(U[ranges[mine[0]][0]:ranges[mine[0]][1]],
                                                        max(b - a for a, b in ranges)))
            for r, (a, b) in enumerate(ranges):
                U[a:b] = parts[r, :b - a]
    finally:
        if tiled and lender >= 0:
            for i, e in enumerate(engines):
                if i != lender:
                    e.p_borrow(None)                   # the loan ends before the lender's buffer goes
        for e in engines:
            if local_shards or not on_comm_engine:
                e.close()
    if return_info:
        return U, V, dict(n_iter=iters, log_likelihood_trace=trace)
    return U, V


def _pad_rows(a, rows):
    out = np.zeros((rows, a.shape[1]), a.dtype)
    out[:a.shape[0]] = a
    return out
