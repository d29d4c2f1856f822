"""Doc-sharded single pLSA fit across GPUs (the MI355X-native counterpart of
enstop/distributed_plsa.py, whose dask graph sums per-tile partial factors with
`da.dstack(...).sum(axis=-1)`, distributed_plsa.py:116-131).

Every rank owns a contiguous row range of X (balanced by nnz), its rows of P(z|d) and a full copy of
P(w|z).  One EM iteration is: local fused E+M pass (`plsa_em_accumulate`) -> ONE all-reduce(sum) of
the un-normalised P(w|z) accumulator [m, kp] floats (RCCL over xGMI; 25.6 MB at config 3, ring
time ~0.3 ms) -> identical normalisation on every rank (`plsa_em_finish`).  The log-likelihood test
is a scalar all-reduce every `n_iter_per_test` iterations.  The loop reproduces plsa_fit_inner's stop
semantics (plsa.py:630-638) with the same one-pass-late decision as the single-GPU fused driver.

`Comm` abstracts the two exchanges so that the same loop runs (a) with torch.distributed (nccl =
RCCL on device memory, gloo through the host) one rank per process, and (b) inside one process over
several engines on one device (tests: emulates N ranks without N GPUs).
"""
import numpy as np
from sklearn.utils import check_random_state

from .engine import Engine
from .plsa import plsa_init


def row_ranges_by_nnz(indptr, parts):
    """Contiguous row ranges with ~equal nnz."""
    nnz = int(indptr[-1])
    cuts = [0]
    for r in range(1, parts):
        cuts.append(int(np.searchsorted(indptr, nnz * r / parts)))
    cuts.append(len(indptr) - 1)
    return [(cuts[i], max(cuts[i], cuts[i + 1])) for i in range(parts)]


class LocalComm:
    """All 'ranks' are engines of this process: sums go through host memory (test double)."""

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


class TorchComm:
    """One rank per process over torch.distributed; device all-reduce when the backend is nccl."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.device_path = dist.get_backend() == "nccl"

    def allreduce_accumulators(self, engines):
        (eng,) = engines
        torch, dist = self.torch, self.dist
        if self.device_path:
            ptr, n = eng.accumulator_device()

            class _View:        # zero-copy view of the engine's accumulator for RCCL
                __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}
            t = torch.as_tensor(_View(), device=torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(t)
            torch.cuda.synchronize()
        else:
            t = torch.from_numpy(eng.accumulator_get())
            dist.all_reduce(t)
            eng.accumulator_set(t.numpy())

    def allreduce_scalar(self, values):
        torch, dist = self.torch, self.dist
        t = torch.tensor([float(np.sum(values))], dtype=torch.float64,
                         device="cuda" if self.device_path else "cpu")
        dist.all_reduce(t)
        return float(t.item())


def sharded_em(engines, comm, sample_weights=None, n_iter=100, n_iter_per_test=10, tolerance=0.001,
               e_step_thresh=1e-32, trace_last=False):
    """plsa_fit_inner (plsa.py:583-640) over row shards.  `engines`: the shards local to this process,
    each with its rows uploaded and factors set (same P(w|z) everywhere).  Returns (iterations,
    float32 log-likelihood trace)."""
    sws = sample_weights or [None] * len(engines)
    trace = []
    prev = np.float32(comm.allreduce_scalar([e.log_likelihood(sw) for e, sw in zip(engines, sws)]))
    trace.append(prev)                                             # plsa.py:591
    pending, iters, stopped = False, 0, False
    for i in range(n_iter):
        parts = [e.em_accumulate(sw, e_step_thresh, want_ll=pending) for e, sw in zip(engines, sws)]
        comm.allreduce_accumulators(engines)
        if pending:
            cur = np.float32(comm.allreduce_scalar(parts))
            trace.append(cur)
            with np.errstate(invalid="ignore", divide="ignore"):
                change = np.abs(cur - prev)
                if change == 0 or float(change / np.abs(cur)) < tolerance:   # plsa.py:634-636
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
                     local_shards=None, return_info=False):
    """pLSA fit of X with the documents sharded over the ranks of torch.distributed (one process per
    GPU) or, when `local_shards` is given, over that many engines of this process (single-GPU
    emulation used by the tests).  Every rank passes the same X and arguments; returns the full
    (P(z|d), P(w|z)) on every rank.  Same initial factors as `plsa_fit` for the same seed."""
    X = X.tocsr()
    n, m = X.shape
    rng = check_random_state(random_state)
    U0, V0 = plsa_init(X, k, init=init, rng=rng)                    # global stream, sliced per shard
    U0 = U0.astype(np.float32, order="C"); V0 = V0.astype(np.float32, order="C")
    sw_all = None
    if sample_weight is not None and np.any(np.asarray(sample_weight) != 1.0):
        sw_all = np.asarray(sample_weight, np.float32)

    if local_shards:
        ranges = row_ranges_by_nnz(X.indptr, local_shards)
        engines = [Engine(device) for _ in ranges]
        comm = LocalComm()
        mine = list(range(local_shards))
    else:
        from . import distributed
        rank, world = distributed.rank_world()
        ranges = row_ranges_by_nnz(X.indptr, world)
        engines = [Engine(device)]
        comm = TorchComm() if world > 1 or distributed._dist() is not None else LocalComm()
        mine = [rank]
    try:
        sws = []
        for e, r in zip(engines, mine):
            a, b = ranges[r]
            e.upload_csr(X[a:b])
            e.set_factors(U0[a:b], V0)
            sws.append(None if sw_all is None else sw_all[a:b])
        iters, trace = sharded_em(engines, comm, sws, n_iter, n_iter_per_test, tolerance, e_step_thresh,
                                  trace_last=return_info)
        U = np.zeros((n, k), np.float32)
        V = None
        for e, r in zip(engines, mine):
            a, b = ranges[r]
            Ur, V = e.get_factors()
            U[a:b] = Ur
        if not local_shards and len(ranges) > 1:                  # assemble P(z|d) on every rank
            t = comm.torch.from_numpy(U)
            if comm.device_path:
                t = t.cuda()
            comm.dist.all_reduce(t)                                # disjoint row ranges: sum == concat
            U = t.cpu().numpy()
    finally:
        for e in engines:
            e.close()
    if return_info:
        return U, V, dict(n_iter=iters, log_likelihood_trace=trace)
    return U, V
