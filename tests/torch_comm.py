"""TEST SCAFFOLDING, NOT PRODUCT: the small communicator interface of enstop_amd/comm.py over a torch.distributed
gloo group, so that the N > 1 control flow (run -> rank dealing, the np.vstack gather of enstop/enstop_.py:231, the
accumulator all-reduce of the doc-sharded fit, enstop/distributed_plsa.py:116-131) can run as two real processes on
machines with one GPU or none.  The product's multi-GPU path is comm.RcclComm (RCCL through the C ABI); nothing in
enstop_amd/ imports this module or looks for torch.  Tests install it explicitly:

    dist.init_process_group("gloo", ...); comm.install(TorchComm())
"""
import numpy as np

from enstop_amd.comm import SingleComm


class TorchComm(SingleComm):
    """Host tensors over the caller's gloo process group."""
    name = "torch/gloo"

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self.dist.barrier()

    def allgather_array(self, a):
        a = np.ascontiguousarray(a)
        t = self.torch.from_numpy(a.view(np.uint8).reshape(-1).copy())
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.stack([o.numpy().view(a.dtype).reshape(a.shape) for o in out])

    def allreduce_f64(self, values, op="sum"):
        t = self.torch.tensor(np.atleast_1d(np.asarray(values, np.float64)), dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX)
        return t.numpy()

    def broadcast_array(self, a, root=0):
        a = np.ascontiguousarray(a)
        t = self.torch.from_numpy(a.view(np.uint8).reshape(-1).copy())
        self.dist.broadcast(t, src=root)
        return t.numpy().view(a.dtype).reshape(a.shape)

    def allreduce_accumulator(self, eng):
        t = self.torch.from_numpy(eng.accumulator_get())
        self.dist.all_reduce(t)
        eng.accumulator_set(t.numpy())
