"""Multi-GPU exchange for the two places where the pLSA path talks across GPUs: the stack of the ensemble
members' topic matrices (np.vstack, enstop/enstop_.py:231) and the per-iteration sum of partial factors of a
doc-sharded fit (enstop/distributed_plsa.py:116-131).  One process per GPU.

The product path is `RcclComm`: RCCL collectives issued from the C ABI (plsa_comm_* in
include/plsa_hip.h, librccl linked into libplsa_hip.so) on the engine's own HIP streams -- no PyTorch in
the process.  Only the 128-byte RCCL unique id has to reach the other ranks out of band; on one node that
is a file (`rendezvous_id`).  Two more implementations of the same small interface exist:

  TorchComm   rides on a torch.distributed process group the CALLER initialised (gloo on CPUs, or nccl);
              optional -- lets the package run inside an existing torch.distributed job, and is what the
              CPU test-suite uses (world_size-2 gloo tests)
  FileComm    host files in a shared directory; a test double (single-GPU boxes cannot host two RCCL ranks:
              RCCL refuses two ranks on one device), never selected automatically

`current()` returns the communicator in force: an explicitly installed one (`install`, `init_from_env`),
else a TorchComm when torch.distributed is initialised, else the single-process identity.
"""
import os
import sys
import time

import numpy as np

ID_BYTES = 128


class SingleComm:
    """world = 1: every exchange is the identity."""
    rank, world, name = 0, 1, "single"

    def barrier(self):
        pass

    def allgather_array(self, a):
        return np.asarray(a)[None]

    def allreduce_f64(self, values, op="sum"):
        return np.atleast_1d(np.asarray(values, np.float64)).copy()

    def broadcast_array(self, a, root=0):
        return np.asarray(a)

    def allgather_components(self, eng):
        _, V = eng.get_factors(want_u=False)
        return V[None]

    def allreduce_accumulator(self, eng):
        pass

    def close(self):
        pass


class RcclComm(SingleComm):
    """RCCL through the C ABI; bound to one Engine (one GPU)."""
    name = "rccl"

    def __init__(self, eng, rank, world, id_bytes):
        self.eng, self.rank, self.world = eng, int(rank), int(world)
        eng.comm_init(id_bytes, rank, world)

    def barrier(self):
        self.eng.comm_barrier()

    def allgather_array(self, a):
        a = np.ascontiguousarray(a)
        return self.eng.comm_allgather_host(a).reshape((self.world,) + a.shape)

    def allreduce_f64(self, values, op="sum"):
        return self.eng.comm_allreduce_f64(values, op)

    def broadcast_array(self, a, root=0):
        return self.eng.comm_broadcast_host(np.ascontiguousarray(a), root)

    def allgather_components(self, eng):
        assert eng is self.eng
        return eng.comm_allgather_components()

    def allreduce_accumulator(self, eng):
        assert eng is self.eng
        eng.allreduce_accumulator()

    def close(self):
        self.eng.comm_destroy()


class TorchComm(SingleComm):
    """Over the caller's torch.distributed process group (gloo: host tensors; nccl: device tensors)."""
    name = "torch"

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.on_device = dist.get_backend() == "nccl"
        self.name = "torch/" + dist.get_backend()

    def _dev(self, eng=None):
        if not self.on_device:
            return self.torch.device("cpu")
        # an Engine lives on LOCAL_RANK / ENSTOP_AMD_DEVICE, which need not be torch's current device
        return self.torch.device("cuda", eng.device if eng is not None else self.torch.cuda.current_device())

    def barrier(self):
        self.dist.barrier()

    def allgather_array(self, a):
        a = np.ascontiguousarray(a)
        t = self.torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(self._dev())
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.stack([o.cpu().numpy().view(a.dtype).reshape(a.shape) for o in out])

    def allreduce_f64(self, values, op="sum"):
        t = self.torch.tensor(np.atleast_1d(np.asarray(values, np.float64)), dtype=self.torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX)
        return t.cpu().numpy()

    def broadcast_array(self, a, root=0):
        a = np.ascontiguousarray(a)
        t = self.torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(self._dev())
        self.dist.broadcast(t, src=root)
        return t.cpu().numpy().view(a.dtype).reshape(a.shape)

    def allgather_components(self, eng):
        _, V = eng.get_factors(want_u=False)
        return self.allgather_array(V)

    def allreduce_accumulator(self, eng):
        torch, dist = self.torch, self.dist
        if self.on_device:
            ptr, n = eng.accumulator_device()

            class _View:        # zero-copy view of the engine's accumulator
                __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}
            eng.synchronize()
            with torch.cuda.device(eng.device):
                t = torch.as_tensor(_View(), device=self._dev(eng))
                assert t.data_ptr() == ptr, "torch copied the accumulator instead of viewing it"
                dist.all_reduce(t)
                torch.cuda.synchronize(eng.device)
        else:
            t = torch.from_numpy(eng.accumulator_get())
            dist.all_reduce(t)
            eng.accumulator_set(t.numpy())


class FileComm(SingleComm):
    """Test double: exchanges through .npy files in a directory shared by the ranks of one node."""
    name = "files"

    def __init__(self, directory, rank, world, timeout=300.0):
        self.dir, self.rank, self.world, self.timeout = directory, int(rank), int(world), timeout
        self.seq = 0
        os.makedirs(directory, exist_ok=True)

    def _exchange(self, a):
        a = np.ascontiguousarray(a)
        self.seq += 1
        mine = os.path.join(self.dir, "x%06d_r%d.npy" % (self.seq, self.rank))
        with open(mine + ".tmp", "wb") as f:
            np.save(f, a)
        os.replace(mine + ".tmp", mine)
        out, t0 = [], time.time()
        for r in range(self.world):
            path = os.path.join(self.dir, "x%06d_r%d.npy" % (self.seq, r))
            while not os.path.exists(path):
                if time.time() - t0 > self.timeout:
                    raise TimeoutError("FileComm: rank %d never wrote %s" % (r, path))
                time.sleep(0.0005)
            out.append(np.load(path))
        return np.stack(out)

    def barrier(self):
        self._exchange(np.zeros(1, np.int8))

    def allgather_array(self, a):
        return self._exchange(a)

    def allreduce_f64(self, values, op="sum"):
        g = self._exchange(np.atleast_1d(np.asarray(values, np.float64)))
        return g.sum(axis=0) if op == "sum" else g.max(axis=0)

    def broadcast_array(self, a, root=0):
        return self._exchange(a)[root]

    def allgather_components(self, eng):
        _, V = eng.get_factors(want_u=False)
        return self._exchange(V)

    def allreduce_accumulator(self, eng):
        g = self._exchange(eng.accumulator_get())
        eng.accumulator_set(g.astype(np.float64).sum(axis=0).astype(np.float32))


# ------------------------------------------------------------------------------------------------
# rendezvous + the communicator in force
# ------------------------------------------------------------------------------------------------
_installed = None


def install(comm):
    """Make `comm` the communicator `current()` returns (None: back to automatic selection)."""
    global _installed
    _installed = comm
    return comm


def current():
    if _installed is not None:
        return _installed
    dist = sys.modules.get("torch.distributed")      # never import torch just to find out
    if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return TorchComm()
    return SingleComm()


def default_id_file():
    """Where the ranks of one launch meet.  PLSA_COMM_ID_FILE wins; otherwise a name that is unique per
    launch: every worker of one launcher (torchrun, bench.py's own spawner) has the same parent process."""
    explicit = os.environ.get("PLSA_COMM_ID_FILE")
    if explicit:
        return explicit
    tmp = os.environ.get("TMPDIR", "/tmp")
    return os.path.join(tmp, "plsa_comm_%s_%s_%d.id" % (os.environ.get("MASTER_ADDR", "local"),
                                                        os.environ.get("MASTER_PORT", "0"), os.getppid()))


def rendezvous_id(rank, path=None, timeout=600.0):
    """Rank 0 creates the RCCL unique id and publishes it atomically; the others wait for the file."""
    from . import _lib
    path = path or default_id_file()
    if rank == 0:
        import ctypes as C
        buf = C.create_string_buffer(ID_BYTES)
        L = _lib.load()
        if L.plsa_comm_unique_id(buf):
            raise RuntimeError(L.plsa_last_error(None).decode())
        with open(path + ".tmp", "wb") as f:
            f.write(buf.raw)
        os.replace(path + ".tmp", path)
        return buf.raw
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                data = f.read()
            if len(data) == ID_BYTES:
                return data
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError("rank %d: no RCCL id at %s after %.0f s" % (rank, path, timeout))
        time.sleep(0.005)


def init_from_env(eng=None, id_file=None):
    """One process per GPU: read RANK / WORLD_SIZE / LOCAL_RANK (torchrun's, or bench.py's own
    spawner's), create the RCCL communicator on this rank's engine and install it.  Returns the
    communicator; with WORLD_SIZE <= 1 the single-process identity."""
    from .engine import get_engine
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world <= 1:
        return install(SingleComm())
    eng = eng if eng is not None else get_engine()
    path = id_file or default_id_file()
    comm = RcclComm(eng, rank, world, rendezvous_id(rank, path))
    comm.barrier()
    if rank == 0:                       # every rank has read the id by now
        try:
            os.unlink(path)
        except OSError:
            pass
    return install(comm)


def shutdown():
    global _installed
    if _installed is not None:
        _installed.close()
    _installed = None
