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


def _run_order(g):
    """[world, slots, k, m] as gathered rank by rank -> [slots * world, k, m] in run order (r = slot * world + rank)."""
    world, slots = g.shape[:2]
    out = np.empty((slots, world) + g.shape[2:], g.dtype)           # always a fresh array (g may view a pinned buffer)
    out[...] = g.transpose(1, 0, 2, 3)
    return out.reshape((slots * world,) + g.shape[2:])


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

    def gather_host_stack(self, local):
        """[slots, k, m] of this rank -> [slots * world, k, m] of all ranks in run order (host arrays)."""
        return _run_order(self.allgather_array(np.asarray(local)))

    def gather_stack(self, eng, slots, k, m):
        """The stack of ALL ranks' members in run order, [slots * world, k, m]: `eng` holds this rank's
        members in its device stack (Engine.stack_reserve; run r = slot r // world of rank r % world).
        Communicators without a device path bring the local stack to the host (one copy) and exchange it there."""
        return self.gather_host_stack(eng.comm_allgather_stack(slots, k, m, copy=False))

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

    def gather_stack(self, eng, slots, k, m):
        assert eng is self.eng            # one grouped ncclAllGather on the engine's stream + one copy to the host
        return eng.comm_allgather_stack(slots, k, m)

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
        # one sub-directory per launch (launch_token: the launcher's nonce / pid), so that a re-used base
        # directory can never hand a rank the payload of an earlier run
        self.dir = os.path.join(directory, "run_" + launch_token())
        self.rank, self.world, self.timeout = int(rank), int(world), timeout
        self.seq = 0
        os.makedirs(self.dir, exist_ok=True)
        stale = [f for f in os.listdir(self.dir) if f.endswith("_r%d.npy" % self.rank)]
        if stale:
            raise RuntimeError("FileComm: %s already holds exchange files of rank %d (%s ...): a previous run with the "
                               "same launch token; remove the directory" % (self.dir, self.rank, stale[0]))

    def _path(self, seq, r):
        return os.path.join(self.dir, "x%06d_r%d.npy" % (seq, r))

    def _exchange(self, a):
        a = np.ascontiguousarray(a)
        self.seq += 1
        mine = self._path(self.seq, self.rank)
        with open(mine + ".tmp", "wb") as f:
            np.save(f, a)
        os.replace(mine + ".tmp", mine)
        out, t0 = [], time.time()
        for r in range(self.world):
            path = self._path(self.seq, r)
            while not os.path.exists(path):
                if time.time() - t0 > self.timeout:
                    raise TimeoutError("FileComm: rank %d never wrote %s" % (r, path))
                time.sleep(0.0005)
            out.append(np.load(path))
        # every rank has written exchange `seq`, hence finished reading `seq - 1`: my older file can go
        try:
            os.unlink(self._path(self.seq - 1, self.rank))
        except OSError:
            pass
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


def launch_token():
    """Identifies ONE launch of the ranks: the launcher's own nonce when it exports one (torchrun's
    TORCHELASTIC_RUN_ID, PLSA_LAUNCH_NONCE from bench.py's spawner), always combined with the parent pid and the
    parent's start time -- two launches from the same parent pid (containers restart at the same pid) differ in it."""
    nonce = os.environ.get("PLSA_LAUNCH_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID") or "x"
    nonce = "".join(ch if ch.isalnum() else "_" for ch in nonce)[:40]
    return "%s_%d_%d" % (nonce, os.getppid(), int(_parent_start_ticks()))


def _parent_start_ticks():
    """Start time of the parent process in clock ticks since boot (/proc/<ppid>/stat field 22); 0 if unknown."""
    try:
        with open("/proc/%d/stat" % os.getppid()) as f:
            return int(f.read().rsplit(")", 1)[1].split()[19])
    except Exception:
        return 0


def _parent_start_epoch():
    """Wall-clock start of the parent process (the launcher): nothing published before it belongs to this launch."""
    try:
        ticks = _parent_start_ticks()
        with open("/proc/stat") as f:
            btime = next(int(line.split()[1]) for line in f if line.startswith("btime"))
        return btime + ticks / float(os.sysconf("SC_CLK_TCK"))
    except Exception:
        return 0.0


def default_id_file():
    """Where the ranks of one launch meet.  PLSA_COMM_ID_FILE wins; otherwise a name that is unique per
    launch (`launch_token`: every worker of one launcher -- torchrun, bench.py's own spawner -- has the same parent)."""
    explicit = os.environ.get("PLSA_COMM_ID_FILE")
    if explicit:
        return explicit
    tmp = os.environ.get("TMPDIR", "/tmp")
    return os.path.join(tmp, "plsa_comm_%s_%s_%s.id" % (os.environ.get("MASTER_ADDR", "local"),
                                                        os.environ.get("MASTER_PORT", "0"), launch_token()))


def rendezvous_id(rank, path=None, timeout=600.0):
    """Rank 0 creates the RCCL unique id and publishes it atomically (after removing whatever an earlier, crashed
    launch left at that path; the file is private to the user); the others wait for a file that is YOUNGER than
    their launcher -- a stale id would otherwise be read before rank 0 replaces it and ncclCommInitRank would hang."""
    from . import _lib
    path = path or default_id_file()
    if rank == 0:
        import ctypes as C
        for old in (path, path + ".tmp"):
            try:
                os.unlink(old)
            except OSError:
                pass
        buf = C.create_string_buffer(ID_BYTES)
        L = _lib.load()
        if L.plsa_comm_unique_id(buf):
            raise RuntimeError(L.plsa_last_error(None).decode())
        fd = os.open(path + ".tmp", os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(buf.raw)
        os.replace(path + ".tmp", path)
        return buf.raw
    not_before = _parent_start_epoch() - 1.0
    t0 = time.time()
    while True:
        try:
            if os.stat(path).st_mtime >= not_before:
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
