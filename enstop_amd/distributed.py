"""Ensemble across GPUs: one process per GPU, run r of the ensemble is fitted by rank r % world, and the
members' topic matrices are brought together by ONE all-gather -- the np.vstack over thread results of
enstop/enstop_.py:231.  No collective sits on the EM data path: members are independent.

The exchange itself lives in `comm.py` (RCCL through the C ABI by default; a caller-initialised
torch.distributed group is honoured).  Typical use under a launcher that sets RANK / WORLD_SIZE /
LOCAL_RANK (torchrun, or `python bench.py --gpus N`):

    import enstop_amd
    enstop_amd.distributed.init()                       # RCCL communicator on this rank's GPU
    topics = enstop_amd.ensemble_of_topics(X, k, n_runs=32)     # identical stack on every rank
"""
import numpy as np

from . import comm as _comm


def init(eng=None, id_file=None):
    """Create and install the RCCL communicator of this rank (no-op with WORLD_SIZE <= 1)."""
    return _comm.init_from_env(eng, id_file)


def shutdown():
    _comm.shutdown()


def rank_world():
    c = _comm.current()
    return c.rank, c.world


def broadcast_seed():
    """One random seed, the same on every rank (drawn by rank 0)."""
    c = _comm.current()
    seed = np.array([np.random.randint(0, 2 ** 31 - 1) if c.rank == 0 else 0], np.int64)
    return int(c.broadcast_array(seed, root=0)[0])


def gather_topics(mine, n_runs, k, m, eng=None):
    """mine: {run index -> (k, m) float32 topics computed by this rank}.  Returns the
    (n_runs * k, m) stack in run order on every rank."""
    c = _comm.current()
    if c.world == 1:
        return np.vstack([mine[r] for r in range(n_runs)])
    per_rank = (n_runs + c.world - 1) // c.world
    send = np.zeros((per_rank, k, m), np.float32)
    for slot, r in enumerate(range(c.rank, n_runs, c.world)):
        send[slot] = mine[r]
    recv = c.allgather_array(send)                     # [world, per_rank, k, m]
    out = np.empty((n_runs * k, m), np.float32)
    for r in range(n_runs):
        out[r * k:(r + 1) * k] = recv[r % c.world, r // c.world]
    return out
