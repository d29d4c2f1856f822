"""Ensemble across GPUs: one process per GPU, run r of the ensemble is fitted by rank r % world, and the
members' topic matrices are brought together by ONE all-gather -- the np.vstack over thread results of
enstop/enstop_.py:231.  No collective sits on the EM data path: members are independent.

The exchange itself lives in `comm.py`: RCCL through the C ABI (or a communicator the caller installed with
`comm.install`); nothing else in the process is consulted.  Typical use under a launcher that sets RANK / WORLD_SIZE /
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


def gather_host_stack(local, n_runs):
    """Host form of `gather_stack`: `local` [slots, k, m] holds this rank's members (slot s = run s * world + rank).
    What the communicators without a device path do after bringing the device stack to the host."""
    c = _comm.current()
    g = c.gather_host_stack(local)
    return g[:n_runs].reshape(n_runs * g.shape[1], g.shape[2])


def gather_stack(eng, n_runs, k, m, out=None, view=False):
    """The (n_runs * k, m) stack of all members in run order on every rank -- the np.vstack of
    enstop/enstop_.py:231.  `eng` holds this rank's members in its device stack (slot s = run s * world + rank,
    written by Engine.copy_components_to_device); with RCCL the stacks are exchanged device to device by ONE
    grouped all-gather and reach the host in ONE copy, straight into the result array
    (plsa_comm_allgather_stack_to).  `out`: a float32 array of n_runs * k * m elements to receive the stack (a caller
    that gathers repeatedly re-uses it; n_runs must then be a multiple of the number of ranks).  `view=True`: the result
    is a view of the engine's page-locked host buffer (no page faults, the fastest copy), valid until the next gather."""
    c = _comm.current()
    slots = (n_runs + c.world - 1) // c.world
    if out is not None and slots * c.world != n_runs:
        raise ValueError("out= needs n_runs (%d) to be a multiple of the number of ranks (%d)" % (n_runs, c.world))
    g = c.gather_stack(eng, slots, k, m, out=out, view=view)       # [slots * world, k, m]
    return g[:n_runs].reshape(n_runs * k, m)
