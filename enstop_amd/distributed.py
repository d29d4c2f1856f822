"""Multi-GPU plumbing for the ensemble: one process per GPU (torchrun), torch.distributed for
rendezvous, RCCL ("nccl" backend on ROCm) for the single exchange step -- the all-gather of the
members' topic matrices that replaces np.vstack over thread results (enstop/enstop_.py:231).
No collective sits on the EM data path: members are independent."""
import os
import sys

import numpy as np


def _dist():
    # a process group can only have been initialised by code that imported torch.distributed itself:
    # look it up instead of importing it (importing torch costs seconds, minutes on a cold box, and
    # the single-process ensemble does not need it)
    dist = sys.modules.get("torch.distributed")
    if dist is None:
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def rank_world():
    d = _dist()
    if d is None:
        return 0, 1
    return d.get_rank(), d.get_world_size()


def broadcast_seed():
    import torch
    d = _dist()
    t = torch.zeros(1, dtype=torch.int64)
    if d.get_rank() == 0:
        t[0] = int(np.random.randint(0, 2 ** 31 - 1))
    if d.get_backend() == "nccl":
        t = t.cuda()
    d.broadcast(t, src=0)
    return int(t.item())


def gather_topics(mine, n_runs, k, m, eng=None):
    """mine: {run index -> (k, m) float32 topics computed by this rank}.  Returns the
    (n_runs * k, m) stack in run order on every rank."""
    d = _dist()
    rank, world = rank_world()
    if d is None or world == 1:
        return np.vstack([mine[r] for r in range(n_runs)])
    import torch
    per_rank = (n_runs + world - 1) // world
    use_gpu = d.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_gpu else torch.device("cpu")
    send = torch.zeros((per_rank, k, m), dtype=torch.float32, device=dev)
    for slot, r in enumerate(range(rank, n_runs, world)):
        send[slot].copy_(torch.from_numpy(mine[r]))
    recv = torch.empty((world, per_rank, k, m), dtype=torch.float32, device=dev)
    d.all_gather_into_tensor(recv.view(-1), send.view(-1)) if use_gpu else \
        d.all_gather(list(recv.unbind(0)), send)
    recv = recv.cpu().numpy()
    out = np.empty((n_runs * k, m), np.float32)
    for r in range(n_runs):
        out[r * k:(r + 1) * k] = recv[r % world, r // world]
    return out
