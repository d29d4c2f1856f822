"""N > 1 plumbing on CPU: world_size-2 gloo processes exercise the run -> rank dealing and the
all-gather that rebuilds the np.vstack order of enstop_.py:231 (the RCCL path uses the same code
with the "nccl" backend)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.environ["REPO_ROOT"])
    from enstop_amd import distributed
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = distributed.rank_world()
    assert (rank, world) == (int(os.environ["RANK"]), 2)
    n_runs, k, m = int(os.environ["N_RUNS"]), 3, 7
    # member r is a deterministic function of r only (what per-run RandomState(seed + r) gives)
    mine = {r: np.full((k, m), float(r), np.float32) + np.arange(m, dtype=np.float32)[None, :] * 0.01
            for r in range(rank, n_runs, world)}
    stack = distributed.gather_topics(mine, n_runs, k, m)
    expect = np.vstack([np.full((k, m), float(r), np.float32) + np.arange(m, dtype=np.float32)[None, :] * 0.01
                        for r in range(n_runs)])
    assert stack.shape == (n_runs * k, m)
    np.testing.assert_array_equal(stack, expect)
    seed = distributed.broadcast_seed()
    seeds = [None, None]
    dist.all_gather_object(seeds, seed)
    assert seeds[0] == seeds[1]
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n_runs", [4, 5])
def test_gather_topics_world2_gloo(tmp_path, n_runs):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), REPO_ROOT=ROOT, N_RUNS=str(n_runs))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\\n%s" % (rank, out)
        assert "ok" in out


def test_single_process_gather_is_vstack():
    from enstop_amd import distributed
    mine = {r: np.random.RandomState(r).rand(2, 5).astype(np.float32) for r in range(3)}
    np.testing.assert_array_equal(distributed.gather_topics(mine, 3, 2, 5), np.vstack([mine[r] for r in range(3)]))
    assert distributed.rank_world() == (0, 1)
