"""N > 1 plumbing on CPU: world_size-2 processes exercise the run -> rank dealing and the all-gather
that rebuilds the np.vstack order of enstop_.py:231, once over a torch.distributed gloo group
(tests/torch_comm.py: test scaffolding the worker installs explicitly -- the product never looks for torch) and once
over the host-file test double (comm.FileComm).  The product communicator
(comm.RcclComm, RCCL through the C ABI) implements the same interface and is covered by the GPU tests;
its rendezvous (a file carrying the RCCL unique id) is exercised here for the waiting ranks."""
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
    sys.path.insert(0, os.path.join(os.environ["REPO_ROOT"], "tests"))
    from enstop_amd import comm, distributed
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    assert distributed.rank_world() == (0, 1)        # an initialised process group alone changes nothing in the product
    assert "torch" in sys.modules and comm.current().name == "single"
    from torch_comm import TorchComm
    comm.install(TorchComm())
    rank, world = distributed.rank_world()
    assert (rank, world) == (int(os.environ["RANK"]), 2)
    n_runs, k, m = int(os.environ["N_RUNS"]), 3, 7
    # member r is a deterministic function of r only (what per-run RandomState(seed + r) gives)
    mine = {r: np.full((k, m), float(r), np.float32) + np.arange(m, dtype=np.float32)[None, :] * 0.01
            for r in range(rank, n_runs, world)}
    # this rank's member stack as it sits on the device: run r in slot r // world (unfilled slots hold anything)
    local = np.full(((n_runs + world - 1) // world, k, m), -7.0, np.float32)
    for r, v in mine.items():
        local[r // world] = v
    stack = distributed.gather_host_stack(local, n_runs)
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
def test_gather_stack_world2_gloo(tmp_path, n_runs):
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
    local = np.stack([mine[r] for r in range(3)])
    got = distributed.gather_host_stack(local, 3)
    np.testing.assert_array_equal(got, np.vstack([mine[r] for r in range(3)]))
    assert not np.shares_memory(got, local)
    assert distributed.rank_world() == (0, 1)


FILE_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["REPO_ROOT"])
    from enstop_amd import comm, distributed
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    c = comm.install(comm.FileComm(os.environ["XDIR"], rank, world))
    assert distributed.rank_world() == (rank, world)
    n_runs, k, m = 5, 3, 7
    mine = {r: np.full((k, m), float(r), np.float32) for r in range(rank, n_runs, world)}
    local = np.full(((n_runs + world - 1) // world, k, m), -7.0, np.float32)
    for r, v in mine.items():
        local[r // world] = v
    stack = distributed.gather_host_stack(local, n_runs)
    np.testing.assert_array_equal(stack, np.vstack([np.full((k, m), float(r), np.float32) for r in range(n_runs)]))
    assert c.allreduce_f64([rank + 1.0, 10.0])[0] == 3.0 and c.allreduce_f64([rank + 1.0], "max")[0] == 2.0
    got = c.broadcast_array(np.arange(4) + 100 * rank, root=1)
    np.testing.assert_array_equal(got, np.arange(4) + 100)
    seed = distributed.broadcast_seed()
    assert list(c.allgather_array(np.array([seed]))[:, 0]) == [seed, seed]
    c.barrier()
    print("rank", rank, "ok")
""")


def test_file_comm_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(FILE_WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", REPO_ROOT=ROOT, XDIR=str(tmp_path / "x"))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "ok" in out, "rank %d failed:\n%s" % (rank, out)


def _publish(comm, path, payload, token=None, delay=0.2):
    """what rank 0 does in comm.rendezvous_id, without needing a GPU for ncclGetUniqueId"""
    import time
    time.sleep(delay)
    for old in (path, path + ".tmp"):
        try:
            os.unlink(old)
        except OSError:
            pass
    token = comm._rendezvous_token(path) if token is None else token
    fd = os.open(path + ".tmp", os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    with os.fdopen(fd, "wb") as f:
        f.write(comm.ID_MAGIC + token + payload)
    os.replace(path + ".tmp", path)


def test_rendezvous_file_for_waiting_ranks(tmp_path, monkeypatch):
    """Ranks > 0 wait for the file rank 0 publishes atomically; the default name is unique per launcher."""
    import threading
    from enstop_amd import comm
    monkeypatch.delenv("PLSA_COMM_ID_FILE", raising=False)
    path = str(tmp_path / "rccl.id")
    payload = bytes(range(128))
    t = threading.Thread(target=_publish, args=(comm, path, payload))
    t.start()
    assert comm.rendezvous_id(1, path, timeout=30) == payload
    t.join()
    assert (os.stat(path).st_mode & 0o777) == 0o600
    with pytest.raises(TimeoutError):
        comm.rendezvous_id(1, str(tmp_path / "never.id"), timeout=0.2)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29999")
    monkeypatch.delenv("PLSA_LAUNCH_NONCE", raising=False)
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    name = comm.default_id_file()
    assert "127.0.0.1" in name and "29999" in name and str(os.getppid()) in name
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "run/42")                 # the launcher's nonce is part of the name
    assert "run_42" in comm.default_id_file() and comm.default_id_file() != name
    monkeypatch.setenv("PLSA_COMM_ID_FILE", "/some/where.id")
    assert comm.default_id_file() == "/some/where.id"
    assert isinstance(comm.current(), comm.SingleComm) and comm.current().world == 1


def test_stale_rendezvous_file_is_ignored(tmp_path, monkeypatch):
    """A crashed earlier launch may leave an id at the very path this launch uses.  Waiting ranks must not take it:
    the file carries the launch token of whoever wrote it and only THIS launch's token is accepted (no clock is
    compared: ADVICE r03 -- a launcher that starts after rank 0 published, or a file system whose clock lags, made
    the mtime rule reject valid ids); rank 0 removes what it finds before it publishes."""
    import threading
    from enstop_amd import comm
    monkeypatch.delenv("PLSA_COMM_ID_FILE", raising=False)
    path = str(tmp_path / "rccl.id")
    other_launch = b"x_1_1".ljust(comm.TOKEN_BYTES, b"\0")
    _publish(comm, path, bytes([7]) * 128, token=other_launch, delay=0)
    with pytest.raises(TimeoutError, match="token mismatch"):
        comm.rendezvous_id(1, path, timeout=0.3)
    with open(path, "wb") as f:                                         # the pre-round-4 format: 128 raw bytes
        f.write(bytes([7]) * 128)
    with pytest.raises(TimeoutError, match="malformed"):
        comm.rendezvous_id(1, path, timeout=0.3)
    fresh = bytes(range(128))
    t = threading.Thread(target=_publish, args=(comm, path, fresh))
    t.start()
    assert comm.rendezvous_id(1, path, timeout=30) == fresh
    t.join()
    # the file may be much OLDER than the waiting rank's launcher and is still accepted (same token)
    os.utime(path, (1.0, 1.0))
    assert comm.rendezvous_id(1, path, timeout=1) == fresh


def test_explicit_id_file_shared_by_two_launchers(tmp_path, monkeypatch):
    """PLSA_COMM_ID_FILE shared by ranks whose parents differ (a second torchrun, ssh sessions): with a launcher nonce
    the parent pid is not part of the token -- only the nonce is compared.  WITHOUT a nonce (plain
    `torchrun --nproc_per_node=N`: TORCHELASTIC_RUN_ID is the constant 'none') the token never degenerates to "any
    well-formed file" (ADVICE r04): the ranks are children of one launcher and the full launch token applies, so an id
    a crashed launch left at the explicit path is refused."""
    from enstop_amd import comm
    path = str(tmp_path / "shared.id")
    monkeypatch.setenv("PLSA_COMM_ID_FILE", path)
    monkeypatch.delenv("PLSA_LAUNCH_NONCE", raising=False)
    for static_id in (None, "none"):
        if static_id is None:
            monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
        else:
            monkeypatch.setenv("TORCHELASTIC_RUN_ID", static_id)
        tok = comm._rendezvous_token(path)
        assert tok == comm.launch_token().encode().ljust(comm.TOKEN_BYTES, b"\0") and str(os.getppid()).encode() in tok
        stale = b"x_1_1".ljust(comm.TOKEN_BYTES, b"\0")                       # what another (dead) launcher wrote there
        _publish(comm, path, bytes([9]) * 128, token=stale, delay=0)
        with pytest.raises(TimeoutError, match="token mismatch"):
            comm.rendezvous_id(1, path, timeout=0.3)
        _publish(comm, path, bytes(range(128)), delay=0)
        assert comm.rendezvous_id(1, path, timeout=1) == bytes(range(128))
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    monkeypatch.setenv("PLSA_LAUNCH_NONCE", "job7")
    assert comm._rendezvous_token(path) == b"job7".ljust(comm.TOKEN_BYTES, b"\0")
    with pytest.raises(TimeoutError, match="token mismatch"):          # written without the nonce: another launch
        comm.rendezvous_id(1, path, timeout=0.3)
    _publish(comm, path, bytes(range(128)), delay=0)
    assert comm.rendezvous_id(1, path, timeout=1) == bytes(range(128))


def test_file_comm_instances_and_cleanup(tmp_path, monkeypatch):
    from enstop_amd import comm
    monkeypatch.setenv("PLSA_LAUNCH_NONCE", "abc")
    c = comm.FileComm(str(tmp_path / "x"), 0, 1)
    c.barrier(); c.barrier()
    assert len([f for f in os.listdir(c.dir) if f.endswith(".npy")]) == 1   # older exchanges are deleted
    d1 = c.dir
    c.close()                                                               # final barrier, last file and directory go
    assert not os.path.exists(d1)
    c2 = comm.FileComm(str(tmp_path / "x"), 0, 1)                           # same launch, same process: a fresh directory
    assert c2.dir != d1
    c2.barrier()
    # a directory that already holds this rank's files (a crashed run with the same token and instance number) is refused
    comm.FileComm._instances -= 1
    with pytest.raises(RuntimeError):
        comm.FileComm(str(tmp_path / "x"), 0, 1)
    monkeypatch.setenv("PLSA_LAUNCH_NONCE", "def")
    comm.FileComm(str(tmp_path / "x"), 0, 1).barrier()                      # a new launch gets a fresh directory


def test_failure_line_names_rank_stage_and_id_file(tmp_path, monkeypatch, capsys):
    """a rank that dies in the multi-GPU start-up says where.  Here: the rendezvous times out
    (rank 0 never publishes, e.g. it was killed); the line carries rank, world, device, stage and the id file."""
    from enstop_amd import comm

    class FakeEngine:
        device = 3
        _h = None
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("PLSA_RENDEZVOUS_TIMEOUT", "0.3")
    monkeypatch.delenv("PLSA_COMM_ID_FILE", raising=False)
    path = str(tmp_path / "never.id")
    with pytest.raises(TimeoutError):
        comm.init_from_env(eng=FakeEngine(), id_file=path)
    err = capsys.readouterr().err
    line = [ln for ln in err.splitlines() if ln.startswith("[enstop_amd rank 1/2 device 3]")]
    assert len(line) == 1, err
    assert "stage 'id read'" in line[0] and path in line[0] and "TimeoutError" in line[0] and "rccl:" in line[0]
    comm.install(None)


def test_terminated_rank_reports_its_stage(tmp_path):
    """A launcher tears the job down with SIGTERM while this rank sits inside a C call (a collective): the
    termination reporter still prints the stage line and the rank exits with 143."""
    import signal
    import time
    code = (
        "import ctypes, os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "from enstop_amd import comm\n"
        "comm.install_termination_reporter(rank=3, world=8)\n"
        "comm._STATE.update(device=3, id_file='/tmp/x.id')\n"
        "comm.set_stage('timed')\n"
        "print('ready', flush=True)\n"
        "libc = ctypes.CDLL(None)\n"
        "t0 = time.time()\n"
        "while time.time() - t0 < 60: libc.usleep(200000)   # blocked in C, like a rank inside ncclAllGather\n"
    ) % ROOT
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == "ready"
    time.sleep(0.3)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=30)
    assert p.returncode == 143, (p.returncode, err)
    line = [ln for ln in err.splitlines() if ln.startswith("[enstop_amd rank 3/8 device 3]")]
    assert len(line) == 1 and "stage 'timed'" in line[0] and "SIGTERM" in line[0], err


def test_bench_refuses_gpus_world_mismatch(tmp_path):
    """`bench.py --gpus N` under a launcher whose WORLD_SIZE differs must fail, not report n_gpus wrongly."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and out.stdout.strip() == ""
