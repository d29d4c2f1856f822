"""Multi-GPU exchange for the two places where the pLSA path talks across GPUs: the stack of the ensemble
members' topic matrices (np.vstack, enstop/enstop_.py:231) and the per-iteration sum of partial factors of a
doc-sharded fit (enstop/distributed_plsa.py:116-131).  One process per GPU.

The product path is `RcclComm`: RCCL collectives issued from the C ABI (plsa_comm_* in
include/plsa_hip.h, librccl linked into libplsa_hip.so) on the engine's own HIP streams -- no PyTorch in
the process.  Only the 128-byte RCCL unique id has to reach the other ranks out of band; on one node that
is a file (`rendezvous_id`).  One more implementation of the same small interface lives here:

  FileComm    host files in a shared directory; a test double (single-GPU boxes cannot host two RCCL ranks:
              RCCL refuses two ranks on one device), never selected automatically

`current()` returns the communicator in force: an explicitly installed one (`install`, `init_from_env`), else the
single-process identity.  Nothing here imports, detects or dispatches on PyTorch (rounds 1-4 auto-selected a
torch.distributed-backed communicator when a process group was initialised; that adapter is test scaffolding now:
tests/torch_comm.py, installed explicitly by the world-size-2 gloo tests).
"""
import os
import sys
import time

import numpy as np

ID_BYTES = 128
ID_MAGIC = b"PLSAID2\n"
TOKEN_BYTES = 120

# Where this rank is in the multi-GPU start-up / exchange sequence; `report_failure` prints it.  Stages, in order:
# "id read" (rendezvous file) -> "comm init" (ncclCommInitRank) -> "warm-up gather" (first collective) ->
# "timed" / "ensemble" (set by the callers around their own collectives).
_STATE = {"stage": "start", "id_file": None, "device": None}


def set_stage(stage):
    _STATE["stage"] = stage


def report_failure(exc, rank=None, world=None, eng=None):
    """ONE line on stderr per rank that says where a multi-GPU run died: rank, device, stage reached, the id file,
    the exception and RCCL's own last error text (ncclGetLastError through the C ABI).  Callers keep the rule
    "non-zero exit status, no JSON line" -- this only makes the first 8-GPU failure cheap to read."""
    if _STATE.get("reported") is exc:          # the same failure travelling up through several handlers: one line
        return None
    _STATE["reported"] = exc
    rank = os.environ.get("RANK", "0") if rank is None else rank
    world = os.environ.get("WORLD_SIZE", "1") if world is None else world
    rccl = ""
    try:
        from . import _lib
        import ctypes as C
        buf = C.create_string_buffer(1024)
        _lib.load().plsa_comm_last_error(eng._h if eng is not None else None, buf, 1024)
        rccl = buf.value.decode(errors="replace")
    except Exception as e:          # never let the reporter mask the real error
        rccl = "unavailable (%s)" % type(e).__name__
    dev = _STATE["device"] if eng is None else getattr(eng, "device", _STATE["device"])
    line = "[enstop_amd rank %s/%s device %s] FAILED at stage '%s'; id file %s; %s: %s; rccl: %s" % (
        rank, world, dev, _STATE["stage"], _STATE["id_file"], type(exc).__name__, str(exc).replace("\n", " | "),
        rccl.replace("\n", " | ") or "-")
    sys.stderr.write(line + "\n")
    sys.stderr.flush()
    return line


def install_termination_reporter(rank=None, world=None, eng=None):
    """When a launcher (torchrun, bench.py's spawner) tears a job down after ONE rank failed, the surviving ranks
    are usually blocked inside a collective -- in C, where a Python-level signal handler never runs.  This installs
    a wake-up descriptor for SIGTERM (written by CPython's C-level handler the moment the signal arrives) and a
    daemon thread that prints this rank's `report_failure` line and exits with status 143.  Main thread only."""
    import signal
    import threading
    if threading.current_thread() is not threading.main_thread():
        return None                                          # signal handlers can only be installed from the main thread
    r, w = os.pipe()
    os.set_blocking(w, False)
    signal.signal(signal.SIGTERM, lambda *a: None)          # keep the default action (silent death) from running
    signal.set_wakeup_fd(w, warn_on_full_buffer=False)

    def watch():
        try:
            os.read(r, 1)
        except OSError:
            return
        report_failure(RuntimeError("terminated by the launcher (SIGTERM): another rank failed or a deadline passed"),
                       rank, world, eng)
        os._exit(143)
    t = threading.Thread(target=watch, name="plsa-termination-reporter", daemon=True)
    t.start()
    return t


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

    def gather_stack(self, eng, slots, k, m, out=None, view=False):
        """The stack of ALL ranks' members in run order, [slots * world, k, m]: `eng` holds this rank's
        members in its device stack (Engine.stack_reserve; run r = slot r // world of rank r % world).
        Communicators without a device path bring the local stack to the host (one copy) and exchange it there.
        `out`: the caller's own result array (re-used across calls).  `view=True` (RCCL / single rank): a VIEW of the
        engine's page-locked buffer instead -- the fastest copy (no page faults of a fresh array), valid until the next
        gather on that engine."""
        if self.world == 1:                                   # nothing to exchange: one device-to-host copy
            return eng.comm_allgather_stack(slots, k, m, copy=not view, out=None if view else out)
        g = self.gather_host_stack(eng.comm_allgather_stack(slots, k, m, copy=False))
        if out is not None:
            out.reshape(g.shape)[...] = g
            return out.reshape(g.shape)
        return g

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

    def gather_stack(self, eng, slots, k, m, out=None, view=False):
        if eng is not self.eng:
            # the members were fitted on another engine than the communicator's (ensemble_of_topics(device=d),
            # init_from_env(eng=custom)): bring that engine's stack to the host and exchange it through RCCL's
            # host path -- slower than the grouped device gather, never wrong
            g = self.gather_host_stack(eng.comm_allgather_stack(slots, k, m, copy=False))
            if out is not None:
                out.reshape(g.shape)[...] = g
                return out.reshape(g.shape)
            return g
        # one grouped ncclAllGather on the engine's stream + one copy (into `out`, a fresh array, or the page-locked buffer)
        return eng.comm_allgather_stack(slots, k, m, copy=not view, out=None if view else out)

    def allreduce_accumulator(self, eng):
        if eng is not self.eng:
            raise RuntimeError("RcclComm is bound to the engine on device %s; the doc-sharded fit ran on device %s. "
                               "Create the communicator on the engine that fits (init_from_env(eng=...))"
                               % (self.eng.device, eng.device))
        eng.allreduce_accumulator()

    def close(self):
        self.eng.comm_destroy()


class FileComm(SingleComm):
    """Test double: exchanges through .npy files in a directory shared by the ranks of one node."""
    name = "files"

    _instances = 0      # communicators this process has created: every rank creates them in the same order

    def __init__(self, directory, rank, world, timeout=300.0):
        # one sub-directory per launch (launch_token: the launcher's nonce / pid) AND per communicator of that launch
        # (a process that shuts one down and creates another -- a test session, a notebook -- gets a fresh one), so
        # that a re-used base directory can never hand a rank the payload of an earlier exchange
        FileComm._instances += 1
        self.dir = os.path.join(directory, "run_%s_c%d" % (launch_token(), FileComm._instances))
        self.rank, self.world, self.timeout = int(rank), int(world), timeout
        self.seq = 0
        os.makedirs(self.dir, exist_ok=True)
        stale = [f for f in os.listdir(self.dir) if f.endswith("_r%d.npy" % self.rank)]
        if stale:
            raise RuntimeError("FileComm: %s already holds exchange files of rank %d (%s ...): a previous run with the "
                               "same launch token; remove the directory" % (self.dir, self.rank, stale[0]))

    def close(self):
        """Final barrier, then every rank leaves a `done` marker -- written AFTER it has read the barrier's payloads --
        and only rank 0 removes files, once all markers are there: nothing a slower rank still polls for or loads can
        disappear under it (round 4 unlinked after a 10 ms sleep and a slow peer spun until its timeout)."""
        if self.seq == 0:
            return
        try:
            self.barrier()
            with open(os.path.join(self.dir, "done_r%d" % self.rank), "w"):
                pass
            if self.rank != 0:
                return
            t0 = time.time()
            while not all(os.path.exists(os.path.join(self.dir, "done_r%d" % r)) for r in range(self.world)):
                if time.time() - t0 > min(self.timeout, 30.0):
                    return                          # a peer died: leave the directory for inspection
                time.sleep(0.001)
            for f in os.listdir(self.dir):
                os.unlink(os.path.join(self.dir, f))
            os.rmdir(self.dir)
        except (OSError, TimeoutError):
            pass

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


_warned_unsharded = False


def current():
    """The installed communicator, else the single-process identity -- no detection of anything else in the process.
    A process that was evidently started as one rank of several (WORLD_SIZE > 1 in the environment: torchrun, mpirun
    wrappers) and never installed a communicator gets ONE warning: every rank would otherwise fit the whole ensemble /
    corpus redundantly -- correct results, no scaling, no sign of it (rounds 1-4 picked up an initialised
    torch.distributed group silently; that dispatch is gone on purpose)."""
    global _warned_unsharded
    if _installed is not None:
        return _installed
    if not _warned_unsharded:
        try:
            world = int(os.environ.get("WORLD_SIZE", "1"))
        except ValueError:
            world = 1
        if world > 1:
            _warned_unsharded = True
            import warnings
            warnings.warn("enstop_amd: WORLD_SIZE=%d but no communicator is installed -- this rank works alone (every rank "
                          "repeats the whole job).  Call enstop_amd.distributed.init() (RCCL, one process per GPU) or "
                          "enstop_amd.comm.install(...) once per process to shard ensembles and doc-sharded fits." % world,
                          RuntimeWarning, stacklevel=2)
    return SingleComm()


def _launcher_nonce():
    """The per-launch nonce a launcher exports: PLSA_LAUNCH_NONCE (bench.py's spawner, or the user) or torchrun's
    TORCHELASTIC_RUN_ID -- except its static-rendezvous constant 'none', which identifies nothing."""
    nonce = os.environ.get("PLSA_LAUNCH_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
    if nonce == "none":
        nonce = ""
    return "".join(ch if ch.isalnum() else "_" for ch in nonce)[:40]


def launch_token():
    """Identifies ONE launch of the ranks: the launcher's own nonce when it exports one (`_launcher_nonce`), always
    combined with the parent pid and the parent's start time -- two launches from the same parent pid (containers
    restart at the same pid) differ in it."""
    return "%s_%d_%d" % (_launcher_nonce() or "x", os.getppid(), int(_parent_start_ticks()))


def _parent_start_ticks():
    """Start time of the parent process in clock ticks since boot (/proc/<ppid>/stat field 22); 0 if unknown."""
    try:
        with open("/proc/%d/stat" % os.getppid()) as f:
            return int(f.read().rsplit(")", 1)[1].split()[19])
    except Exception:
        return 0


def default_id_file():
    """Where the ranks of one launch meet.  PLSA_COMM_ID_FILE wins; otherwise a name that is unique per
    launch (`launch_token`: every worker of one launcher -- torchrun, bench.py's own spawner -- has the same parent)."""
    explicit = os.environ.get("PLSA_COMM_ID_FILE")
    if explicit:
        return explicit
    tmp = os.environ.get("TMPDIR", "/tmp")
    return os.path.join(tmp, "plsa_comm_%s_%s_%s.id" % (os.environ.get("MASTER_ADDR", "local"),
                                                        os.environ.get("MASTER_PORT", "0"), launch_token()))


def _rendezvous_token(path):
    """What the id file must carry for a waiting rank to accept it -- always something only THIS launch knows.
    A real launcher nonce (PLSA_LAUNCH_NONCE, a non-static TORCHELASTIC_RUN_ID) on an explicit PLSA_COMM_ID_FILE is
    compared alone: the ranks may belong to different launchers (a second torchrun, ssh sessions) that share nothing
    else.  In every other case -- the default path, or an explicit path without a nonce (plain
    `torchrun --nproc_per_node=N`: TORCHELASTIC_RUN_ID is the constant 'none') -- the ranks are children of ONE
    launcher and the full launch token (nonce + parent pid + parent start time) is compared, so a file a crashed launch
    left behind is never accepted (round 4 accepted any well-formed file there).  Ranks of different launchers on an
    explicit path therefore need PLSA_LAUNCH_NONCE; a waiting rank says so when it sees a foreign token."""
    nonce = _launcher_nonce()
    if nonce and os.environ.get("PLSA_COMM_ID_FILE") and path == os.environ["PLSA_COMM_ID_FILE"]:
        tok = nonce
    else:
        tok = launch_token()
    return tok.encode()[:TOKEN_BYTES].ljust(TOKEN_BYTES, b"\0")


def rendezvous_id(rank, path=None, timeout=600.0):
    """Rank 0 creates the RCCL unique id and publishes it atomically (after removing whatever an earlier, crashed
    launch left at that path; the file is private to the user) as  magic | launch token | 128 id bytes;  the
    others wait for a file that carries THEIR launch's token -- a stale id would otherwise be read before rank 0
    replaces it and ncclCommInitRank would hang.  No clock is compared (`_rendezvous_token`)."""
    from . import _lib
    path = path or default_id_file()
    _STATE["id_file"] = path
    set_stage("id read")
    token = _rendezvous_token(path)
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
            f.write(ID_MAGIC + token + buf.raw)
        os.replace(path + ".tmp", path)
        return buf.raw
    t0 = time.time()
    want = len(ID_MAGIC) + TOKEN_BYTES + ID_BYTES
    seen = "no file"
    hinted = False
    # An EXPLICIT id file without a launcher nonce is compared by the parent-based launch token: ranks with different
    # parents (static multi-node torchrun, whose TORCHELASTIC_RUN_ID is the constant 'none'; ssh-started ranks sharing a
    # path) can then never match.  They fail after `foreign_grace` seconds of looking at a well-formed foreign token with
    # the remedy in the message, instead of spinning until the rendezvous timeout.
    explicit_without_nonce = bool(os.environ.get("PLSA_COMM_ID_FILE")) and path == os.environ.get("PLSA_COMM_ID_FILE") \
        and not _launcher_nonce()
    foreign_grace = float(os.environ.get("PLSA_RENDEZVOUS_FOREIGN_GRACE", "30"))
    foreign_since = None
    while True:
        try:
            with open(path, "rb") as f:
                data = f.read()
            if len(data) == want and data.startswith(ID_MAGIC):
                if data[len(ID_MAGIC):len(ID_MAGIC) + TOKEN_BYTES] == token:
                    return data[-ID_BYTES:]
                seen = ("a file of another launch (token mismatch: a stale file rank 0 has not replaced yet, or ranks "
                        "started by different launchers -- those must share PLSA_LAUNCH_NONCE)")
                foreign_since = foreign_since or time.time()
                if explicit_without_nonce and time.time() - foreign_since > foreign_grace:
                    raise RuntimeError(
                        "rank %d: the RCCL id file %s has carried another launcher's token for %.0f s.  PLSA_COMM_ID_FILE is set "
                        "explicitly and no launch nonce is exported, so ranks are matched by their PARENT process -- ranks "
                        "started by different launchers (multi-node torchrun with a static rendezvous, ssh sessions) never "
                        "match that way: export the same PLSA_LAUNCH_NONCE=<any string> on every rank (or remove a stale "
                        "file of a crashed launch)." % (rank, path, time.time() - foreign_since))
                if not hinted and time.time() - t0 > 10.0:
                    hinted = True
                    sys.stderr.write("[enstop_amd rank %d] still waiting for this launch's RCCL id at %s; saw %s\n"
                                     % (rank, path, seen))
            else:
                seen = "a malformed / half-written file (%d bytes)" % len(data)
                foreign_since = None
        except FileNotFoundError:
            foreign_since = None
        if time.time() - t0 > timeout:
            raise TimeoutError("rank %d: no RCCL id for this launch at %s after %.0f s (last seen: %s)"
                               % (rank, path, timeout, seen))
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
    _STATE["device"] = eng.device
    timeout = float(os.environ.get("PLSA_RENDEZVOUS_TIMEOUT", "600"))
    try:
        id_bytes = rendezvous_id(rank, path, timeout=timeout)
        set_stage("comm init")
        comm = RcclComm(eng, rank, world, id_bytes)
        set_stage("warm-up gather")
        comm.barrier()
    except Exception as e:
        report_failure(e, rank, world, eng)
        raise
    set_stage("ready")
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
