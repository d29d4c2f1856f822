"""Engine: one MI355X + one HIP stream + the device-resident corpus and factors.

Thin object wrapper over the C ABI (include/plsa_hip.h); every method is one or two ABI calls.
The reference keeps all of this state in NumPy arrays handed from function to function
(enstop/plsa.py:707-730); here it lives in HBM for the lifetime of the Engine.
"""
import ctypes as C
import os
import threading

import numpy as np

from . import _lib
from ._lib import (PLSA_FUSED, PLSA_REFERENCE_LL, PLSA_REFERENCE_SUMS, PLSA_SHARDED, PLSA_STOP_NO_ZERO_ARM, PLSA_SW_LL_ONLY,
                   PLSA_TRACE_LL, ptr)


class DeviceError(RuntimeError):
    pass


def default_device():
    """One process per GPU: LOCAL_RANK (torchrun / bench.py's spawner) selects the device unless
    ENSTOP_AMD_DEVICE overrides it; a launcher that narrows the visible devices to one per rank
    (HIP_VISIBLE_DEVICES) leaves only device 0."""
    if "ENSTOP_AMD_DEVICE" in os.environ:
        return int(os.environ["ENSTOP_AMD_DEVICE"])
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    if dev > 0:
        cnt = C.c_int(0)
        if _lib.load().plsa_device_count(C.byref(cnt)) == 0 and 0 < cnt.value <= dev:
            if cnt.value != 1:
                raise RuntimeError("LOCAL_RANK=%d but only %d GPUs are visible: start one rank per visible GPU, "
                                   "narrow the devices to one per rank (HIP_VISIBLE_DEVICES), or set "
                                   "ENSTOP_AMD_DEVICE" % (dev, cnt.value))
            dev = 0
    return dev


def arithmetic_flags(arithmetic):
    """`arithmetic=` of the fit functions -> flag bits.  None / "default": the engine's own summation orders (float64
    norm_pwz and log-likelihood; more accurate than the reference).  "reference": PLSA_REFERENCE_SUMS -- every sum of the
    E- and M-step one float32 accumulator in the reference's loop order: the bits of enstop/plsa.py executed statement
    by statement; the log-likelihood stays float64-accumulated (what the numba-compiled reduction delivers to 1e-7).
    "reference_source": additionally PLSA_REFERENCE_LL -- the log-likelihood as ONE float32 running sum (plsa.py:322
    read literally, one thread).  An int passes through."""
    if arithmetic is None or arithmetic == "default":
        return 0
    if isinstance(arithmetic, (int, np.integer)):
        if int(arithmetic) & ~(PLSA_REFERENCE_SUMS | PLSA_REFERENCE_LL):
            raise ValueError("arithmetic bits must be PLSA_REFERENCE_SUMS | PLSA_REFERENCE_LL")
        return int(arithmetic)
    if arithmetic == "reference":
        return PLSA_REFERENCE_SUMS
    if arithmetic == "reference_source":
        return PLSA_REFERENCE_SUMS | PLSA_REFERENCE_LL
    raise ValueError('arithmetic must be None, "default", "reference" or "reference_source", not %r' % (arithmetic,))


def default_flags():
    """Fit schedule: fused (P(z|w,d) never touches HBM) unless ENSTOP_AMD_MATERIALISE=1, which
    follows the reference's E-step -> M-step kernel sequence through a materialised nnz x k array.
    ENSTOP_AMD_ARITHMETIC=reference / reference_source selects the reference's rounding (arithmetic_flags)
    for every fit that does not say otherwise."""
    flags = 0 if os.environ.get("ENSTOP_AMD_MATERIALISE", "0") == "1" else PLSA_FUSED
    return flags | arithmetic_flags(os.environ.get("ENSTOP_AMD_ARITHMETIC") or None)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Engine:
    def __init__(self, device=None):
        self._L = _lib.load()
        self._h = C.c_void_p()
        dev = default_device() if device is None else int(device)
        if self._L.plsa_create(dev, C.byref(self._h)):
            raise DeviceError(self._L.plsa_last_error(None).decode())
        self.device = dev
        self.k = 0
        self.base_rows = 0
        # a context is not thread-safe: callers that share the process-wide engine (the reference is
        # driven from dask/joblib thread pools, enstop_.py:209-217) serialise on this lock
        self.lock = threading.RLock()

    # -- lifetime --------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.plsa_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ok(self, rc):
        if rc:
            raise DeviceError(self._L.plsa_last_error(self._h).decode())

    def synchronize(self):
        self._ok(self._L.plsa_synchronize(self._h))

    def device_info(self):
        name, arch = C.create_string_buffer(64), C.create_string_buffer(64)
        cus, hbm = C.c_int(0), C.c_int64(0)
        self._ok(self._L.plsa_device_info(self._h, name, arch, C.byref(cus), C.byref(hbm)))
        return dict(name=name.value.decode(), arch=arch.value.decode(), cus=cus.value, hbm_bytes=hbm.value)

    # -- corpus ----------------------------------------------------------------------------------
    def upload_csr(self, X):
        """X: scipy CSR (any dtype; values are cast to float32 like plsa.py:714)."""
        X = X.tocsr()
        n, m = X.shape
        # the device layout is the reference's: int32 row pointers / indices (plsa.py:26 `i4[::1]`, its loop counters are
        # uint32).  scipy switches to int64 index arrays on its own for large matrices: refuse what does not fit
        # instead of letting the cast below wrap around
        if X.nnz > 2**31 - 64 or max(n, m) >= 2**31 - 1:
            raise ValueError("matrix too large for 32-bit indices: %d x %d with %d stored entries (limit 2^31 - 64 entries)"
                             % (n, m, X.nnz))
        indptr = np.ascontiguousarray(X.indptr, dtype=np.int32)
        indices = np.ascontiguousarray(X.indices, dtype=np.int32)
        data = _f32(X.data)
        # the index contract (0 <= column < m, row pointers non-decreasing) is checked ON THE DEVICE by one streaming pass over
        # the copy (k_validate_csr): a host-side min() / max() over the indices of config 3 cost more than the upload itself
        # (62 ms for 100 M entries on the build container).  A violation is the reference-style ValueError either way.
        if self._L.plsa_upload_csr(self._h, indptr, indices, data, n, m, data.shape[0]):
            msg = self._L.plsa_last_error(self._h).decode()
            if "column index" in msg or "indptr" in msg:
                raise ValueError("column index out of range" if "column index" in msg else msg)
            raise DeviceError(msg)
        self.base_rows = n
        return self

    def generate_synthetic(self, n, m, nnz, zipf_s=1.07, seed=0, topics=0, alpha=0.1, background=0.25):
        """Synthetic bag-of-words corpus generated in HBM (becomes base + active matrix); returns the exact nnz.
        topics = 0: every token an independent Zipf draw.  topics = k0 > 0: documents are Dirichlet(alpha) mixtures
        of k0 latent topics with their own Zipf rankings plus a shared `background` ranking (text-like co-occurrence)."""
        out = C.c_int64(0)
        if topics:
            self._ok(self._L.plsa_generate_synthetic_topics(self._h, n, m, nnz, float(zipf_s), int(seed), int(topics),
                                                            float(alpha), float(background), C.byref(out)))
        else:
            self._ok(self._L.plsa_generate_synthetic(self._h, n, m, nnz, float(zipf_s), int(seed), C.byref(out)))
        self.base_rows = n
        return out.value

    def synthetic_dominant_topics(self):
        """Ground truth of the topical synthetic corpus held as the base matrix: the dominant latent topic of every document."""
        out = np.empty(self.base_rows, np.int32)
        self._ok(self._L.plsa_synthetic_dominant_topics(self._h, out))
        return out

    def bootstrap(self, idx):
        """active := base[idx] on the device (enstop_.py:87-88); idx=None restores the base."""
        if idx is None:
            self._ok(self._L.plsa_bootstrap(self._h, None, 0))
        else:
            idx = np.ascontiguousarray(idx, dtype=np.int64)
            self._ok(self._L.plsa_bootstrap(self._h, idx.ctypes.data, idx.shape[0]))
        return self

    @property
    def shape(self):
        n, m, nnz = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._ok(self._L.plsa_active_shape(self._h, C.byref(n), C.byref(m), C.byref(nnz)))
        return n.value, m.value, nnz.value

    def download_active_csr(self):
        import scipy.sparse as sp
        n, m, nnz = self.shape
        indptr = np.empty(n + 1, np.int32)
        indices = np.empty(max(nnz, 1), np.int32)
        data = np.empty(max(nnz, 1), np.float32)
        self._ok(self._L.plsa_download_active_csr(self._h, indptr, indices, data))
        return sp.csr_matrix((data[:nnz], indices[:nnz], indptr), shape=(n, m))

    # -- factors ---------------------------------------------------------------------------------
    def set_factors(self, p_z_given_d, p_w_given_z=None):
        U = _f32(p_z_given_d)
        n, k = U.shape
        _, m, _ = self.shape
        V = None
        if p_w_given_z is not None:
            V = _f32(p_w_given_z)
            if V.shape != (k, m):
                raise ValueError("p_w_given_z has shape %s, expected %s" % (V.shape, (k, m)))
        self._ok(self._L.plsa_set_factors(self._h, ptr(U), ptr(V), n, m, k))
        self.k = k
        return self

    def init_factors_device(self, k, seed=0):
        """Throughput-mode random init on the device (not the reference's MT19937 stream)."""
        self._ok(self._L.plsa_init_factors_device(self._h, int(k), int(seed) & (2 ** 64 - 1)))
        self.k = int(k)
        return self

    def init_factors_numpy_stream(self, k, rng, topics=None):
        """plsa_init(random) + float32 casts on the device, drawing from `rng` (a legacy
        numpy.random.RandomState, MT19937) exactly as rng.rand(k, m); rng.rand(n, k) would; `rng`
        is left in the state the host path would leave it in.  With `topics` ([k, m], fixed) only
        rng.rand(n, k) is drawn: the initialisation of plsa_refit (plsa.py:979-981)."""
        kind, key, pos, has_gauss, cached = rng.get_state()
        if kind != "MT19937":
            raise ValueError("not an MT19937 RandomState")
        state = np.empty(625, np.uint32)
        state[:624] = key
        state[624] = pos
        if topics is None:
            self._ok(self._L.plsa_init_factors_mt19937(self._h, int(k), state))
        else:
            V = _f32(topics)
            if V.shape[0] != int(k):
                raise ValueError("topics have %d rows, expected k=%d" % (V.shape[0], int(k)))
            self._ok(self._L.plsa_refit_init_mt19937(self._h, ptr(V), V.shape[1], int(k), state))
        rng.set_state((kind, state[:624].copy(), int(state[624]), has_gauss, cached))
        self.k = int(k)
        return self

    def mt_marginals(self):
        """float64 row sums of the topic draws of the last init_factors_numpy_stream call (diagnostics / tests)."""
        out = np.empty(self.k, np.float64)
        self._ok(self._L.plsa_mt_marginals(self._h, out, int(self.k)))
        return out

    def get_factors(self, want_u=True, want_v=True):
        n, m, _ = self.shape
        U = np.empty((n, self.k), np.float32) if want_u else None
        V = np.empty((self.k, m), np.float32) if want_v else None
        self._ok(self._L.plsa_get_factors(self._h, ptr(U), ptr(V)))
        return U, V

    def copy_components_to_device(self, device_ptr):
        self._ok(self._L.plsa_copy_components_to_device(self._h, int(device_ptr)))

    # -- kernel-level operators --------------------------------------------------------------------
    def set_arithmetic(self, arithmetic=None):
        """Arithmetic of e_step / m_step / log_likelihood and of later fits on this engine (arithmetic_flags)."""
        self._ok(self._L.plsa_set_arithmetic(self._h, arithmetic_flags(arithmetic)))
        return self

    def e_step(self, thresh=1e-32, out=None, want_host_copy=True):
        _, _, nnz = self.shape
        if want_host_copy and out is None:
            out = np.empty((nnz, self.k), np.float32)
        self._ok(self._L.plsa_e_step(self._h, np.float32(thresh), ptr(out) if want_host_copy else None))
        return out

    def set_p(self, P):
        self._ok(self._L.plsa_set_p(self._h, _f32(P)))

    def m_step(self, sample_weight=None, update_v=True):
        n, _, _ = self.shape
        sw = None if sample_weight is None else _f32(sample_weight)
        npwz = np.zeros(self.k, np.float32)
        npdz = np.zeros(n, np.float32)
        self._ok(self._L.plsa_m_step(self._h, ptr(sw), int(update_v), ptr(npwz), ptr(npdz)))
        return npwz, npdz

    def log_likelihood(self, sample_weight=None):
        sw = None if sample_weight is None else _f32(sample_weight)
        out = C.c_double(0.0)
        self._ok(self._L.plsa_log_likelihood(self._h, ptr(sw), C.byref(out)))
        return out.value

    # -- EM drivers --------------------------------------------------------------------------------
    def _drive(self, fn, sample_weight, n_iter, n_iter_per_test, tolerance, thresh, flags, trace):
        sw = None if sample_weight is None else _f32(sample_weight)
        ll = np.zeros(n_iter + 2, np.float32)
        iters, nll = C.c_int32(0), C.c_int32(0)
        if trace:
            flags |= PLSA_TRACE_LL
        self._ok(fn(self._h, ptr(sw), int(n_iter), int(n_iter_per_test), float(tolerance),
                    np.float32(thresh), int(flags), C.byref(iters), ll.ctypes.data, C.byref(nll)))
        return iters.value, ll[: nll.value].copy()

    def fit(self, sample_weight=None, n_iter=100, n_iter_per_test=10, tolerance=0.001,
            e_step_thresh=1e-32, flags=None, trace=False):
        flags = default_flags() if flags is None else flags
        return self._drive(self._L.plsa_fit, sample_weight, n_iter, n_iter_per_test, tolerance,
                           e_step_thresh, flags, trace)

    def refit(self, sample_weight=None, n_iter=50, n_iter_per_test=10, tolerance=0.005,
              e_step_thresh=1e-32, flags=None, trace=False):
        flags = default_flags() if flags is None else flags
        return self._drive(self._L.plsa_refit, sample_weight, n_iter, n_iter_per_test, tolerance,
                           e_step_thresh, flags, trace)

    # -- doc-sharded fit building blocks ---------------------------------------------------------------
    def em_accumulate(self, sample_weight=None, e_step_thresh=1e-32, want_ll=False, materialised=False):
        """materialised: the reference's kernel sequence over this context's rows (E-step into P(z|w,d), M-step from it)
        instead of the fused passes -- one doc block of a tiled materialised iteration (plsa_em_accumulate_materialised)."""
        sw = None if sample_weight is None else _f32(sample_weight)
        ll = C.c_double(0.0)
        fn = self._L.plsa_em_accumulate_materialised if materialised else self._L.plsa_em_accumulate
        self._ok(fn(self._h, ptr(sw), np.float32(e_step_thresh), C.addressof(ll) if want_ll else None))
        return ll.value if want_ll else None

    def p_bytes(self):
        """bytes of P(z|w,d) for the active matrix and the factors in force (one tile of slack included)"""
        n, m, nnz = self.shape
        return 4 * (nnz + 64) * ((self.k + 3) // 4 * 4)

    def p_reserve(self, nbytes):
        """own P(z|w,d) capacity of at least nbytes; returns its device address (for p_borrow on other contexts)"""
        out = C.c_void_p()
        self._ok(self._L.plsa_p_reserve(self._h, int(nbytes), C.byref(out)))
        return out.value

    def p_borrow(self, device_ptr, nbytes=0):
        """use another context's P(z|w,d) buffer (None: back to own allocations); sharers run one after the other"""
        self._ok(self._L.plsa_p_borrow(self._h, device_ptr, int(nbytes)))

    def em_finish(self):
        self._ok(self._L.plsa_em_finish(self._h))

    def set_sample_weight(self, sample_weight=None):
        """Make the document weights resident on the device (None: clear): later calls that pass no weights use
        them without a host copy -- the per-iteration calls of the doc-sharded loop stay enqueue-only."""
        sw = None if sample_weight is None else _f32(sample_weight)
        self._ok(self._L.plsa_set_sample_weight(self._h, ptr(sw)))

    def accumulator_device(self):
        p, n = C.c_void_p(), C.c_int64(0)
        self._ok(self._L.plsa_accumulator_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def accumulator_get(self):
        _, n = self.accumulator_device()
        out = np.empty(n, np.float32)
        self._ok(self._L.plsa_accumulator_get(self._h, out))
        return out

    def accumulator_set(self, a):
        self._ok(self._L.plsa_accumulator_set(self._h, _f32(a)))

    # -- multi-GPU exchange (RCCL through the C ABI; see comm.py) ------------------------------------------
    def comm_init(self, id_bytes, rank, world):
        if len(id_bytes) != 128:
            raise ValueError("the RCCL unique id has 128 bytes")
        self._ok(self._L.plsa_comm_init(self._h, bytes(id_bytes), int(rank), int(world)))

    def comm_destroy(self):
        self._ok(self._L.plsa_comm_destroy(self._h))

    def comm_info(self):
        r, w = C.c_int32(0), C.c_int32(1)
        self._ok(self._L.plsa_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_barrier(self):
        self._ok(self._L.plsa_comm_barrier(self._h))

    def stack_reserve(self, slots, k, m):
        """Device block [slots][k][m] for the topic matrices of the members this process fits; returns its
        device address (slot s lives at base + 4 * s * k * m)."""
        base = C.c_void_p()
        self._ok(self._L.plsa_stack_reserve(self._h, int(slots), int(m), int(k), C.byref(base)))
        return int(base.value)

    def comm_allgather_stack(self, slots, k, m, copy=True, out=None):
        """[slots * world, k, m] float32 in run order (run r = slot r // world of rank r % world): one grouped
        ncclAllGather, then ONE device-to-host copy -- straight into `out` (a C-contiguous float32 array of that size the
        caller re-uses), into a fresh array (copy=True), or into the engine's page-locked buffer of which a VIEW is
        returned (copy=False: overwritten by the next call on this engine)."""
        _, world = self.comm_info()
        shape = (int(slots) * world, int(k), int(m))
        if out is None and copy:
            out = np.empty(shape, np.float32)
        if out is not None:
            if out.dtype != np.float32 or not out.flags.c_contiguous or out.size != shape[0] * shape[1] * shape[2]:
                raise ValueError("out must be a C-contiguous float32 array of %d elements" % (shape[0] * shape[1] * shape[2]))
            self._ok(self._L.plsa_comm_allgather_stack_to(self._h, int(slots), int(m), int(k), out.reshape(-1)))
            return out.reshape(shape)
        p = C.POINTER(C.c_float)()
        self._ok(self._L.plsa_comm_allgather_stack(self._h, int(slots), int(m), int(k), C.byref(p)))
        return np.ctypeslib.as_array(p, shape=shape)

    def comm_allgather_host(self, a):
        a = np.ascontiguousarray(a)
        _, world = self.comm_info()
        out = np.empty((world,) + a.shape, a.dtype)
        self._ok(self._L.plsa_comm_allgather_host(self._h, a.ctypes.data, a.nbytes, out.ctypes.data))
        return out

    def comm_allreduce_f64(self, values, op="sum"):
        a = np.atleast_1d(np.array(values, dtype=np.float64))
        self._ok(self._L.plsa_comm_allreduce_f64(self._h, a, a.size, {"sum": 0, "max": 1}[op]))
        return a

    def comm_broadcast_host(self, a, root=0):
        a = np.array(a, copy=True, order="C")
        self._ok(self._L.plsa_comm_broadcast_host(self._h, a.ctypes.data, a.nbytes, int(root)))
        return a

    def allreduce_accumulator(self):
        self._ok(self._L.plsa_allreduce_accumulator(self._h))

    def all_pairs_hellinger(self, topics):
        """[t, t] float64 Hellinger distances between the rows of `topics` [t, m] (enstop_.py:258-266)."""
        T = _f32(topics)
        if T.ndim != 2:
            raise ValueError("topics must be a 2-D array")
        t, m = T.shape
        D = np.empty((t, t), np.float64)
        self._ok(self._L.plsa_all_pairs_hellinger(self._h, ptr(T), t, m, D))
        return D

    def all_pairs_kl(self, topics):
        """[t, t] float64 KL(topic i || topic j) in bits (enstop_.py:234-253)."""
        T = _f32(topics)
        if T.ndim != 2:
            raise ValueError("topics must be a 2-D array")
        t, m = T.shape
        D = np.empty((t, t), np.float64)
        self._ok(self._L.plsa_all_pairs_kl(self._h, ptr(T), t, m, D))
        return D

    def cluster_representatives(self, topics, labels, weights=None):
        """Stable topic of every cluster (enstop_.py:299-308, 385-393): [n_clusters, m] float32."""
        T = _f32(topics)
        lab = np.ascontiguousarray(labels, dtype=np.int32)
        t, m = T.shape
        if lab.shape != (t,):
            raise ValueError("labels must have one entry per topic")
        n_clusters = int(lab.max()) + 1 if lab.size and lab.max() >= 0 else 0
        out = np.empty((n_clusters, m), np.float32)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        self._ok(self._L.plsa_cluster_representatives(self._h, ptr(T), t, m, lab, ptr(w), n_clusters, out))
        return out

    def reference_chain_info(self):
        """norm_pwz of the reference arithmetic (plsa_reference_chain_info): chunks of the parity-pair walk that took the slow way,
        chunks walked, and whether this context is on the serial chain for the current corpus."""
        a, b, c_ = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        self._ok(self._L.plsa_reference_chain_info(self._h, C.byref(a), C.byref(b), C.byref(c_)))
        return dict(slow_chunks=a.value, chunks=b.value, serial_chain_now=bool(c_.value))

    def placement_info(self):
        n, a, b = C.c_int32(0), C.c_double(0.0), C.c_double(0.0)
        self._ok(self._L.plsa_placement_info(self._h, C.byref(n), C.byref(a), C.byref(b)))
        return dict(candidates=n.value, kept_fill_GBps=round(a.value, 1), worst_fill_GBps=round(b.value, 1))

    def balance_info(self):
        """Column-pass schedule of the current structure: measured chunk boundaries per XCD, per-XCD finish times of
        the last timed launch, timed launches spent, item length, number of items."""
        lo = (C.c_int32 * 9)(); t = (C.c_double * 8)()
        n, seg, items = C.c_int32(0), C.c_int32(0), C.c_int64(0)
        self._ok(self._L.plsa_schedule_info(self._h, lo, t, C.byref(n), C.byref(seg), C.byref(items)))
        return dict(xcd_chunk_boundaries=list(lo), xcd_finish_us=[round(v, 1) for v in t], timed_launches=n.value,
                    item_entries=seg.value, n_items=items.value)

    def release_scratch(self):
        """Free the materialised P and other large scratch buffers (re-created on demand)."""
        self._ok(self._L.plsa_release_scratch(self._h))

    # -- measurement ---------------------------------------------------------------------------------
    def timing(self, on=True):
        self._ok(self._L.plsa_timing_enable(self._h, int(on)))

    def timing_reset(self):
        self._ok(self._L.plsa_timing_reset(self._h))

    def timing_get(self, prefix):
        ms, n = C.c_double(0.0), C.c_int64(0)
        self._ok(self._L.plsa_timing_get(self._h, prefix.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def stream_bandwidth(self, nbytes=1 << 32, kind=0, reps=5):
        """GB/s of a fill (kind 0 non-temporal, 1 plain) or copy (2) over `nbytes` of scratch HBM."""
        out = C.c_double(0.0)
        self._ok(self._L.plsa_measure_stream_bandwidth(self._h, int(nbytes), int(kind), int(reps), C.byref(out)))
        return out.value

    def timing_report(self):
        buf = C.create_string_buffer(8192)
        self._ok(self._L.plsa_timing_report(self._h, buf, 8192))
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.rsplit(" ", 2)
            out[name] = (int(cnt), float(ms))
        return out


_engines = {}


def get_engine(device=None):
    """Process-wide Engine per device (buffers are reused across fits)."""
    dev = default_device() if device is None else int(device)
    eng = _engines.get(dev)
    if eng is None:
        eng = _engines[dev] = Engine(dev)
    return eng


_member_engines = {}


def get_member_engines(device=None, jobs=1):
    """`jobs` engines on one device for ensemble members fitted concurrently (each context has its own
    HIP streams and buffers): the process-wide engine of the device first, then cached extra ones."""
    first = get_engine(device)
    extra = _member_engines.setdefault(first.device, [])
    while len(extra) < jobs - 1:
        extra.append(Engine(first.device))
    return [first] + extra[:max(jobs - 1, 0)]


def reset_engines():
    """Close the cached per-device engines (tools / tests that change PLSA_* knobs, which a context
    reads when it is created)."""
    for eng in list(_engines.values()):
        eng.close()
    _engines.clear()
    for lst in _member_engines.values():
        for eng in lst:
            eng.close()
    _member_engines.clear()


def host_normalize_rows(a):
    """enstop/utils.py:8-41 normalize(a, axis=1), in place on a C-contiguous float64 array."""
    L = _lib.load()
    assert a.dtype == np.float64 and a.flags.c_contiguous and a.ndim == 2
    L.plsa_host_normalize_rows(a, a.shape[0], a.shape[1])
    return a
