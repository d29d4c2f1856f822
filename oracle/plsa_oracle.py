"""ctypes binding of oracle/plsa_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
enstop_amd/ never does.  Parity status of the underlying C restatement: pinned against
tests/golden/*.npz (generated from the reference itself by tests/golden/make_golden.py).

Function names and argument order follow the reference's kernel-level functions
(enstop/plsa.py:39,124,221,329,517,746,820) so the tests read like calls to the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i64 = C.c_int64


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


class Oracle:
    def __init__(self, fast=False, variant=None):
        """variant: None / "strict" (the checker: the reference's float32 arithmetic), "fast" (the timed
        CPU baseline), "n64" (float64 norm_pwz + log-likelihood accumulator) or "wide" (all accumulators
        float64) -- the last two are diagnostics, not the reference's arithmetic."""
        variant = variant or ("fast" if fast else "strict")
        name = {"strict": "liboracle_plsa.so", "fast": "liboracle_plsa_fast.so",
                "n64": "liboracle_plsa_n64.so", "wide": "liboracle_plsa_wide.so"}[variant]
        self.variant = variant
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        L = self.lib = C.CDLL(path)
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_set_ll_sequential.argtypes = [C.c_int]
        L.oracle_max_threads.restype = C.c_int
        L.oracle_e_step.argtypes = [_i32p, _i32p, _i64, _f32p, _f32p, _f32p, _i64, _i64, C.c_float]
        L.oracle_m_step.argtypes = [_i32p, _i32p, _f32p, _i64, _f32p, _f32p, _f32p, _f32p, _f32p,
                                    _i64, _i64, _i64]
        L.oracle_m_step_w.argtypes = [_i32p, _i32p, _f32p, _i64, _f32p, _f32p, _f32p, _f32p, _f32p,
                                      _f32p, _i64, _i64, _i64]
        L.oracle_refit_m_step.argtypes = [_i32p, _i32p, _f32p, _i64, _f32p, _f32p, _f32p, _i64, _i64]
        L.oracle_log_likelihood.argtypes = [_i32p, _i32p, _f32p, _i64, _f32p, _f32p, _f32p, _i64, _i64]
        L.oracle_log_likelihood.restype = C.c_float
        L.oracle_fit_inner.argtypes = [_i32p, _i32p, _f32p, _i64, _f32p, _f32p, _f32p, _i64, _i64,
                                       _i64, C.c_int32, C.c_int32, C.c_double, C.c_float, C.c_int32,
                                       C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.oracle_fit_inner.restype = C.c_int
        L.oracle_refit_inner.argtypes = [_i32p, _i32p, _f32p, _i64, _f32p, _f32p, _f32p, _i64, _i64,
                                         _i64, C.c_int32, C.c_int32, C.c_double, C.c_float,
                                         C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.oracle_refit_inner.restype = C.c_int
        L.oracle_normalize_rows.argtypes = [_f64p, _i64, _i64]
        L.oracle_streamed_fit_inner.argtypes = [_i32p, _i32p, _f32p, _i64, _f32p, _f32p, _f32p, _i64, _i64, _i64,
                                                _i64, C.c_int32, C.c_int32, C.c_double, C.c_float, C.c_int32,
                                                C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.oracle_streamed_fit_inner.restype = C.c_int
        L.oracle_streamed_refit_inner.argtypes = [_i32p, _i32p, _f32p, _i64, _f32p, _f32p, _f32p, _i64, _i64, _i64,
                                                  _i64, C.c_int32, C.c_int32, C.c_double, C.c_float,
                                                  C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.oracle_streamed_refit_inner.restype = C.c_int

    # -- threading ------------------------------------------------------------------------
    def set_threads(self, t):
        self.lib.oracle_set_threads(int(t))

    def max_threads(self):
        return int(self.lib.oracle_max_threads())

    def set_ll_sequential(self, on=True):
        """log-likelihood reduction on one thread (= set_threads(1) for the result, bit for bit) while the E-step keeps its threads"""
        self.lib.oracle_set_ll_sequential(int(bool(on)))

    # -- kernel-level (numba signatures of plsa.py:26,111,208,314,734) ---------------------
    def plsa_e_step(self, X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, p_z_given_wd,
                    probability_threshold=1e-32):
        k, m = p_w_given_z.shape
        self.lib.oracle_e_step(X_rows, X_cols, X_vals.shape[0], p_w_given_z, p_z_given_d,
                               p_z_given_wd, m, k, np.float32(probability_threshold))
        return p_z_given_wd

    def plsa_m_step(self, X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, p_z_given_wd,
                    norm_pwz, norm_pdz):
        k, m = p_w_given_z.shape
        n = p_z_given_d.shape[0]
        self.lib.oracle_m_step(X_rows, X_cols, X_vals, X_vals.shape[0], p_w_given_z, p_z_given_d,
                               p_z_given_wd, norm_pwz, norm_pdz, n, m, k)
        return p_w_given_z, p_z_given_d

    def plsa_m_step_w_sample_weight(self, X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d,
                                    p_z_given_wd, sample_weight, norm_pwz, norm_pdz):
        k, m = p_w_given_z.shape
        n = p_z_given_d.shape[0]
        self.lib.oracle_m_step_w(X_rows, X_cols, X_vals, X_vals.shape[0], p_w_given_z,
                                 p_z_given_d, p_z_given_wd, sample_weight, norm_pwz, norm_pdz,
                                 n, m, k)
        return p_w_given_z, p_z_given_d

    def plsa_refit_m_step(self, X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, p_z_given_wd,
                          sample_weight, norm_pdz):
        n, k = p_z_given_d.shape
        self.lib.oracle_refit_m_step(X_rows, X_cols, X_vals, X_vals.shape[0], p_z_given_d,
                                     p_z_given_wd, norm_pdz, n, k)
        return p_w_given_z, p_z_given_d

    def log_likelihood(self, X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, sample_weight):
        k, m = p_w_given_z.shape
        return np.float32(self.lib.oracle_log_likelihood(X_rows, X_cols, X_vals, X_vals.shape[0],
                                                         p_w_given_z, p_z_given_d, sample_weight,
                                                         m, k))

    # -- loop-level ------------------------------------------------------------------------
    def plsa_fit_inner(self, X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, sample_weight,
                       n_iter=100, n_iter_per_test=10, tolerance=0.001, e_step_thresh=1e-32,
                       use_sample_weights=False, return_trace=False):
        k, m = p_w_given_z.shape
        n = p_z_given_d.shape[0]
        trace = np.zeros(n_iter + 2, np.float32)
        nll, iters = C.c_int32(0), C.c_int32(0)
        rc = self.lib.oracle_fit_inner(X_rows, X_cols, X_vals, X_vals.shape[0], p_w_given_z,
                                       p_z_given_d, sample_weight, n, m, k, n_iter, n_iter_per_test,
                                       float(tolerance), np.float32(e_step_thresh),
                                       int(bool(use_sample_weights)), trace.ctypes.data,
                                       C.byref(nll), C.byref(iters))
        if rc:
            raise MemoryError("oracle_fit_inner")
        if return_trace:
            return p_z_given_d, p_w_given_z, trace[: nll.value].copy(), iters.value
        return p_z_given_d, p_w_given_z

    def plsa_refit_inner(self, X_rows, X_cols, X_vals, topics, p_z_given_d, sample_weight,
                         n_iter=50, n_iter_per_test=10, tolerance=0.005, e_step_thresh=1e-32,
                         return_trace=False):
        k, m = topics.shape
        n = p_z_given_d.shape[0]
        trace = np.zeros(n_iter + 2, np.float32)
        nll, iters = C.c_int32(0), C.c_int32(0)
        rc = self.lib.oracle_refit_inner(X_rows, X_cols, X_vals, X_vals.shape[0], topics,
                                         p_z_given_d, sample_weight, n, m, k, n_iter,
                                         n_iter_per_test, float(tolerance),
                                         np.float32(e_step_thresh), trace.ctypes.data,
                                         C.byref(nll), C.byref(iters))
        if rc:
            raise MemoryError("oracle_refit_inner")
        if return_trace:
            return p_z_given_d, trace[: nll.value].copy(), iters.value
        return p_z_given_d

    def normalize(self, ndarray, axis=1):
        assert axis == 1 and ndarray.dtype == np.float64 and ndarray.flags.c_contiguous
        self.lib.oracle_normalize_rows(ndarray, ndarray.shape[0], ndarray.shape[1])

    # -- driver-level restatement of plsa_fit / plsa_refit (plsa.py:707-730, 975-997) ---------
    def plsa_fit(self, X, k, sample_weight, init="random", n_iter=100, n_iter_per_test=10,
                 tolerance=0.001, e_step_thresh=1e-32, random_state=None, return_trace=False):
        from sklearn.utils import check_random_state
        rng = check_random_state(random_state)
        n, m = X.shape
        if isinstance(init, str) and init == "random":
            V = rng.rand(k, m)                                   # plsa.py:455 (this order)
            U = rng.rand(n, k)                                   # plsa.py:456
        elif isinstance(init, (tuple, list)):
            U, V = (np.array(a, dtype=np.float64, order="C") for a in init)
        else:
            raise ValueError("Unrecognized init {}".format(init))
        self.normalize(V, axis=1)                                # plsa.py:510
        self.normalize(U, axis=1)                                # plsa.py:511
        U = U.astype(np.float32, order="C")
        V = V.astype(np.float32, order="C")
        use_sw = bool(np.any(sample_weight != 1.0))              # plsa.py:712
        A = X.tocoo().astype(np.float32)                         # plsa.py:714
        return self.plsa_fit_inner(np.ascontiguousarray(A.row, np.int32),
                                   np.ascontiguousarray(A.col, np.int32),
                                   np.ascontiguousarray(A.data, np.float32), V, U,
                                   np.ascontiguousarray(sample_weight, np.float32),
                                   n_iter, n_iter_per_test, tolerance, e_step_thresh, use_sw,
                                   return_trace=return_trace)

    def plsa_refit(self, X, topics, sample_weight, n_iter=50, n_iter_per_test=10, tolerance=0.005,
                   e_step_thresh=1e-32, random_state=None, return_trace=False):
        from sklearn.utils import check_random_state
        A = X.tocoo().astype(np.float32)
        k = topics.shape[0]
        rng = check_random_state(random_state)
        U = rng.rand(A.shape[0], k)                              # plsa.py:979
        self.normalize(U, axis=1)
        U = U.astype(np.float32)
        topics = np.ascontiguousarray(topics, np.float32)
        return self.plsa_refit_inner(np.ascontiguousarray(A.row, np.int32),
                                     np.ascontiguousarray(A.col, np.int32),
                                     np.ascontiguousarray(A.data, np.float32), topics, U,
                                     np.ascontiguousarray(sample_weight, np.float32),
                                     n_iter, n_iter_per_test, tolerance, e_step_thresh,
                                     return_trace=return_trace)

    # -- enstop/streamed_plsa.py:469-603 plsa_fit_inner_blockwise on caller-held factors (V, U mutated in place) ---
    def streamed_plsa_fit_inner(self, X_rows, X_cols, X_vals, p_w_given_z, p_z_given_d, sample_weight,
                                block_size=65536, n_iter=100, n_iter_per_test=10, tolerance=0.001,
                                e_step_thresh=1e-32, use_sample_weights=False, return_trace=False):
        k, m = p_w_given_z.shape
        n = p_z_given_d.shape[0]
        trace = np.zeros(n_iter + 2, np.float32)
        nll, iters = C.c_int32(0), C.c_int32(0)
        rc = self.lib.oracle_streamed_fit_inner(X_rows, X_cols, X_vals, X_vals.shape[0], p_w_given_z, p_z_given_d,
                                                sample_weight, n, m, k, int(block_size), n_iter, n_iter_per_test,
                                                float(tolerance), np.float32(e_step_thresh),
                                                int(bool(use_sample_weights)), trace.ctypes.data, C.byref(nll),
                                                C.byref(iters))
        if rc:
            raise MemoryError("oracle_streamed_fit_inner")
        if return_trace:
            return p_z_given_d, p_w_given_z, trace[: nll.value].copy(), iters.value
        return p_z_given_d, p_w_given_z

    # -- enstop/streamed_plsa.py: plsa_fit (:606-699) and plsa_refit (:959-1039) -----------------
    def streamed_plsa_fit(self, X, k, sample_weight, init="random", block_size=65536, n_iter=100,
                          n_iter_per_test=10, tolerance=0.001, e_step_thresh=1e-32, random_state=None,
                          return_trace=False):
        from sklearn.utils import check_random_state
        rng = check_random_state(random_state)
        n, m = X.shape
        assert isinstance(init, str) and init == "random"
        V = rng.rand(k, m)                                       # plsa_init, shared with plsa.py (:680)
        U = rng.rand(n, k)
        self.normalize(V, axis=1)
        self.normalize(U, axis=1)
        U = U.astype(np.float32, order="C")
        V = V.astype(np.float32, order="C")
        sw = np.ascontiguousarray(sample_weight, np.float32)
        A = X.tocoo().astype(np.float32)
        trace = np.zeros(n_iter + 2, np.float32)
        nll, iters = C.c_int32(0), C.c_int32(0)
        rc = self.lib.oracle_streamed_fit_inner(
            np.ascontiguousarray(A.row, np.int32), np.ascontiguousarray(A.col, np.int32),
            np.ascontiguousarray(A.data, np.float32), A.nnz, V, U, sw, n, m, k, int(block_size), n_iter,
            n_iter_per_test, float(tolerance), np.float32(e_step_thresh), int(bool(np.any(sw != 1.0))),
            trace.ctypes.data, C.byref(nll), C.byref(iters))
        if rc:
            raise MemoryError("oracle_streamed_fit_inner")
        if return_trace:
            return U, V, trace[: nll.value].copy(), iters.value
        return U, V

    def streamed_plsa_refit(self, X, topics, sample_weight, block_size=65536, n_iter=50, n_iter_per_test=10,
                            tolerance=0.005, e_step_thresh=1e-32, random_state=None, return_trace=False):
        from sklearn.utils import check_random_state
        A = X.tocoo().astype(np.float32)
        k = topics.shape[0]
        rng = check_random_state(random_state)
        U = rng.rand(A.shape[0], k)                              # streamed_plsa.py:1020-1022
        self.normalize(U, axis=1)
        U = U.astype(np.float32)
        topics = np.ascontiguousarray(topics, np.float32)
        trace = np.zeros(n_iter + 2, np.float32)
        nll, iters = C.c_int32(0), C.c_int32(0)
        rc = self.lib.oracle_streamed_refit_inner(
            np.ascontiguousarray(A.row, np.int32), np.ascontiguousarray(A.col, np.int32),
            np.ascontiguousarray(A.data, np.float32), A.nnz, topics, U,
            np.ascontiguousarray(sample_weight, np.float32), A.shape[0], A.shape[1], k, int(block_size),
            n_iter, n_iter_per_test, float(tolerance), np.float32(e_step_thresh), trace.ctypes.data,
            C.byref(nll), C.byref(iters))
        if rc:
            raise MemoryError("oracle_streamed_refit_inner")
        if return_trace:
            return U, trace[: nll.value].copy(), iters.value
        return U
