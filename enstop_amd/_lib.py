"""ctypes binding of libplsa_hip.so (the C ABI of include/plsa_hip.h).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is visible, every entry
point raises.  The library is built in-tree by ``python -m enstop_amd.build`` (or
``__graft_entry__.build()``).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ENSTOP_AMD_LIB", os.path.join(_HERE, "libplsa_hip.so"))   # override: A/B builds

PLSA_FUSED = 1
PLSA_TRACE_LL = 4
PLSA_SW_LL_ONLY = 8
PLSA_STOP_NO_ZERO_ARM = 16
PLSA_SHARDED = 64
PLSA_REFERENCE_SUMS = 256   # the reference's float32 sums, rounding for rounding (include/plsa_hip.h)
PLSA_REFERENCE_LL = 512     # + the log-likelihood as one sequential float32 sum (plsa.py:322 read literally)

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_ctx = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_vp = C.c_void_p          # nullable array arguments are passed as raw addresses

# name -> (restype, argtypes); must list every symbol include/plsa_hip.h declares
SIGNATURES = {
    "plsa_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "plsa_hw_queues": (C.c_int, []),
    "plsa_create": (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    "plsa_destroy": (None, [_ctx]),
    "plsa_last_error": (C.c_char_p, [_ctx]),
    "plsa_synchronize": (C.c_int, [_ctx]),
    "plsa_device_info": (C.c_int, [_ctx, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(_i64)]),
    "plsa_upload_csr": (C.c_int, [_ctx, _i32p, _i32p, _f32p, _i64, _i64, _i64]),
    "plsa_bootstrap": (C.c_int, [_ctx, _vp, _i64]),
    "plsa_active_shape": (C.c_int, [_ctx, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "plsa_download_active_csr": (C.c_int, [_ctx, _i32p, _i32p, _f32p]),
    "plsa_set_factors": (C.c_int, [_ctx, _vp, _vp, _i64, _i64, _i32]),
    "plsa_get_factors": (C.c_int, [_ctx, _vp, _vp]),
    "plsa_init_factors_device": (C.c_int, [_ctx, _i32, C.c_uint64]),
    "plsa_init_factors_mt19937": (C.c_int, [_ctx, _i32, np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")]),
    "plsa_mt_marginals": (C.c_int, [_ctx, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), _i32]),
    "plsa_refit_init_mt19937": (C.c_int, [_ctx, C.c_void_p, _i64, _i32,
                                          np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")]),
    "plsa_copy_components_to_device": (C.c_int, [_ctx, _vp]),
    "plsa_set_arithmetic": (C.c_int, [_ctx, _i32]),
    "plsa_e_step": (C.c_int, [_ctx, C.c_float, _vp]),
    "plsa_set_p": (C.c_int, [_ctx, _f32p]),
    "plsa_m_step": (C.c_int, [_ctx, _vp, _i32, _vp, _vp]),
    "plsa_log_likelihood": (C.c_int, [_ctx, _vp, C.POINTER(C.c_double)]),
    "plsa_fit": (C.c_int, [_ctx, _vp, _i32, _i32, C.c_double, C.c_float, _i32, C.POINTER(_i32), _vp,
                           C.POINTER(_i32)]),
    "plsa_refit": (C.c_int, [_ctx, _vp, _i32, _i32, C.c_double, C.c_float, _i32, C.POINTER(_i32), _vp,
                             C.POINTER(_i32)]),
    "plsa_em_accumulate": (C.c_int, [_ctx, _vp, C.c_float, _vp]),
    "plsa_em_accumulate_materialised": (C.c_int, [_ctx, _vp, C.c_float, _vp]),
    "plsa_p_reserve": (C.c_int, [_ctx, C.c_int64, C.POINTER(C.c_void_p)]),
    "plsa_p_borrow": (C.c_int, [_ctx, C.c_void_p, C.c_int64]),
    "plsa_em_finish": (C.c_int, [_ctx]),
    "plsa_set_sample_weight": (C.c_int, [_ctx, _vp]),
    "plsa_comm_last_error": (C.c_int, [_ctx, C.c_char_p, _i64]),
    "plsa_accumulator_device": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(_i64)]),
    "plsa_accumulator_get": (C.c_int, [_ctx, _f32p]),
    "plsa_accumulator_set": (C.c_int, [_ctx, _f32p]),
    "plsa_comm_unique_id": (C.c_int, [C.c_void_p]),
    "plsa_comm_init": (C.c_int, [_ctx, C.c_char_p, _i32, _i32]),
    "plsa_comm_destroy": (C.c_int, [_ctx]),
    "plsa_comm_info": (C.c_int, [_ctx, C.POINTER(_i32), C.POINTER(_i32)]),
    "plsa_comm_barrier": (C.c_int, [_ctx]),
    "plsa_stack_reserve": (C.c_int, [_ctx, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]),
    "plsa_comm_allgather_stack": (C.c_int, [_ctx, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.POINTER(C.c_float))]),
    "plsa_comm_allgather_stack_to": (C.c_int, [_ctx, _i64, _i64, _i32, _f32p]),
    "plsa_comm_allgather_host": (C.c_int, [_ctx, _vp, _i64, _vp]),
    "plsa_comm_allreduce_f64": (C.c_int, [_ctx, _f64p, _i64, _i32]),
    "plsa_comm_broadcast_host": (C.c_int, [_ctx, _vp, _i64, _i32]),
    "plsa_allreduce_accumulator": (C.c_int, [_ctx]),
    "plsa_reference_chain_info": (C.c_int, [_ctx, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)]),
    "plsa_placement_info": (C.c_int, [_ctx, C.POINTER(_i32), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "plsa_schedule_info": (C.c_int, [_ctx, C.POINTER(_i32), C.POINTER(C.c_double), C.POINTER(_i32), C.POINTER(_i32),
                                     C.POINTER(C.c_int64)]),
    "plsa_release_scratch": (C.c_int, [_ctx]),
    "plsa_timing_enable": (C.c_int, [_ctx, _i32]),
    "plsa_timing_reset": (C.c_int, [_ctx]),
    "plsa_timing_get": (C.c_int, [_ctx, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "plsa_timing_report": (C.c_int, [_ctx, C.c_char_p, _i64]),
    "plsa_measure_stream_bandwidth": (C.c_int, [_ctx, _i64, _i32, _i32, C.POINTER(C.c_double)]),
    "plsa_all_pairs_hellinger": (C.c_int, [_ctx, C.c_void_p, _i64, _i64, _f64p]),
    "plsa_all_pairs_kl": (C.c_int, [_ctx, C.c_void_p, _i64, _i64, _f64p]),
    "plsa_cluster_representatives": (C.c_int, [_ctx, C.c_void_p, _i64, _i64, _i32p, _vp, _i32, _f32p]),
    "plsa_host_normalize_rows": (None, [_f64p, _i64, _i64]),
    "plsa_host_mt19937_jump": (C.c_int, [C.POINTER(C.c_uint32), _i32]),
    "plsa_generate_synthetic": (C.c_int, [_ctx, _i64, _i64, _i64, C.c_double, C.c_uint64,
                                          C.POINTER(_i64)]),
    "plsa_generate_synthetic_topics": (C.c_int, [_ctx, _i64, _i64, _i64, C.c_double, C.c_uint64, _i32, C.c_double,
                                                 C.c_double, C.POINTER(_i64)]),
    "plsa_synthetic_dominant_topics": (C.c_int, [_ctx, _i32p]),
}

_lib = None
HW_QUEUES = {"set_by": None, "hip_mapped_before_load": None}


def _default_hw_queues():
    """Ensemble members of a small corpus are fitted on up to four contexts of one GPU (enstop_.py's thread pool,
    enstop_.py:209-217); with the HIP runtime's default of 4 hardware queues their streams share queues and wait for each
    other (20NG shape: +10 % fits/min with 8).  The runtime reads GPU_MAX_HW_QUEUES at its first call, so it is set
    HERE, visibly, before the library is loaded -- never by the library itself -- unless the user set it or opted out
    with ENSTOP_AMD_HW_QUEUES=0 (any other value of that variable is the number to use).  It cannot take effect when the
    process initialised HIP earlier (e.g. torch.cuda): `hw_queues()` says whether the runtime was already mapped."""
    want = os.environ.get("ENSTOP_AMD_HW_QUEUES", "8").strip()
    if "GPU_MAX_HW_QUEUES" in os.environ:
        HW_QUEUES["set_by"] = "user"
        return
    try:
        n_queues = int(want)
    except ValueError:
        raise ValueError("ENSTOP_AMD_HW_QUEUES=%r: expected a positive integer (hardware queues to ask the HIP runtime for) "
                         "or 0 to leave the runtime's default" % (want,)) from None
    if n_queues <= 0:                     # 0 (documented) or a negative number: opt out -- nothing is exported, nothing probed
        HW_QUEUES["set_by"] = "opt-out"
        return
    try:
        with open("/proc/self/maps") as f:
            HW_QUEUES["hip_mapped_before_load"] = "libamdhip64" in f.read()
    except OSError:
        pass
    os.environ["GPU_MAX_HW_QUEUES"] = str(n_queues)
    HW_QUEUES["set_by"] = "enstop_amd"


def hw_queues():
    """{'value': hardware queues this process asks the HIP runtime for, 'set_by': 'enstop_amd' | 'user' | 'opt-out',
    'hip_mapped_before_load': True when libamdhip64 was already in the process (the setting may have come too late)}"""
    return dict(HW_QUEUES, value=int(load().plsa_hw_queues()))


def load():
    """Load libplsa_hip.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "enstop_amd: %s not found -- build the HIP extension first "
            "(python -m enstop_amd.build). There is no CPU fallback." % LIB_PATH)
    _default_hw_queues()
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(a):
    """Address of a C-contiguous ndarray, or None."""
    return None if a is None else a.ctypes.data
