"""How should the member stack reach the caller's host array?  (a) today: D2H into the engine's page-locked buffer, then
a NumPy copy into a fresh array; (b) hipMemcpy D2H straight into a fresh pageable array; (c) into an already touched one;
(d) view of the page-locked buffer (no host copy).  Sizes: 8 x (64 x 100k) = 205 MB (config 3 at 8 GPUs), 32 x (20 x 174k) = 445 MB."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from enstop_amd.engine import Engine

hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
with Engine() as eng:
    for slots, k, m in ((8, 64, 100_000), (32, 20, 173_762)):
        base = eng.stack_reserve(slots, k, m)
        nbytes = slots * k * m * 4
        eng.comm_allgather_stack(slots, k, m, copy=False)              # first call: allocates the page-locked buffer
        def best(f, reps=4):
            ts = []
            for _ in range(reps):
                t = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t); del r
            return round(min(ts) * 1e3, 2), round(max(ts) * 1e3, 2)
        a = best(lambda: eng.comm_allgather_stack(slots, k, m, copy=True))
        d = best(lambda: eng.comm_allgather_stack(slots, k, m, copy=False))
        def direct_fresh():
            out = np.empty((slots, k, m), np.float32)
            hip.hipMemcpy(out.ctypes.data, base, nbytes, 2)
            return out
        b = best(direct_fresh)
        out = np.zeros((slots, k, m), np.float32)
        c = best(lambda: hip.hipMemcpy(out.ctypes.data, base, nbytes, 2))
        print(json.dumps({"stack_MB": round(nbytes / 2**20, 1), "ms_min_max": {"a_pinned_then_numpy_copy_fresh": a, "b_direct_into_fresh_pageable": b,
                                                                              "c_direct_into_touched_pageable": c, "d_view_of_pinned_buffer": d}}), flush=True)
