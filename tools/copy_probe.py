import sys, time, json, os
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from enstop_amd.engine import Engine
cfg = bench.CONFIGS[3]
eng = Engine(0)
eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], seed=0)
X = eng.download_active_csr()
eng.init_factors_numpy_stream(64, np.random.RandomState(1))
out = {}
for _ in range(2):
    t0 = time.perf_counter(); eng.upload_csr(X); eng.synchronize(); up = time.perf_counter() - t0
    eng.init_factors_numpy_stream(64, np.random.RandomState(1))
    t0 = time.perf_counter(); U, V = eng.get_factors(); down = time.perf_counter() - t0
    t0 = time.perf_counter(); U2, _ = eng.get_factors(want_v=False); down_u = time.perf_counter() - t0
print(json.dumps({"threads": os.environ.get("PLSA_STAGE_THREADS", "4"), "upload_ms": round(up * 1e3, 2), "download_ms": round(down * 1e3, 2), "download_U_only_ms": round(down_u * 1e3, 2)}))
# device-to-host into an array whose pages are already resident (what a pre-faulted result array would give)
from enstop_amd._lib import ptr
n, m, _ = eng.shape
Upre = np.zeros((n, 64), np.float32)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); eng._ok(eng._L.plsa_get_factors(eng._h, ptr(Upre), None)); ts.append(time.perf_counter() - t0)
print(json.dumps({"threads": os.environ.get("PLSA_STAGE_THREADS", "4"), "download_U_into_resident_pages_ms": [round(t * 1e3, 2) for t in ts]}))
