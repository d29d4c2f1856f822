"""Do ensemble members on SEPARATE engines (contexts, streams) of one GPU overlap?  T threads, each fits members on its own
engine (20NG shape, k = 20, 50 iterations, NumPy-identical device initialisation); throughput against T = 1."""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from enstop_amd.engine import Engine, PLSA_FUSED

N, M, NNZ, K = 18_846, 173_762, 2_950_000, 20
with Engine(0) as e0:
    e0.generate_synthetic(N, M, NNZ, seed=0)
    X = e0.download_active_csr()
engines = [Engine(0) for _ in range(4)]
for e in engines:
    e.upload_csr(X)

def member(e, seed, what):
    rng = np.random.RandomState(seed)
    if "b" in what: e.bootstrap(rng.randint(0, N, size=N))
    if "i" in what: e.init_factors_numpy_stream(K, rng)
    if "f" in what: e.fit(None, n_iter=50, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-16, flags=PLSA_FUSED)

for e in engines:
    member(e, 1, "bif"); member(e, 2, "bif")
for what in ("bif", "f", "i", "b"):
    for T in (1, 2, 4):
        per = 12
        def work(j):
            for r in range(per): member(engines[j], 100 + j * per + r, what)
        ths = [threading.Thread(target=work, args=(j,)) for j in range(T)]
        for e in engines[:T]: e.synchronize()
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
        print(json.dumps({"stages": what, "threads": T, "members": T * per, "ms_per_member": round(dt / (T * per) * 1e3, 3)}), flush=True)
