"""One process, one engine, M members back to back (20NG shape): run P copies side by side to see whether members of
SEPARATE PROCESSES overlap better on one GPU than members of separate threads (HIP runtime locks, queues)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from enstop_amd.engine import Engine, PLSA_FUSED
N, M, NNZ, K = 18_846, 173_762, 2_950_000, 20
members = int(sys.argv[1]) if len(sys.argv) > 1 else 24
start_at = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
eng = Engine(0)
eng.generate_synthetic(N, M, NNZ, seed=0)
def member(seed):
    rng = np.random.RandomState(seed)
    eng.bootstrap(rng.randint(0, N, size=N)); eng.init_factors_numpy_stream(K, rng)
    eng.fit(None, n_iter=50, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-16, flags=PLSA_FUSED)
member(1); member(2)
while time.time() < start_at: time.sleep(0.001)       # all copies start their timed loop together
t0 = time.time()
for r in range(members): member(100 + r)
eng.synchronize()
t1 = time.time()
print(json.dumps({"pid": os.getpid(), "members": members, "t0": t0, "t1": t1, "ms_per_member": round((t1 - t0) / members * 1e3, 3)}), flush=True)
