mkdir -p gpurun_out/r3
timeout 1500 python tools/soak.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r3/soak.txt | tail -6
# bit-reproducibility of a mid-size fit across 20 fresh fits in one process (pipelined path, early stop, weights)
python - <<'PY' 2>&1 | tail -4 | tee -a gpurun_out/r3/soak.txt
import sys; sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp, enstop_amd
rs = np.random.RandomState(0)
X = sp.random(20000, 8000, density=0.01, format="csr", random_state=rs, dtype=np.float32); X.data = np.ceil(X.data * 4).astype(np.float32)
sw = (0.5 + rs.rand(20000)).astype(np.float32)
ref = None
for rep in range(20):
    U, V, info = enstop_amd.plsa_fit(X, 32, sw, n_iter=37, n_iter_per_test=3, tolerance=1e-6, random_state=3, return_info=True)
    if ref is None: ref = (U.copy(), V.copy(), info["log_likelihood_trace"].copy(), info["n_iter"])
    assert np.array_equal(U, ref[0]) and np.array_equal(V, ref[1]) and np.array_equal(info["log_likelihood_trace"], ref[2]) and info["n_iter"] == ref[3], rep
print("20 repeated weighted fits with frequent likelihood tests: bit-identical (n_iter %d)" % ref[3])
PY
