#!/usr/bin/env python3
"""BASELINE configs[4] as north_star words it -- "5M docs x 200k vocab, 500M nnz, k=128, doc-blocks tiled in HBM" -- for the
MATERIALISED schedule: P(z|w,d) is 256 GB untiled; in 8 doc blocks that share one buffer it is 32 GB.  Runs 2 iterations tiled,
the same 2 iterations fused (which never stores P), compares, and reports the HBM high-water mark of the tiled run."""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                                   # noqa: E402  (mem_get_info only)
import bench                                                   # noqa: E402
import enstop_amd                                              # noqa: E402
from enstop_amd.engine import Engine, PLSA_FUSED               # noqa: E402
from enstop_amd.sharded import sharded_plsa_fit                # noqa: E402

cfg = dict(bench.CONFIGS[5])
if len(sys.argv) > 1:                                          # scale factor for a dry run
    f = float(sys.argv[1])
    cfg.update(n=int(cfg["n"] * f), nnz=int(cfg["nnz"] * f))
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.time()
with Engine(0) as eng:
    nnz = eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], seed=0)
    X = eng.download_active_csr()
print("corpus on the host: %d x %d, %d nnz, %.0f s" % (X.shape[0], X.shape[1], X.nnz, time.time() - t0), flush=True)
total = torch.cuda.mem_get_info(0)[1]
low = [total]
stop = threading.Event()


def watch():
    while not stop.is_set():
        low[0] = min(low[0], torch.cuda.mem_get_info(0)[0])
        time.sleep(0.05)


kw = dict(n_iter=2, n_iter_per_test=1, tolerance=0.0, e_step_thresh=1e-32, random_state=11)
th = threading.Thread(target=watch, daemon=True)
th.start()
t0 = time.time()
U, V, info = sharded_plsa_fit(X, cfg["k"], local_shards=blocks, flags=0, return_info=True, **kw)
t_tiled = time.time() - t0
stop.set(); th.join()
used_gb = (total - low[0]) / 1e9
print("tiled materialised: %d blocks, 2 iterations, %.0f s wall (host init and uploads included), HBM high-water %.1f GB"
      % (blocks, t_tiled, used_gb), flush=True)
t0 = time.time()
U2, V2, info2 = enstop_amd.plsa_fit(X, cfg["k"], np.ones(X.shape[0], np.float32), flags=PLSA_FUSED, return_info=True, **kw)
t_fused = time.time() - t0


def peak_rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())


out = {"shape": list(X.shape), "nnz": int(X.nnz), "k": cfg["k"], "blocks": blocks, "iterations": [int(info["n_iter"]), int(info2["n_iter"])],
       "p_untiled_GB": round(4.0 * (X.nnz + 64) * cfg["k"] / 1e9, 1), "hbm_high_water_GB": round(used_gb, 1),
       "tiled_vs_fused": {"U": peak_rel(U, U2), "V": peak_rel(V, V2),
                          "ll_rel": float(np.max(np.abs(info["log_likelihood_trace"][:3].astype(np.float64) - info2["log_likelihood_trace"][:3])
                                                 / np.abs(info2["log_likelihood_trace"][:3])))},
       "wall_s": {"tiled": round(t_tiled, 1), "fused": round(t_fused, 1)}}
print(json.dumps(out))
os.makedirs("gpurun_out/s3", exist_ok=True)
with open("gpurun_out/s3/cfg5_tiled_materialised.json", "w") as f:
    json.dump(out, f, indent=1)
