#!/usr/bin/env python3
"""Per-iteration cost of the doc-sharded EM loop (enstop_amd/sharded.py) on ONE GPU with the RCCL
all-reduce in place (world size 1: the collective degenerates to a device-side no-op plus its launch
and synchronisation cost), against the single-GPU fused driver on the same corpus.  Says how much
the split into accumulate / all-reduce / finish costs before any xGMI traffic is added."""
import json
import os
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                   # noqa: E402
import torch.distributed as dist                               # noqa: E402
import numpy as np                                             # noqa: E402
from enstop_amd.engine import Engine, PLSA_FUSED               # noqa: E402
from enstop_amd.sharded import TorchComm, sharded_em           # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    comm = TorchComm()
    out = []
    for name, (n, m, nnz, k) in {"cfg3": (1_000_000, 100_000, 100_000_000, 64), "cfg2": (100_000, 50_000, 10_000_000, 32)}.items():
        with Engine(0) as eng:
            eng.generate_synthetic(n, m, nnz, seed=0)
            iters = 30
            eng.init_factors_numpy_stream(k, np.random.RandomState(42))
            eng.fit(None, n_iter=3, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)          # warm-up, structures
            eng.init_factors_numpy_stream(k, np.random.RandomState(42))
            t0 = time.perf_counter()
            eng.fit(None, n_iter=iters, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
            t_single = (time.perf_counter() - t0) / iters
            _, V1 = eng.get_factors(want_u=False)
            eng.init_factors_numpy_stream(k, np.random.RandomState(42))
            sharded_em([eng], comm, None, n_iter=3, n_iter_per_test=10, tolerance=0.0)
            eng.init_factors_numpy_stream(k, np.random.RandomState(42))
            t0 = time.perf_counter()
            sharded_em([eng], comm, None, n_iter=iters, n_iter_per_test=10, tolerance=0.0)
            t_sharded = (time.perf_counter() - t0) / iters
            _, V2 = eng.get_factors(want_u=False)
            out.append({"config": name, "ms_per_iter_single_driver": round(t_single * 1e3, 3),
                        "ms_per_iter_sharded_loop_world1": round(t_sharded * 1e3, 3),
                        "max_abs_diff_topics": float(np.abs(V1 - V2).max())})
            print(json.dumps(out[-1]), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
