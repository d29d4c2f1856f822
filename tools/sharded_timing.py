#!/usr/bin/env python3
"""Per-iteration cost of the doc-sharded EM loop on ONE GPU with a one-rank RCCL communicator (the
collectives degenerate to device-side no-ops plus their launch cost): plsa_fit(..., PLSA_SHARDED) -- the
all-reduce enqueued underneath the document pass, four-kernel column tail -- and the three-call form
(plsa_em_accumulate / plsa_allreduce_accumulator / plsa_em_finish), against the single-GPU fused driver on
the same corpus.  Says what the sharded schedule costs before any xGMI traffic is added."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                             # noqa: E402
from enstop_amd import comm                                    # noqa: E402
from enstop_amd.engine import Engine, PLSA_FUSED, PLSA_SHARDED  # noqa: E402


def main():
    for name, (n, m, nnz, k) in {"cfg3": (1_000_000, 100_000, 100_000_000, 64), "cfg2": (100_000, 50_000, 10_000_000, 32)}.items():
        with Engine(0) as eng:
            c = comm.RcclComm(eng, 0, 1, comm.rendezvous_id(0, "/tmp/plsa_sharded_timing_%d.id" % os.getpid()))
            eng.generate_synthetic(n, m, nnz, seed=0)
            iters = 30
            res = {}
            for label, flags in (("single_driver", PLSA_FUSED), ("sharded_world1", PLSA_FUSED | PLSA_SHARDED)):
                eng.init_factors_numpy_stream(k, np.random.RandomState(42))
                eng.fit(None, n_iter=3, n_iter_per_test=10, tolerance=0.0, flags=flags)          # warm-up, structures
                eng.init_factors_numpy_stream(k, np.random.RandomState(42))
                eng.synchronize()
                t0 = time.perf_counter()
                eng.fit(None, n_iter=iters, n_iter_per_test=10, tolerance=0.0, flags=flags)
                eng.synchronize()
                res[label] = ((time.perf_counter() - t0) / iters, eng.get_factors(want_u=False)[1])
            eng.init_factors_numpy_stream(k, np.random.RandomState(42))
            eng.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                eng.em_accumulate(None); eng.allreduce_accumulator(); eng.em_finish()
            eng.synchronize()
            t3 = (time.perf_counter() - t0) / iters
            V3 = eng.get_factors(want_u=False)[1]
            V1 = res["single_driver"][1]
            print(json.dumps({"config": name, "ms_per_iter_single_driver": round(res["single_driver"][0] * 1e3, 3),
                              "ms_per_iter_PLSA_SHARDED_world1": round(res["sharded_world1"][0] * 1e3, 3),
                              "ms_per_iter_three_call_form_world1": round(t3 * 1e3, 3),
                              "max_rel_diff_topics_sharded": float(np.abs(V1 - res["sharded_world1"][1]).max() / V1.max()),
                              "max_rel_diff_topics_three_call": float(np.abs(V1 - V3).max() / V1.max())}), flush=True)
            c.close()


if __name__ == "__main__":
    main()
