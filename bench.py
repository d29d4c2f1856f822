#!/usr/bin/env python3
"""bench.py -- EM iterations/sec of the pLSA hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--schedule fused|materialised]

A "step" is one EM iteration (E-step + M-step, plus the log-likelihood test at the reference's
schedule, plsa.py:630) over the whole synthetic corpus, driven through the C ABI (plsa_fit).
N = 1 : one fit on the corpus.   N > 1 : the ensemble path -- one process per GPU (torchrun), rank r
fits bootstrap member r of the same corpus (device-side row gather), no collective on the EM path,
one RCCL all-gather of the topic matrices at the end (inside the timed region).  `value` is the
whole-job aggregate: (N * K) EM iterations / max-over-ranks wall time.

Inputs are generated in HBM before the timed region (plsa_generate_synthetic); factors are
initialised on the host exactly as plsa_init does and uploaded before the timed region.

Extra objects in the JSON line:
  roofline      the per-nnz materialising E-step kernel (plsa.py:39-107; the kernel the north_star
                roofline target names): algorithmic bytes (SURVEY.md section 8d / DESIGN.md) / average
                launch duration, HIP events on the engine's stream over a second timed leg that runs
                the reference's own kernel sequence (E-step -> M-step -> LL) on the same data
  roofline_dominant_fused   same figures for the dominant kernel of the main (fused) timed region
  cpu_baseline  the CPU port (oracle/plsa_oracle.c, -O3 -ffast-math, OpenMP, reference thread
                structure) timed on a bounded row-sample of the same corpus, rank 0 / N = 1 only
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs (index = position in `configs`, 1-based like SURVEY.md section 8)
CONFIGS = {
    1: dict(n=18_846, m=173_762, nnz=2_950_000, k=20, name="20NG-shaped synthetic CSR 18846 x 173762, 2.95M nnz, k=20"),
    2: dict(n=100_000, m=50_000, nnz=10_000_000, k=32, name="synthetic CSR 100k docs x 50k vocab, 10M nnz, k=32"),
    3: dict(n=1_000_000, m=100_000, nnz=100_000_000, k=64, name="synthetic CSR 1M docs x 100k vocab, 100M nnz, k=64"),
    5: dict(n=5_000_000, m=200_000, nnz=500_000_000, k=128, name="synthetic CSR 5M docs x 200k vocab, 500M nnz, k=128"),
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
FITS_ITERS = 50                # EM iterations per ensemble member when quoting fits/min


def algorithmic_bytes(kind, n, m, nnz, k):
    """Compulsory HBM traffic per launch, each array counted once."""
    if kind == "e_step":        # indptr + indices + P write + U,V read
        return 4 * (n + 1) + 4 * nnz + 4 * k * nnz + 4 * k * (n + m)
    if kind == "m_step_p":      # indptr + indices + data + P read + U,V write
        return 4 * (n + 1) + 8 * nnz + 4 * k * nnz + 4 * k * (n + m)
    if kind == "loglik":
        return 4 * (n + 1) + 8 * nnz + 4 * k * (n + m)
    if kind == "fused":         # CSR once, factors read once and written once
        return 4 * (n + 1) + 8 * nnz + 8 * k * (n + m)
    if kind == "fused_col":     # CSC copy once, factors read once, V written once
        return 4 * (m + 1) + 8 * nnz + 4 * k * (n + m) + 4 * k * m
    raise KeyError(kind)


KERNEL_KIND = {
    "k_e_step": "e_step", "k_row_pass<P>": "m_step_p", "k_col_pass<P>": "m_step_p", "k_loglik": "loglik",
    "k_row_pass<fused>": "fused", "k_row_pass<fused,LL>": "fused", "k_col_pass<fused>": "fused_col",
}


def init_factors(n, m, k, seed):
    """plsa_init(random) + float32 casts, plsa.py:455-456, 510-511, 709-710."""
    from enstop_amd.plsa import plsa_init

    class S:
        shape = (n, m)
    U, V = plsa_init(S, k, init="random", rng=np.random.RandomState(seed))
    return U.astype(np.float32), V.astype(np.float32)


def cpu_baseline(eng, cfg, k, budget_cells=1.2e9, iters=3):
    """Reference-structure CPU port on the first rows of the same corpus, all host cores."""
    from oracle.plsa_oracle import Oracle
    o = Oracle(fast=True)
    cores = os.cpu_count() or 1
    o.set_threads(cores)
    A = eng.download_active_csr()
    n, m = A.shape
    nnz_full = A.nnz
    target_nnz = int(budget_cells / k)
    rows = int(min(n, max(1000, np.searchsorted(A.indptr, target_nnz))))
    S = A[:rows]
    Ac = S.tocoo()
    r, c, v = (np.ascontiguousarray(Ac.row, np.int32), np.ascontiguousarray(Ac.col, np.int32),
               np.ascontiguousarray(Ac.data, np.float32))
    U, V = init_factors(rows, m, k, 42)
    sw = np.ones(rows, np.float32)
    t0 = time.perf_counter()
    _, _, _, done = o.plsa_fit_inner(r, c, v, V, U, sw, n_iter=iters, n_iter_per_test=10, tolerance=0.0,
                                     e_step_thresh=1e-32, use_sample_weights=False, return_trace=True)
    dt = time.perf_counter() - t0
    frac = S.nnz / nnz_full
    return {
        "value": round(done / dt * frac, 5), "unit": "iter/s (full-corpus equivalent)", "cores": cores,
        "kind": "port",
        "sample": "first %d docs (%d nnz = %.3f of the corpus, full %d-word vocabulary), %d EM iterations in %.2f s; "
                  "iterations/s on the sample x nnz fraction" % (rows, S.nnz, frac, m, done, dt),
        "sample_iter_per_s": round(done / dt, 4),
        "gcell_per_s": round(S.nnz * k * done / dt / 1e9, 4),
    }


def quick_config(eng, cfg_id, steps, warmup, seed):
    """Compact measurement of another BASELINE.json config on the same device (N = 1 only): fused EM
    iterations/s and the materialising E-step's roofline fraction, same method as the main legs."""
    from enstop_amd.engine import PLSA_FUSED
    cfg = CONFIGS[cfg_id]
    n, m, k = cfg["n"], cfg["m"], cfg["k"]
    nnz = eng.generate_synthetic(n, m, cfg["nnz"], zipf_s=1.07, seed=seed)
    U0, V0 = init_factors(n, m, k, 42)
    eng.release_scratch()
    eng.set_factors(U0, V0)
    eng.fit(None, n_iter=warmup, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=PLSA_FUSED)
    eng.synchronize()
    t0 = time.perf_counter()
    it, _ = eng.fit(None, n_iter=steps, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=PLSA_FUSED)
    eng.synchronize()
    dt = time.perf_counter() - t0
    eng.timing(True)
    eng.e_step(1e-32, want_host_copy=False)
    eng.timing_reset()
    for _ in range(5):
        eng.e_step(1e-32, want_host_copy=False)
    ms, cnt = eng.timing_get("k_e_step")
    eng.timing(False)
    b = algorithmic_bytes("e_step", n, m, nnz, k)
    return {"workload": cfg["name"], "nnz": nnz, "k": k, "steps": it, "value": round(it / dt, 2), "unit": "iter/s",
            "ms_per_step": round(dt / it * 1e3, 4),
            "e_step": {"avg_launch_ms": round(ms / cnt, 5), "achieved_GBps": round(b / 1e9 / (ms / cnt / 1e3), 1),
                       "frac": round(b / 1e9 / (ms / cnt / 1e3) / HBM_PEAK_GBS, 4)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--schedule", default=os.environ.get("PLSA_BENCH_SCHEDULE", "fused"),
                    choices=["fused", "materialised"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    # Only the JSON line may reach stdout: route file descriptor 1 to stderr for the whole run (RCCL and
    # the HIP runtime print banners from C) and write the result to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    backend = None
    force_dist = os.environ.get("PLSA_BENCH_FORCE_DIST", "0") == "1"   # exercise the RCCL path with 1 rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # torch first: its HIP runtime / RCCL are then the ones libplsa_hip.so binds to
        import torch
        import torch.distributed as dist
        if local_rank >= torch.cuda.device_count():      # launcher restricted the visible devices
            local_rank = 0
        torch.cuda.set_device(local_rank)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe)                        # forces communicator creation now
            torch.cuda.synchronize()
            backend = "nccl"
        except Exception as e:                            # keep the measurement alive: control plane
            print("bench.py: RCCL init failed (%r); falling back to gloo for barrier/gather" % (e,),
                  file=sys.stderr)                        # over gloo, topics gathered through the host
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group("gloo", rank=rank, world_size=world)
            backend = "gloo"
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torchrun for N > 1" % (args.gpus, world),
              file=sys.stderr)

    from enstop_amd.engine import Engine, PLSA_FUSED
    cfg = CONFIGS[args.config]
    n, m, k = cfg["n"], cfg["m"], cfg["k"]
    flags = PLSA_FUSED if args.schedule == "fused" else 0

    eng = Engine(local_rank)
    info = eng.device_info()
    t_gen = time.perf_counter()
    nnz = eng.generate_synthetic(n, m, cfg["nnz"], zipf_s=1.07, seed=args.seed)
    if dist is not None:  # ensemble member `rank`: bootstrap rows on the device (enstop_.py:87-88)
        idx = np.random.RandomState(args.seed + 1000 + rank).randint(0, n, size=n)
        eng.bootstrap(idx)
    n_act, m_act, nnz_act = eng.shape
    t_gen = time.perf_counter() - t_gen
    U0, V0 = init_factors(n_act, m, k, 42 + rank)
    eng.set_factors(U0, V0)

    def barrier():
        eng.synchronize()
        if dist is not None:
            if backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()

    gather_buf = None
    if dist is not None:
        dev = "cuda" if backend == "nccl" else "cpu"
        gather_buf = (torch.empty((k, m), dtype=torch.float32, device=dev),
                      torch.empty((world, k, m), dtype=torch.float32, device=dev))

    def gather_components():
        """the np.vstack of enstop_.py:231 as one all-gather of the (k, m) topic matrices"""
        if dist is None:
            return
        send, recv = gather_buf
        if backend == "nccl":
            eng.copy_components_to_device(send.data_ptr())        # D2D into the RCCL send buffer
            dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
            torch.cuda.synchronize()
        else:
            _, V = eng.get_factors(want_u=False)
            send.copy_(torch.from_numpy(V))
            dist.all_gather(list(recv.unbind(0)), send)

    # ---- warmup (untimed): W EM iterations + the collective --------------------------------------
    if args.warmup > 0:
        it, _ = eng.fit(None, n_iter=args.warmup, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=flags)
        assert it == args.warmup
    gather_components()

    # ---- timed region: exactly K EM iterations ---------------------------------------------------
    eng.timing(True)
    eng.timing_reset()
    barrier()
    t0 = time.perf_counter()
    it, _ = eng.fit(None, n_iter=args.steps, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=flags)
    gather_components()
    barrier()
    dt = time.perf_counter() - t0
    assert it == args.steps, "early stop inside the timed region (%d of %d)" % (it, args.steps)
    report = eng.timing_report()
    eng.timing(False)

    if dist is not None:
        rdev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([float(nnz_act)], dtype=torch.float64, device=rdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        nnz_total = float(tot.item())
    else:
        nnz_total = float(nnz_act)

    # ---- per-kernel roofline figures from the HIP events of the timed region ----------------------
    kernels = {}
    for name, (cnt, ms) in report.items():
        kind = KERNEL_KIND.get(name)
        entry = {"launches": cnt, "avg_ms": round(ms / cnt, 5), "total_ms": round(ms, 4)}
        if kind:
            b = algorithmic_bytes(kind, n_act, m, nnz_act, k)
            entry["algorithmic_GB"] = round(b / 1e9, 4)
            entry["GBps"] = round(b / 1e9 / (ms / cnt / 1e3), 1)
        kernels[name] = entry
    dom = max((kv for kv in kernels.items() if "GBps" in kv[1]), key=lambda kv: kv[1]["total_ms"])

    def roof(name, e):
        return {"kernel": name, "bound": "hbm", "achieved": e["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(e["GBps"] / HBM_PEAK_GBS, 4), "traffic": None,
                "algorithmic_GB_per_launch": e["algorithmic_GB"], "avg_launch_ms": e["avg_ms"],
                "launches": e["launches"]}

    # ---- second timed leg: the reference's own kernel sequence (E-step -> M-step -> LL test) through
    # the materialised nnz x k array.  It carries the kernel the north_star roofline target is
    # stated on (k_e_step) and is reported next to the fused schedule, never instead of it.
    mat = None
    e_entry = kernels.get("k_e_step")
    if e_entry is None:
        k_mat = min(args.steps, 10)
        eng.set_factors(U0, V0)
        eng.fit(None, n_iter=2, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=0)   # warm-up
        eng.timing(True)
        eng.timing_reset()
        barrier()
        t1 = time.perf_counter()
        it2, _ = eng.fit(None, n_iter=k_mat, n_iter_per_test=10, tolerance=0.0, e_step_thresh=1e-32, flags=0)
        barrier()
        dt2 = time.perf_counter() - t1
        rep2 = eng.timing_report()
        eng.timing(False)
        kernels_mat = {}
        for name, (cnt, ms) in rep2.items():
            kind = KERNEL_KIND.get(name)
            entry = {"launches": cnt, "avg_ms": round(ms / cnt, 5), "total_ms": round(ms, 4)}
            if kind:
                bts = algorithmic_bytes(kind, n_act, m, nnz_act, k)
                entry["algorithmic_GB"] = round(bts / 1e9, 4)
                entry["GBps"] = round(bts / 1e9 / (ms / cnt / 1e3), 1)
            kernels_mat[name] = entry
        e_entry = kernels_mat["k_e_step"]
        mat = {"schedule": "materialised (reference kernel sequence)", "steps": it2, "p_placement": eng.placement_info(),
               "value": round(it2 / dt2, 4), "ms_per_step": round(dt2 / it2 * 1e3, 4), "kernels": kernels_mat}

    out = {
        "metric": "EM iterations/sec", "value": round(n_gpus * args.steps / dt, 4), "unit": "iter/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["name"], "n_docs": n, "n_vocab": m, "nnz": nnz, "k": k,
                   "schedule": args.schedule,
                   "parallelism": "single fit" if n_gpus == 1 else
                   "ensemble: one bootstrap member per GPU x%d, %s all-gather of topics" % (n_gpus, "RCCL" if backend == "nccl" else "gloo(host)"),
                   "ll_test_every": 10, "tolerance": 0.0, "e_step_thresh": 1e-32},
        "gcell_per_s": round(nnz_total * k * args.steps / dt / 1e9, 3),
        "ensemble_fits_per_min": round(n_gpus * args.steps / dt / FITS_ITERS * 60.0, 3),
        # north_star's roofline kernel: the per-nnz materialising E-step (HBM-bound)
        "roofline": roof("k_e_step", e_entry),
        # dominant kernel of the (fused) timed region against its own compulsory bytes; it is
        # gather/VALU-bound, not an HBM-roofline claim
        "roofline_dominant_fused": roof(*dom),
        "kernels": kernels,
        "materialised_leg": mat,
        "device": info["name"] or "AMD Instinct MI355X", "arch": info["arch"], "generate_s": round(t_gen, 2),
    }
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):        # HBM bytes per launch from committed rocprofv3 --pmc passes
        try:
            table = json.load(open(pmc)).get("config%d" % args.config, {})
            for key in ("roofline", "roofline_dominant_fused"):
                rec = table.get(out[key]["kernel"])
                if rec:
                    out[key]["traffic"] = rec["hbm_bytes_per_launch"]
                    out[key]["traffic_source"] = rec["source"]
        except Exception:
            pass
    if rank == 0:
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(eng, cfg, k)
            except Exception as e:       # the baseline must never cost the GPU measurement
                out["cpu_baseline"] = {"value": None, "unit": "iter/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
    if rank == 0 and n_gpus == 1 and dist is None and args.config == 3 and not args.no_cpu_baseline:
        # BASELINE.json configs[1] (100k x 50k, 10M nnz, k=32) on the same device, compact form
        try:
            out["other_configs"] = {"config2": quick_config(eng, 2, args.steps, args.warmup, args.seed)}
        except Exception as e:
            out["other_configs"] = {"config2": "failed: %r" % (e,)}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
