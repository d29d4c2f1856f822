#!/usr/bin/env python3
"""EXPERIMENT (round 5): the materialising E-step over (document, word-segment) pieces -- every XCD walks its documents once per
segment of the vocabulary, in dispatch order, so that the P(w|z) rows it gathers at any time are one segment (S segments of
m / S words: 3.2 MB at config 3 with S = 8) that its L2 can hold.  Pieces are built on the host here (numpy), handed to the
engine through the experiment hook plsa_set_e_pieces, and timed against the shipped traversal; P is compared bit for bit on a
corpus small enough to download."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                                   # noqa: E402
from enstop_amd.engine import Engine                           # noqa: E402


def build_pieces(X, S, max_len=64):
    n, m = X.shape
    indptr = X.indptr.astype(np.int64)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
    keys = rows * m + X.indices.astype(np.int64)
    assert np.all(np.diff(keys) > 0), "rows must hold sorted, distinct column indices"
    seg_w = -(-m // S)
    q = (np.arange(n, dtype=np.int64)[:, None] * m + np.minimum(np.arange(S + 1, dtype=np.int64)[None, :] * seg_w, m)).ravel()
    pos = np.searchsorted(keys, q).reshape(n, S + 1)
    start = pos[:, :-1].ravel(); end = pos[:, 1:].ravel()
    row = np.repeat(np.arange(n, dtype=np.int64), S)
    seg = np.tile(np.arange(S, dtype=np.int64), n)
    keep = end > start
    row, seg, start, end = row[keep], seg[keep], start[keep], end[keep]
    long_ = (end - start) > max_len                              # rare: cut into max_len chunks
    if long_.any():
        extra = []
        for r, s_, a, b in zip(row[long_], seg[long_], start[long_], end[long_]):
            for c in range(a + max_len, b, max_len):
                extra.append((r, s_, c, min(c + max_len, b)))
        end = np.where(long_, start + max_len, end)
        if extra:
            e = np.array(extra, np.int64)
            row = np.concatenate([row, e[:, 0]]); seg = np.concatenate([seg, e[:, 1]])
            start = np.concatenate([start, e[:, 2]]); end = np.concatenate([end, e[:, 3]])
    # XCD x takes the x-th eighth of the documents by nnz
    cuts = np.searchsorted(indptr, np.arange(1, 8) * indptr[-1] / 8.0)
    xcd = np.searchsorted(cuts, row, side="right")
    ln = end - start
    order = np.lexsort((-ln, seg, xcd))
    row, start, end, xcd = row[order], start[order], end[order], xcd[order]
    lo = np.searchsorted(xcd, np.arange(9)).astype(np.int32)
    return row.astype(np.int32), start.astype(np.int32), end.astype(np.int32), lo, float(ln.mean())


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    cfg = bench.CONFIGS[2 if what == "check" else 3]
    eng = Engine(0)
    L, h = eng._L, eng._h
    L.plsa_set_e_pieces.restype = C.c_int
    L.plsa_set_e_pieces.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    nnz = eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], seed=0)
    X = eng.download_active_csr()
    U0, V0 = bench.init_factors(cfg["n"], cfg["m"], cfg["k"], 42)
    eng.set_factors(U0, V0)

    def timed():
        eng.timing(True)
        eng.e_step(1e-32, want_host_copy=False)
        best = None
        for _ in range(3):
            eng.timing_reset()
            for _ in range(8):
                eng.e_step(1e-32, want_host_copy=False)
            ms, cnt = eng.timing_get("k_e_step")
            best = ms / cnt if best is None else min(best, ms / cnt)
        eng.timing(False)
        return best
    b = bench.algorithmic_bytes("e_step", cfg["n"], cfg["m"], nnz, cfg["k"])
    if what == "check":
        P0 = eng.e_step(1e-32).copy()
    base = timed()
    print(json.dumps({"traversal": "shipped", "ms": round(base, 4), "frac": round(b / 1e9 / (base / 1e3) / 8000.0, 4)}), flush=True)
    for S in ((8,) if what == "check" else (4, 8, 16)):
        t0 = time.time()
        row, start, end, lo, mean_len = build_pieces(X, S)
        build_s = time.time() - t0
        rc = L.plsa_set_e_pieces(h, row.ctypes.data, start.ctypes.data, end.ctypes.data, len(row), lo.ctypes.data)
        assert rc == 0
        if what == "check":
            P1 = eng.e_step(1e-32)
            print(json.dumps({"segments": S, "pieces": int(len(row)), "bit_identical": bool(np.array_equal(P0, P1))}), flush=True)
        t = timed()
        print(json.dumps({"traversal": "word segments", "segments": S, "pieces": int(len(row)), "mean_piece": round(mean_len, 1),
                          "ms": round(t, 4), "frac": round(b / 1e9 / (t / 1e3) / 8000.0, 4), "host_build_s": round(build_s, 1)}), flush=True)
        L.plsa_set_e_pieces(h, None, None, None, 0, None)
    eng.close()


if __name__ == "__main__":
    main()
