#!/usr/bin/env python3
"""Register / LDS / occupancy table of the hot kernels, from hipcc's own remarks:

    python tools/kernel_resources.py [--out profiles/r04_kernel_resource_usage.txt] [-D PLSA_UNR_COL=6 ...]

Compiles enstop_amd/csrc/plsa_hip.hip for gfx950 with -Rpass-analysis=kernel-resource-usage (no GPU needed)
and prints one line per instantiation of the hot kernels for the BASELINE shapes:
k = 20 -> Shape<8,1,false>, k = 32 -> Shape<8,1,true>, k = 64 -> Shape<16,1,true>, k = 128 -> Shape<16,2,true>."""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enstop_amd import build as hb  # noqa: E402

HOT = ("k_col_pass", "k_row_pass", "k_e_step", "k_loglik", "k_col_reduce_norm", "k_col_tail")
SHAPES = {"Shape<8, 1, false>": "k=20", "Shape<8, 1, true>": "k=32", "Shape<16, 1, true>": "k=64",
          "Shape<16, 2, true>": "k=128"}


def collect(defines):
    cmd = [hb.HIPCC] + hb.FLAGS + ["-Rpass-analysis=kernel-resource-usage"] + ["-D" + d for d in defines] + \
          [hb.SRC, "-o", "/tmp/plsa_resource_probe.so", "-L" + os.path.join(hb.ROCM, "lib"), "-lrccl"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = re.split(r"remark: Function Name: ", txt)[1:]
    rows = []
    for b in blocks:
        name = b.split(" ")[0].strip()

        def g(key):
            mm = re.search(re.escape(key) + r": (\d+)", b)
            return int(mm.group(1)) if mm else -1
        rows.append([name, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize [bytes/lane]"),
                     g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")])
    dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
    out = []
    for r, d in zip(rows, dem):
        d = d.replace("plsa::", "").replace("(anonymous namespace)::", "")
        if not any(d.startswith("void " + h) or d.startswith(h) for h in HOT):
            continue
        shape = next((v for s, v in SHAPES.items() if s in d), None)
        if shape is None:
            continue
        short = re.sub(r"\(.*", "", d.replace("void ", ""))
        out.append((shape, short, r[1], r[2], r[3], r[4], r[5], r[6]))
    return sorted(set(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("-D", action="append", default=[])
    a = ap.parse_args()
    lines = ["# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, flags: %s %s" % (
        " ".join(hb.FLAGS), " ".join("-D" + d for d in a.D)),
        "# 512 VGPRs per SIMD lane: waves/SIMD = floor(512 / VGPRs rounded up to 8), at most 8",
        "%-6s %-78s %5s %5s %5s %8s %6s %8s" % ("shape", "kernel", "VGPR", "AGPR", "SGPR", "scratch", "waves", "LDS(B)")]
    for row in collect(a.D):
        lines.append("%-6s %-78s %5d %5d %5d %8d %6d %8d" % row)
    text = "\n".join(lines) + "\n"
    sys.stdout.write(text)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)
