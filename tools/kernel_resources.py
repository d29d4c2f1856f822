#!/usr/bin/env python3
"""Register / LDS / occupancy table of the hot kernels, from hipcc's own remarks:

    python tools/kernel_resources.py [--out profiles/r04_kernel_resource_usage.txt] [-D PLSA_UNR_COL=6 ...]

Compiles enstop_amd/csrc/plsa_hip.hip for gfx950 with -Rpass-analysis=kernel-resource-usage (no GPU needed)
and prints one line per instantiation of the hot kernels for the BASELINE shapes:
k = 20 -> Shape<8,1,false>, k = 32 -> Shape<8,1,true>, k = 64 -> Shape<16,1,true>, k = 128 -> Shape<16,2,true>."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enstop_amd import build as hb  # noqa: E402

def collect(defines):
    cmd = [hb.HIPCC] + hb.FLAGS + ["-Rpass-analysis=kernel-resource-usage"] + ["-D" + d for d in defines] + \
          [hb.SRC, "-o", "/tmp/plsa_resource_probe.so", "-L" + os.path.join(hb.ROCM, "lib"), "-lrccl"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    return [(r["shape"], r["kernel"], r["vgprs"], r["agprs"], r["sgprs"], r["scratch"], r["waves_per_simd"], r["lds_bytes"])
            for r in hb.parse_resource_remarks(txt)]


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
