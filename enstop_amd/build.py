"""In-tree build of libplsa_hip.so for gfx950:  python -m enstop_amd.build"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "plsa_hip.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "plsa_kernels.hpp"), os.path.join(HERE, "csrc", "plsa_ref_kernels.hpp"),
        os.path.join(HERE, "csrc", "plsa_synth.hpp"), os.path.join(HERE, "csrc", "mt_jump.hpp"),
        os.path.join(os.path.dirname(HERE), "include", "plsa_hip.h"), os.path.join(os.path.dirname(HERE), "include", "plsa_hip_diag.h")]
OUT = os.path.join(HERE, "libplsa_hip.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.environ.get("HIPCC", os.path.join(ROCM, "bin", "hipcc"))
# -fno-slp-vectorize: the SLP vectoriser pairs the per-topic multiplies/adds into v_pk_*_f32, which then
# need register-pair shuffles and keep the DPP moves of the group sums from folding into their adds;
# without it the fused document pass is 9 % faster (1.68 -> 1.52 ms at config 3), everything else equal
FLAGS = ["--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function", "-Wno-pass-failed"]


HOT = ("k_col_pass", "k_row_pass", "k_e_step", "k_loglik", "k_col_reduce_norm")
# (LPN, CH, FULL, WIDE = false: 32-bit gather offsets, every table below 4 GB); k = 64: the document pass runs as 8 x 2
SHAPES = {"Shape<8, 1, false, false>": "k=20", "Shape<8, 1, true, false>": "k=32", "Shape<16, 1, true, false>": "k=64",
          "Shape<8, 2, true, false>": "k=64", "Shape<16, 2, true, false>": "k=128"}
RESOURCES = os.path.join(HERE, "kernel_resources.json")


def parse_resource_remarks(text):
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> one record per instantiation of the hot kernels at
    the BASELINE shapes: registers, scratch, waves per SIMD (occupancy) and static LDS."""
    import re
    blocks = re.split(r"remark: Function Name: ", text)[1:]
    rows = []
    for b in blocks:
        def g(key):
            mm = re.search(re.escape(key) + r": (\d+)", b)
            return int(mm.group(1)) if mm else -1
        rows.append((b.split(" ")[0].strip(), g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize [bytes/lane]"),
                     g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
    if not rows:
        return []
    dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
    out = {}
    for r, d in zip(rows, dem):
        d = d.replace("plsa::", "").replace("void ", "")
        d = re.sub(r"\(.*", "", d)
        shape = next((v for s_, v in SHAPES.items() if s_ in d), None)
        if shape is None or not d.startswith(HOT):
            continue
        out[(shape, d)] = dict(shape=shape, kernel=d, vgprs=r[1], agprs=r[2], sgprs=r[3], scratch=r[4],
                               waves_per_simd=r[5], lds_bytes=r[6])
    return [out[key] for key in sorted(out)]


def build(force=False, verbose=True):
    deps = [d for d in DEPS if os.path.exists(d)]
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps)):
        return OUT
    # RCCL is linked directly: the multi-GPU exchange (plsa_comm_*) is part of the C ABI
    rocm_lib = os.path.join(ROCM, "lib")
    # -Rpass-analysis: the compiler's own register / occupancy report of every kernel, kept next to the library
    # (kernel_resources.json; bench.py quotes it per hot kernel)
    cmd = [HIPCC] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", SRC, "-o", OUT, "-L" + rocm_lib, "-lrccl",
                             "-Wl,-rpath," + rocm_lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    noise = ("remark:", "In file included from")
    rest = [ln for ln in proc.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in ln]
    # the remark lines are followed by source excerpts ("  468 | ...", "      | ^"): drop those too
    import re
    rest = [ln for ln in rest if not re.match(r"^\s*\d*\s*\|", ln) and not ln.startswith(noise)]
    if rest:
        sys.stderr.write("\n".join(rest) + "\n")
    if proc.returncode:
        raise subprocess.CalledProcessError(proc.returncode, cmd)
    try:
        import json
        with open(RESOURCES, "w") as f:
            json.dump({"flags": FLAGS, "kernels": parse_resource_remarks(proc.stderr)}, f, indent=1)
    except Exception as e:                      # the table is a by-product: never fail the build over it
        print("build: kernel resource table not written (%r)" % (e,), file=sys.stderr)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
