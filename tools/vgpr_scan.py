"""Register counts of every kernel of the library against the occupancy steps of a 512-register SIMD (gfx950): compiles each .hip to assembly (device only, no GPU
needed) and lists kernels that spill or sit a few registers above a step (96 = 5 waves, 128 = 4, 168 = 3, 256 = 2).  Found round 4's LayerNorm-backward regression.

    python tools/vgpr_scan.py [--all]
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "nextgen-uia_amd", "csrc")
STEPS = [(64, 8), (72, 7), (80, 6), (96, 5), (128, 4), (168, 3), (256, 2)]


def main():
    show_all = "--all" in sys.argv
    os.makedirs("/tmp/vgpr_scan", exist_ok=True)
    for f in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
        name = os.path.basename(f)[:-4]
        out = f"/tmp/vgpr_scan/{name}.s"
        extra = ["-fno-slp-vectorize"] if name == "attention_bwd" else []
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-result", "-S", "--cuda-device-only", f, "-o", out] + extra,
                           capture_output=True, text=True)
        if r.returncode:
            print(name, "did not compile:", r.stderr[-300:])
            continue
        txt = open(out).read()
        for blk in re.split(r"\n  - \.agpr_count", txt)[1:]:
            n, v, sp = re.search(r"\.name:\s+(\S+)", blk), re.search(r"\.vgpr_count:\s+(\d+)", blk), re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
            if not (n and v):
                continue
            vg, spill = int(v.group(1)), int(sp.group(1)) if sp else 0
            note = ""
            for st, waves in STEPS:
                if st < vg <= st + 10:
                    note = f"{vg - st} over the {waves}-wave step"
            if show_all or spill or note:
                print(f"{name:16s} {n.group(1)[:90]:90s} vgpr {vg:3d} spill {spill:3d}  {note}")


if __name__ == "__main__":
    main()
