#!/usr/bin/env python3
"""Steady-state launch census of ONE training step from a rocprofv3 kernel trace (…_kernel_trace.csv): the launches between the last two
optimiser updates but one (adamw_guarded_kernel), grouped into this library's kernels and everything else (ATen elementwise / fill / reduce,
rocclr copy / fill blits, rocBLAS).  rocprofv3's --stats summary sums over the whole process — model set-up, weight packing, the host-to-device
copies of 150 M parameters — and divided by the step count it overstates the per-step foreign launches by an order of magnitude.

    python tools/kernel_census.py gpurun_out/<dir>/<name>_kernel_trace.csv > profiles/rNN_kernel_census_steady_step.txt
"""
import collections
import csv
import sys

OWN = ("gemm_", "attn_", "ln_", "mona_", "wgrad", "embed", "im2col", "dfeat", "scale_cast", "pack_weights", "transpose_cast", "infonce", "adamw", "sumsq", "accum_guarded",
       "guarded", "gather_rows", "fill_cls", "cast_kernel", "dropout", "layernorm", "colsum", "three_byte", "rowsum", "logits_kernel", "normalize_kernel", "dlogits", "lse_kernel",
       "lora_", "film_", "col2im", "unshuffle", "shuffle", "act_bwd", "upsample", "segment_mean", "dicece")


def main(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "adamw_guarded" in r["Kernel_Name"] or "adamw_kernel" in r["Kernel_Name"]]
    if len(marks) < 3:
        sys.exit("fewer than three optimiser updates in the trace")
    a, b = marks[-3], marks[-2]
    win = rows[a + 1:b + 1]
    own, other = collections.Counter(), collections.Counter()
    t_own = t_other = 0
    for r in win:
        n, dur = r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        short = n.replace("(anonymous namespace)::", "").replace("void ", "")[:100]
        if any(m in n for m in OWN):
            own[short] += 1
            t_own += dur
        else:
            other[short] += 1
            t_other += dur
    print(f"trace: {path}")
    print(f"one steady-state step (between optimiser updates {len(marks) - 2} and {len(marks) - 1} of {len(marks)}): {len(win)} launches, "
          f"{sum(own.values())} of this library ({t_own / 1e6:.3f} ms of kernel time), {sum(other.values())} foreign ({t_other / 1e3:.1f} us)")
    print("\nforeign launches:")
    for k, v in other.most_common():
        print(f"  {v:4d}  {k}")
    print("\nthis library's launches:")
    for k, v in own.most_common():
        print(f"  {v:4d}  {k}")


if __name__ == "__main__":
    main(sys.argv[1])
