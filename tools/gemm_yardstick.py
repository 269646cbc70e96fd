"""Yardstick for the ring GEMM (tools/ only — the vendor library is never on the product path).

Times, on the same device, in ONE process, interleaved rounds (cdna_hip_programming.md §5.4 rule 24), on the same random bf16 operands:
    * torch.matmul(a, w.T)            -> hipBLASLt / rocBLAS, plain bf16 output
    * uia_gemm, epilogue mask 128     -> this repo's ring kernel, plain bf16 output (row-major A; and K-blocked A + K-blocked W as the step runs it)
for the twelve (M, N, K) bf16 shapes of one config-2 training step (BENCH per_shape: M in {65 536, 50 432}, N in {768, 2304, 3072},
K in {768, 3072}) plus 4096^3 / 8192^3 as the usual vendor reference points.

    python tools/gemm_yardstick.py [--rounds 5] [--iters 10] [--out gpurun_out/gemm_yardstick.json]
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops

MT, MV = 256 * 256, 256 * 197
SHAPES = [(M, N, K) for M in (MT, MV) for (N, K) in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304))]
SHAPES += [(MV, 768, 64), (MV, 64, 768), (4096, 4096, 4096), (8192, 8192, 8192)]


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "gemm_yardstick.json"))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    dt = torch.bfloat16
    rows = []
    print(f"{'M':>6s} {'N':>5s} {'K':>5s} | {'hipBLASLt us':>12s} {'TF/s':>7s} | {'uia rm us':>10s} {'TF/s':>7s} | {'uia kb us':>10s} {'TF/s':>7s} | uia/vendor (best)")
    for M, N, K in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(M + N + K)
        a = torch.rand(M, K, generator=g).mul_(2).sub_(1).to(dev).to(dt)               # uniform [-1, 1): full-range random operands (rule 25)
        w = (torch.rand(N, K, generator=g).mul_(2).sub_(1) * K ** -0.5).to(dev).to(dt)
        wt = w.t()
        pw = ops.PackedW(w)
        out_v = torch.empty(M, N, device=dev, dtype=dt)
        out_u = torch.empty(M, N, device=dev, dtype=dt)
        kb_ok = ops.kb_ok(M, N, K, dt) if N > 64 else False
        a_kb = None
        if kb_ok:
            gk = ops.kb_group(dt)
            a_kb = ops.KBlocked(a.view(M, K // gk, gk).permute(1, 0, 2).contiguous())
        f_v = lambda: torch.matmul(a, wt, out=out_v)
        f_rm = lambda: ops.gemm(a, pw, out_t=out_u)
        f_kb = (lambda: ops.gemm(a_kb, pw, out_t=out_u)) if a_kb is not None else None
        for f in (f_v, f_rm, f_kb):
            if f is not None:
                for _ in range(3):
                    f()
        torch.cuda.synchronize()
        # parity of the two on the way (fp32 accumulate on both sides; outputs rounded to bf16 once)
        err = float((out_u.float() - out_v.float()).abs().max() / (out_v.float().abs().max() + 1e-12))
        t = {"vendor": [], "uia_rowmajor": [], "uia_kblocked": []}
        for _ in range(args.rounds):
            t["vendor"].append(timeit(f_v, args.iters))
            t["uia_rowmajor"].append(timeit(f_rm, args.iters))
            if f_kb is not None:
                t["uia_kblocked"].append(timeit(f_kb, args.iters))
        fl = 2.0 * M * N * K
        row = {"M": M, "N": N, "K": K, "max_rel_diff_uia_vs_vendor": err}
        for k, v in t.items():
            if v:
                row[k] = {"median_us": round(statistics.median(v), 1), "min_us": round(min(v), 1), "tflops_median": round(fl / statistics.median(v) * 1e-6, 1)}
        best = min(row[k]["median_us"] for k in ("uia_rowmajor", "uia_kblocked") if k in row)
        row["uia_over_vendor_time"] = round(best / row["vendor"]["median_us"], 3)
        rows.append(row)
        kb = row.get("uia_kblocked", {"median_us": float("nan"), "tflops_median": float("nan")})
        print(f"{M:6d} {N:5d} {K:5d} | {row['vendor']['median_us']:12.1f} {row['vendor']['tflops_median']:7.1f} | {row['uia_rowmajor']['median_us']:10.1f} "
              f"{row['uia_rowmajor']['tflops_median']:7.1f} | {kb['median_us']:10.1f} {kb['tflops_median']:7.1f} | {row['uia_over_vendor_time']:.3f}  (diff {err:.1e})", flush=True)
        del a, w, out_v, out_u, a_kb, pw
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "rounds": args.rounds, "iters": args.iters,
               "how": "HIP events around `iters` back-to-back launches, `rounds` interleaved rounds per shape in one process; uniform [-1,1) bf16 operands; "
                      "vendor = torch.matmul (hipBLASLt), plain bf16 output; uia = uia_gemm epilogue mask 128 (plain bf16 output)", "shapes": rows},
              open(args.out, "w"), indent=1)
    print("written", args.out)


if __name__ == "__main__":
    main()
