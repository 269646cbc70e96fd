"""Four-wave 256x256 kernel (tile cfg 25, csrc/gemm_quad.hip) against the eight-wave ring kernel (cfg 8) and the vendor library, same operands, one
process, interleaved rounds.  Checks on the way that cfg 25 and cfg 8 agree bit for bit (same products added in the same order).  tools/ only.

    python tools/time_quad.py [--rounds 5] [--iters 10] [--epi plain|bias_gelu|resid32]
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops

MT, MV = 256 * 256, 256 * 197
SHAPES = [(M, N, K) for M in (MT, MV) for (N, K) in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304))]


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--epi", default="plain")
    ap.add_argument("--no-vendor", action="store_true")
    ap.add_argument("--diag", action="store_true", help="K-loop ablation of cfg 25 (plain store): without DMA / fragment reads / MFMAs")
    args = ap.parse_args()
    dev, dt = torch.device("cuda", 0), torch.bfloat16
    if args.diag:
        names = {0: "full", 1: "no DMA", 2: "no frag reads", 4: "no MFMA", 3: "MFMA only", 5: "frag reads only", 6: "DMA only", 7: "barriers only"}
        for M, N, K in ((MT, 2304, 768), (MT, 768, 3072), (MT, 768, 768)):
            g = torch.Generator(device="cpu").manual_seed(1)
            a = torch.rand(M, K, generator=g).mul_(2).sub_(1).to(dev).to(dt)
            w = (torch.rand(N, K, generator=g).mul_(2).sub_(1) * K ** -0.5).to(dev).to(dt)
            pw = ops.PackedW(w)
            gk = ops.kb_group(dt)
            a_kb = ops.KBlocked(a.view(M, K // gk, gk).permute(1, 0, 2).contiguous())
            out = torch.empty(M, N, device=dev, dtype=dt)
            row = []
            for d in (0, 1, 2, 4, 3, 5, 6, 7):
                f = lambda: ops.gemm(a_kb, pw, out_t=out, tile_cfg=25 | (d << 16))
                for _ in range(3):
                    f()
                torch.cuda.synchronize()
                t = statistics.median(timeit(f, args.iters) for _ in range(args.rounds))
                row.append(f"{names[d]} {t:.1f}")
            f8 = lambda: ops.gemm(a_kb, pw, out_t=out, tile_cfg=8)
            for _ in range(3):
                f8()
            t8 = statistics.median(timeit(f8, args.iters) for _ in range(args.rounds))
            print(f"M={M} N={N} K={K} (us): cfg 8 {t8:.1f} | cfg 25: " + " | ".join(row), flush=True)
        return
    print(f"epilogue: {args.epi}")
    print(f"{'M':>6s} {'N':>5s} {'K':>5s} | {'vendor us':>10s} | {'cfg 8 us':>9s} {'TF/s':>6s} | {'cfg 25 us':>9s} {'TF/s':>6s} | 25/8   25/vendor  bit-equal")
    for M, N, K in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(M + N + K)
        a = torch.rand(M, K, generator=g).mul_(2).sub_(1).to(dev).to(dt)
        w = (torch.rand(N, K, generator=g).mul_(2).sub_(1) * K ** -0.5).to(dev).to(dt)
        bias = torch.rand(N, generator=g).to(dev)
        pw = ops.PackedW(w)
        gk = ops.kb_group(dt)
        a_kb = ops.KBlocked(a.view(M, K // gk, gk).permute(1, 0, 2).contiguous())
        kw = {}
        outs = {}
        for cfg in (8, 25):
            if args.epi == "resid32":
                outs[cfg] = torch.empty(M, N, device=dev, dtype=torch.float32)
            else:
                outs[cfg] = torch.empty(M, N, device=dev, dtype=dt)
        resid = torch.rand(M, N, generator=g).to(dev) if args.epi == "resid32" else None
        aux = torch.empty(M, N, device=dev, dtype=dt) if args.epi == "bias_gelu_aux" else None

        def run(cfg):
            if args.epi == "plain":
                ops.gemm(a_kb, pw, out_t=outs[cfg], tile_cfg=cfg)
            elif args.epi == "bias_gelu":
                ops.gemm(a_kb, pw, bias=bias, act="gelu", out_t=outs[cfg], tile_cfg=cfg)
            elif args.epi == "bias_gelu_aux":
                ops.gemm(a_kb, pw, bias=bias, act="gelu", aux_out=aux, out_t=outs[cfg], tile_cfg=cfg)
            elif args.epi == "resid32":
                ops.gemm(a_kb, pw, bias=bias, resid=resid, out32=outs[cfg], tile_cfg=cfg)
            else:
                raise SystemExit("unknown --epi")
        wt = w.t()
        out_v = torch.empty(M, N, device=dev, dtype=dt)
        f_v = lambda: torch.matmul(a, wt, out=out_v)
        fns = {"v": f_v, 8: lambda: run(8), 25: lambda: run(25)}
        if args.no_vendor:
            del fns["v"]
        for f in fns.values():
            for _ in range(3):
                f()
        torch.cuda.synchronize()
        equal = bool(torch.equal(outs[8], outs[25]))
        t = {k: [] for k in fns}
        for _ in range(args.rounds):
            for k, f in fns.items():
                t[k].append(timeit(f, args.iters))
        med = {k: statistics.median(v) for k, v in t.items()}
        fl = 2.0 * M * N * K
        v = med.get("v", float("nan"))
        print(f"{M:6d} {N:5d} {K:5d} | {v:10.1f} | {med[8]:9.1f} {fl / med[8] * 1e-6:6.0f} | {med[25]:9.1f} {fl / med[25] * 1e-6:6.0f} | {med[25] / med[8]:.3f}  {med[25] / v:.3f}    {equal}", flush=True)
        del a, w, pw, a_kb, outs, out_v


if __name__ == "__main__":
    main()
