#!/usr/bin/env python3
"""bench.py — the headline measurement of BASELINE.json:

    images/sec, forward + backward + optimiser, BiomedCLIP ViT-B/16 + Mona fine-tune step,
    synthetic 224x224 image-text pairs, bs = 256 per GPU, bf16 operands  (BASELINE configs[1]; configs[2] at N=8)

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; every rank runs the same per-GPU workload (weak scaling) on its own synthetic batch and the
flat adapter-gradient buffer (1.34 M fp32) is all-reduced once per step with RCCL.  A "step" = encode_image (ViT-B/16,
12 blocks each followed by a Mona adapter) + encode_text (frozen BERT-base, dense L = 256, no padded-token skipping) +
InfoNCE + backward through blocks 11..1 and all 12 adapters + gradient clipping + AdamW.  Inputs are resident in HBM
before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "nextgen-uia_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# algorithmic work per image-caption pair: 69.95 GF image tower fwd+bwd (+Mona) + 45.90 GF text fwd
GFLOP_PER_PAIR = 115.86
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="pairs per GPU")
    ap.add_argument("--variant", default="freq_enhanced", help="Mona variant (reference default: biomedclip/finetune.py:76)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap-text", action="store_true", help="run the frozen text tower on a second HIP stream beside encode_image (+2 % pairs/s); "
                    "off by default so that the per-launch HIP-event durations behind `roofline` are not inflated by the other stream's kernels "
                    "and agree with the rocprofv3 summary of the same command")
    ap.add_argument("--no-overlap-text", action="store_true", help=argparse.SUPPRESS)        # former spelling of the default
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-ln-fold", action="store_true", help="A/B: run the stand-alone LayerNorm kernels instead of folding each frozen LayerNorm into the "
                    "GEMMs on either side of it (UF.set_ln_fold)")
    ap.add_argument("--no-deferred-text-ln", action="store_true", help="scheduling A/B (same results): every text-tower LayerNorm also writes its fp32 output "
                    "instead of leaving it to the consuming GEMM epilogue")
    ap.add_argument("--unpad-text", action="store_true", help="opt-in: the frozen text tower computes only the valid tokens of each caption "
                    "(identical features, less executed work than the reference's dense 256 positions; not the headline configuration)")
    ap.add_argument("--no-kblock-w", action="store_true", help="A/B knob: keep the GEMM weights row-major (default: K-blocked for the ring kernels)")
    ap.add_argument("--no-kblock-act", action="store_true", help="A/B knob: GEMM -> GEMM activations stay row-major (default: the producing epilogue writes them "
                    "K-blocked for the ring kernel that reads them)")
    ap.add_argument("--persist-store-only", action="store_true", help="experiment knob: store-only 256x256 GEMM launches on the persistent ring variant (tile cfg 12)")
    ap.add_argument("--tile-group", default="", help="experiment knob: N=group[,N=group] overrides the ring kernels' tile-order group (row panels per group; "
                    "255 = row-panel-major) for launches with that many columns, e.g. 3072=8,2304=4")
    ap.add_argument("--no-k64-cfg14", action="store_true", help="A/B knob: single-K-step GEMMs on the 256x256 tiles")
    ap.add_argument("--no-tail-split", action="store_true", help="A/B knob: no half-height tiles for the M tail of a launch")
    ap.add_argument("--global-loss", action="store_true", help="opt-in: InfoNCE over the global batch (all-gathered features) instead of "
                    "the reference-equivalent local loss; changes the objective, not the headline configuration")
    return ap.parse_args()


def synthetic_batch(batch, rank, device):
    import torch
    g = torch.Generator().manual_seed(1 + rank)                     # --seed default 1 (finetune.py:93), offset per rank
    images = torch.rand(batch, 3, 224, 224, generator=g)            # U[0,1), no mean/std normalisation (datasets/finetune.py:17-24)
    ids = torch.zeros(batch, 256, dtype=torch.long)
    lens = torch.randint(24, 129, (batch,), generator=g)
    for b in range(batch):
        n = int(lens[b])
        ids[b, 1:n - 1] = torch.randint(1000, 30000, (n - 2,), generator=g)
        ids[b, 0], ids[b, n - 1] = 2, 3                              # [CLS] / [SEP]; pad = 0
    return images.to(device), ids.to(device)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def text_attention_flops(ids, heads=12, dh=64, layers=12, pad_id=0):
    """(algorithmic, executed) attention FLOPs of the frozen text tower for one batch: the reference computes all L x L scores
    (HF BertSelfAttention with an additive mask); attn_fwd skips key tiles that are entirely padding, 64 keys (4 MFMA tiles) at a time."""
    B, L = ids.shape
    klen = (ids != pad_id).sum(dim=1).clamp(min=1)
    keys_exec = ((klen + 15) // 16 + 3) // 4 * 64
    keys_exec = keys_exec.clamp(max=(L + 15) // 16 * 16)
    per_key = 4.0 * heads * L * dh * layers
    return float(B * L * per_key), float(keys_exec.sum().item() * per_key)


def cpu_baseline(state, variant, batch, steps):
    """The oracle (CPU restatement of the reference path) timed on this host's cores: the reported CPU baseline."""
    import torch
    from oracle import train_ref
    torch.set_num_threads(min(16, os.cpu_count()))                  # more threads than this only adds contention at micro-batch 8
    g = torch.Generator().manual_seed(1)
    images = torch.rand(batch, 3, 224, 224, generator=g)
    ids = torch.zeros(batch, 256, dtype=torch.long)
    for b in range(batch):
        n = 24 + (b * 13) % 105
        ids[b, 1:n - 1] = torch.randint(1000, 30000, (n - 2,), generator=g)
        ids[b, 0], ids[b, n - 1] = 2, 3
    names = [k for k in state if "mona" in k]
    mona = dict(variant=variant, hw=(14, 14))
    times, budget = [], 30.0                                        # bounded sample: stop once ~30 s of CPU work are spent
    for i in range(steps + 1):
        t0 = time.perf_counter()
        train_ref.grads_of(lambda Pq, im, tk: train_ref.biomedclip_loss(Pq, im, tk, mona=mona), state, names, [(images, ids)])
        times.append(time.perf_counter() - t0)
        if sum(times) > budget:
            break
    timed = times[1:] if len(times) > 1 else times
    dt = sum(timed) / len(timed)
    return {"value": round(batch / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "host_logical_cpus": os.cpu_count(),
            "host_cpu_model": _cpu_model(), "kind": "port",
            "sample": f"{len(timed)} fwd+bwd step(s){' after 1 warm-up' if len(times) > 1 else ' (no warm-up: first step exceeded the 30 s budget)'} "
                      f"of the same model at micro-batch {batch}, fp32, oracle/train_ref.py",
            "s_per_step": round(dt, 3)}


def main():
    args = parse()
    import torch
    from uia_hip import functional as UF
    from uia_hip import ops
    from uia_hip.engine import FlatAdapterOptimizer, contrastive_step, init_data_parallel
    from src.adapters import inject_mona_variant_to_open_clip
    from src.losses import InfoNCELoss
    from src.third_party.biomedclip.model import create_biomedclip

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    UF.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    UF.set_unpad_text(args.unpad_text)
    UF.set_deferred_text_ln(not args.no_deferred_text_ln)
    UF.set_ln_fold(not args.no_ln_fold)
    ops.KBLOCK_W, ops.TAIL_SPLIT, ops.K64_CFG14 = not args.no_kblock_w, not args.no_tail_split, not args.no_k64_cfg14
    ops.KBLOCK_ACT = not args.no_kblock_act
    ops.PERSIST_STORE_ONLY = args.persist_store_only
    ops.TILE_GROUP = {int(k): int(v) for k, v in (kv.split("=") for kv in args.tile_group.split(",") if kv)}

    model = create_biomedclip(seed=0)                                # same weights on every rank (random init: no network for checkpoints)
    for p in model.parameters():
        p.requires_grad_(False)
    with contextlib.redirect_stdout(sys.stderr):          # the injector's banner (reference-compatible print) must not share stdout with the JSON line
        inject_mona_variant_to_open_clip(model, variant=args.variant, bottleneck_dim=64)
    for k, p in model.named_parameters():
        p.requires_grad_("mona" in k.lower())                        # finetune.py:173-175
    cpu_state = {k: v.detach().clone() for k, v in model.state_dict().items()} if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    model = model.to(device)
    model.train()                                                    # Mona dropout p=0.1 active (finetune.py:218)
    opt = FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.95),
                               weight_decay=0.01, max_norm=1.0)
    init_data_parallel(opt)
    criterion = InfoNCELoss(0.07)
    images, ids = synthetic_batch(args.batch, rank, device)
    UF.set_dropout_seed(1234 + rank)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    loss = None
    for _ in range(args.warmup):
        loss = contrastive_step(model, criterion, opt, images, ids, overlap_text=args.overlap_text, global_loss=args.global_loss)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        loss = contrastive_step(model, criterion, opt, images, ids, overlap_text=args.overlap_text, global_loss=args.global_loss)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    # per-launch HIP events (on the launch stream) around every uia_gemm of ONE more step of the same loop, outside the timed region:
    # the event records cost host time the throughput figure should not carry; the step itself is identical to the timed ones
    ops.GEMM_PROFILE = []
    contrastive_step(model, criterion, opt, images, ids, overlap_text=args.overlap_text, global_loss=args.global_loss)
    torch.cuda.synchronize()
    prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    prof_serial = prof
    if args.overlap_text:
        # one extra, untimed step with the two towers serialised on one stream: the kernels' rates without the other stream beside them
        ops.GEMM_PROFILE = []
        contrastive_step(model, criterion, opt, images, ids, overlap_text=False, global_loss=args.global_loss)
        torch.cuda.synchronize()
        prof_serial, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t[0])
    final_loss = float(loss)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * args.batch * args.steps / elapsed
        # ---- roofline of the dominant kernel.  Launches are grouped by the kernel instantiation they run on (tile config,
        #      compile-time epilogue mask) — the same granularity as a row of rocprofv3's kernel_stats.csv — and the
        #      instantiation with the largest share of the step is reported; the whole GEMM family is given beside it.
        def group(events, key):
            acc = {}
            for e0, e1, M, N, K, dt, cfg, nbytes, mask in events:
                d = acc.setdefault(key(cfg, mask), [0.0, 0.0, 0, 0.0])
                d[0] += e0.elapsed_time(e1) * 1e-3
                d[1] += 2.0 * M * N * K
                d[2] += 1
                d[3] += nbytes
            return acc
        def shape_table(events, peak):
            acc = {}
            for e0, e1, M, N, K, dt, cfg, nbytes, mask in events:
                d = acc.setdefault((cfg, mask if mask in ops._SPECIALISED else ops.EPI_GENERIC, M, N, K), [0.0, 0, nbytes])
                d[0] += e0.elapsed_time(e1) * 1e-3
                d[1] += 1
            rows = []
            for (cfg, mask, M, N, K), (tsec, n, nbytes) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
                tf = 2.0 * M * N * K * n / tsec * 1e-12
                rows.append({"kernel": ops.gemm_kernel_name(cfg, mask, torch.bfloat16 if args.dtype == "bf16" else torch.float32)[0], "M": M, "N": N, "K": K,
                             "launches": n, "avg_us": round(tsec / n * 1e6, 1), "tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / peak, 4),
                             "algorithmic_GBps": round(nbytes * n / tsec * 1e-9), "frac_of_hbm_spec": round(nbytes * n / tsec * 1e-12 / 8.0, 3)})
            return rows
        per_kernel = lambda c, m: (c, m if m in ops._SPECIALISED else ops.EPI_GENERIC)
        by_k, by_k_serial = group(prof, per_kernel), group(prof_serial, per_kernel)
        fam, fam_serial = group(prof, lambda c, m: c in (8, 12, 13, 14)), group(prof_serial, lambda c, m: c in (8, 12, 13, 14))
        dom = max(by_k, key=lambda k: by_k[k][0]) if by_k else None
        roof = None
        if dom is not None:
            tsec, flops, n, algo_bytes = by_k[dom]
            achieved = flops / tsec * 1e-12
            peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
            ts, fs, ns, _ = by_k_serial.get(dom, (tsec, flops, n, algo_bytes))
            kname, kmangled = ops.gemm_kernel_name(dom[0], dom[1], torch.bfloat16 if args.dtype == "bf16" else torch.float32)
            traffic, traffic_src = None, None
            tdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            tpath = os.path.join(tdir, "r02_traffic_pmc.json")
            if not os.path.exists(tpath):
                tpath = os.path.join(tdir, "r01_g_traffic_pmc.json")
            if args.dtype == "bf16" and args.batch == 256 and os.path.exists(tpath):
                # HBM bytes per launch of this kernel from rocprofv3 PMC passes over the same workload (tools/pmc_traffic.sh):
                # FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced
                # reads at 64 B, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B stores.
                pm = json.load(open(tpath))
                f, w = pm["FETCH_SIZE"].get(kmangled), pm["WRITE_SIZE"].get(kmangled)
                if f and w:
                    traffic = round((2.0 * f["sum"] / f["launches"] + w["sum"] / w["launches"]) * 1024)
                    traffic_src = (f"profiles/{os.path.basename(tpath)} (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, same "
                                   "workload; FETCH_SIZE doubled per the gfx950 note of the microarchitecture guide)")
            ft, ff, fn, _ = fam.get(True, (tsec, flops, n, 0.0))
            fts, ffs, fns, _ = fam_serial.get(True, (ts, fs, ns, 0.0))
            roof = {"bound": "mfma", "kernel": kname, "kernel_in_rocprof_csv": kmangled, "achieved": round(achieved, 1), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "bytes per launch (HBM, PMC)",
                    "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(algo_bytes / n), "launches_per_step": n,
                    "avg_launch_us": round(tsec / n * 1e6, 2), "flop_per_launch_avg": round(flops / n),
                    "share_of_step": round(tsec / (elapsed / args.steps), 3),
                    "note": ("HIP events around every launch of this kernel, on the launch stream, during one extra step of the same loop right after the timed region" +
                             ("; --overlap-text: the text tower runs on a second stream, so a launch's duration includes time shared with "
                              "that stream's kernels (see standalone)" if args.overlap_text else "")),
                    "gemm_family": {"kernels": "gemm_tn_ring_kernel<...,EPI> (+ gemm_tn_persist_kernel when selected), all epilogue masks",
                                    "launches_per_step": fn, "achieved": round(ff / ft * 1e-12, 1), "frac": round(ff / ft * 1e-12 / peak, 4)},
                    "per_shape": shape_table(prof_serial, peak),
                    # with --unpad-text the executed text-tower work is below the dense count GFLOP_PER_PAIR is quoted on: no fraction then
                    "whole_step_frac_of_peak": None if args.unpad_text else round(value / world * GFLOP_PER_PAIR * 1e-3 / peak, 4)}
            if not args.unpad_text:
                algo, execd = text_attention_flops(ids.cpu())
                gf_exec = GFLOP_PER_PAIR - (algo - execd) * 1e-9 / args.batch
                roof["text_attention_flops"] = {"algorithmic_GF_per_step": round(algo * 1e-9, 1), "executed_GF_per_step": round(execd * 1e-9, 1),
                                                "note": "fully padded 64-key chunks are skipped; every GEMM runs all 256 positions"}
                roof["gflop_per_pair_executed"] = round(gf_exec, 2)
                roof["whole_step_frac_of_peak_executed"] = round(value / world * gf_exec * 1e-3 / peak, 4)
            if args.overlap_text:
                roof["standalone"] = {"achieved": round(fs / ts * 1e-12, 1), "frac": round(fs / ts * 1e-12 / peak, 4), "avg_launch_us": round(ts / ns * 1e6, 2),
                                      "family_achieved": round(ffs / fts * 1e-12, 1), "how": "one extra untimed step with both towers on one stream"}
        out = {"metric": "images/sec fwd+bwd BiomedCLIP+Mona bs=256", "value": round(value, 2), "unit": "images/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "BiomedCLIP ViT-B/16 + Mona fine-tune step (BASELINE configs[1]): image tower fwd+bwd with 12 Mona adapters, "
                                      "frozen BERT-base text tower fwd (all 256 positions through every GEMM; attention skips key tiles that are entirely padding), "
                                      "InfoNCE, clip+AdamW; random-init weights",
                          "mona_variant": args.variant, "batch_per_gpu": args.batch, "global_batch": args.batch * world, "image": "3x224x224",
                          "text_len": 256, "text_positions_computed": "valid tokens only (opt-in --unpad-text)" if args.unpad_text else "all 256",
                          "parallelism": f"dp{world}", "text_tower_stream": "second stream" if args.overlap_text else "same stream", "contrastive_batch": "global (opt-in)" if args.global_loss else "per-rank (reference-equivalent)", "mona_dropout": 0.1, "bert_dropout_emulated": False, "layernorm": "stand-alone kernels" if args.no_ln_fold else "folded into the neighbouring GEMMs (row sums in the producer epilogue, normalised accumulators in the consumer)",
                          "gflop_per_pair_algorithmic": GFLOP_PER_PAIR},
               "loss": round(final_loss, 5), "roofline": roof}
        if cpu_state is not None:
            out["cpu_baseline"] = cpu_baseline(cpu_state, args.variant, args.cpu_batch, args.cpu_steps)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        ops.comm_destroy()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
