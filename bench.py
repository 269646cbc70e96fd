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

# the hosts of this pool support dmabuf IPC only: without this RCCL (and any device-buffer hand-off between the ranks of one node) fails with
# `hipIpcGetMemHandle: invalid argument`.  Read when the HSA runtime starts, i.e. at the first HIP call of the process — set it before anything can make one.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "nextgen-uia_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# algorithmic work per image-caption pair: 69.95 GF image tower fwd+bwd (+Mona) + 45.90 GF text fwd
GFLOP_PER_PAIR = 115.86
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
TRAFFIC_FILE = "r06_traffic_pmc.json"   # tools/pmc_traffic.sh on the tree that is benchmarked; tests/test_host_logic.py checks that every instantiation is in it


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="mona", choices=["mona", "clipseg", "vitl_lora"],
                    help="mona (default) = the headline: BASELINE configs[1] / [2]; clipseg = configs[3] (OpenAI ViT-B/16 + FiLM decoder, bs 128, DiceCE); "
                         "vitl_lora = the per-GPU shape of configs[4] (ViT-L/14 + LoRA r=16, 128 pairs per GPU).  The secondary lines carry the same keys")
    ap.add_argument("--batch", type=int, default=0, help="samples per GPU (default: 256 for mona, 128 for the secondary configs)")
    ap.add_argument("--variant", default="freq_enhanced", help="Mona variant (reference default: biomedclip/finetune.py:76)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap-text", action="store_true", help=argparse.SUPPRESS)           # the default since round 4 (kept so that old command lines still parse)
    ap.add_argument("--no-overlap-text", action="store_true", help="run the frozen text tower on the image tower's stream.  Default (round 4): on a second HIP "
                    "stream beside encode_image (engine.contrastive_micro(overlap_text=True), also what the fine-tune entry points run; +2.5 %% pairs/s); `roofline` then comes from one "
                    "extra untimed step with both towers on ONE stream, where a launch's HIP events see only that launch (and agree with rocprofv3)")
    ap.add_argument("--no-entry-point", action="store_true", help="skip the `entry_point` form: the fine-tune CLI (src/models/biomedclip/finetune.py --method mona --synthetic, bs 256, "
                    "one update per batch) run as a child process after the timed region, its steady-state ms per update printed beside the headline")
    ap.add_argument("--entry-steps", type=int, default=60, help="updates per epoch of the entry-point run (3 epochs; the first is warm-up)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the two secondary lines (BASELINE configs[3] and the per-GPU shape of configs[4], 15 steps each, "
                    "no CPU leg) that the default single-GPU run prints under `secondary`")
    ap.add_argument("--also-streams", type=int, default=1, help="after the timed region, time 5 more steps each of two other forms of the step (everything on one stream; the text tower on "
                    "valid tokens only) and report them beside the headline as `other_forms` (0 = skip)")
    ap.add_argument("--streams", type=int, default=1, help="cut each rank's batch into this many slices whose towers run on as many HIP streams (same batch, "
                    "same single InfoNCE over all pairs, same gradients: engine.contrastive_step(streams=...))")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-cpu-wide", action="store_true", help="skip the second CPU sample (micro-batch 32)")
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
    ap.add_argument("--bf16-heads", action="store_true", help="A/B knob: the two feature heads on bf16 operands (default: fp32 operands, UF.set_fp32_heads)")
    ap.add_argument("--wgrad-side-stream", action="store_true", help="A/B knob: the adapters' weight-gradient launches on a second HIP stream beside the data-gradient chain")
    ap.add_argument("--mona-fused", action="store_true", help="A/B knob: the adapter forward as ONE launch (uia_mona_fused_fwd) instead of pre, project1, spatial, project2")
    ap.add_argument("--no-tail-split", action="store_true", help="A/B knob: no half-height tiles for the M tail of a launch")
    ap.add_argument("--no-tail-side-stream", action="store_true", help="A/B: small M tails of a split GEMM behind the main launch instead of beside it on a side stream")
    ap.add_argument("--no-direct-train-grads", action="store_true", help="A/B: the trainable decoder / head layers hand their weight gradients to autograd instead of accumulating them into the flat gradient buffer")
    ap.add_argument("--no-lora-regen-drop", action="store_true", help="A/B: the LoRA forward writes the dropped input rows out for the dA weight-gradient launch instead of that launch regenerating the mask")
    ap.add_argument("--no-mona-pre-bwd-du", action="store_true", help="A/B: project1's data gradient as a K = 64 GEMM launch in front of the Mona pre-norm backward instead of inside it")
    ap.add_argument("--no-mona-pre-fwd-t", action="store_true", help="A/B: project1 as the N = 64 stream launch behind the Mona pre-norm kernel instead of inside it")
    ap.add_argument("--no-ln-lora-down", action="store_true", help="A/B: the LoRA block's LayerNorm and its three down-projections as four launches instead of one (uia_ln_lora_down)")
    ap.add_argument("--short-k-half-n", type=int, default=-1, help="experiment knob: short-K bf16 launches with N at or above this value run on half-height tiles (tile cfg 14); 0 = off, -1 = the library default")
    ap.add_argument("--short-k-half-bytes", type=int, default=-1, help="experiment knob: the longest K row in bytes the --short-k-half-n rule applies to (-1 = the library default)")
    ap.add_argument("--short-k-half-no-stash", action="store_true", help="experiment knob: the --short-k-half-n rule skips launches with an aux_out stash")
    ap.add_argument("--half-height-short-k-always", action="store_true", help="experiment knob: N <= 768, K <= 768 launches on half-height tiles also without a ragged last round")
    ap.add_argument("--image-split", type=float, default=-1.0, help="scheduling knob: the image tower as two slices (this fraction of the images, the rest) on two HIP streams beside the text tower's; 0 = one slice, -1 = the engine's default")
    ap.add_argument("--no-text-ahead", action="store_true", help="A/B: the frozen text tower's stream waits for the previous step's backward and optimiser launches (default here: contrastive_step(inputs_ready=True) — "
                    "the synthetic batch is resident in HBM before the timed region — so each step's text tower starts as soon as the host has enqueued it)")
    ap.add_argument("--text-slices", type=int, default=1, help="experiment knob: the text tower in this many slices on as many streams beside the image tower's slices")
    ap.add_argument("--image-slices", type=int, default=2, help="experiment knob: number of image-tower slices (streams) when --image-split is on")
    ap.add_argument("--no-grad-resid3", action="store_true", help="A/B: the residual gradient between the image tower's backward Functions as fp32 + bf16 copy (rounds 1-3) instead of a three-byte tensor")
    ap.add_argument("--no-lora-wgrad-group", action="store_true", help="A/B: the q | k | v weight gradients of a LoRA attention block as three launches each instead of one uia_wgrad_group launch")
    ap.add_argument("--no-lora-rank3", action="store_true", help="A/B: the q | k | v rank terms of a LoRA block's data gradient as three K = 64 launches instead of one uia_lora_rank_update pass")
    ap.add_argument("--no-lora-kext", action="store_true", help="A/B knob: the LoRA rank update as a launch of its own (tile cfg 23) instead of inside the frozen GEMM's K loop")
    ap.add_argument("--quad", action="store_true", help="experiment knob: 256x256 bf16 launches on the four-wave kernel (tile cfg 25, csrc/gemm_quad.hip) instead of the eight-wave ring kernel")
    ap.add_argument("--quadv", type=int, nargs="?", const=27, default=0, choices=(0, 27, 29), help="experiment knob: 256x256 bf16 launches on the four-wave register-staged kernel (tile cfg 27; 29 = its persistent grid; csrc/gemm_quadv.hip)")
    ap.add_argument("--ring5", action="store_true", help="experiment knob: long-K / wide-N 256x256 launches on the 5-deep ring (tile cfg 24) instead of the 4-deep one")
    ap.add_argument("--no-tail-split-k", action="store_true", help="A/B knob: the M tail launches run their whole K chain (default: long-K tails of a few tiles are split over K)")
    ap.add_argument("--no-half-height-short-k", action="store_true", help="A/B knob: N <= 768, K <= 768 launches with a ragged last round as main + tail launches (round 2) "
                    "instead of one launch on half-height tiles")
    ap.add_argument("--no-text-resid3", action="store_true", help="A/B knob: the text tower's sub-layer sums as fp32 + bf16 copy (rounds 2-3) instead of three-byte tensors "
                    "(bf16 copy + one low byte per element: 6 epilogue bytes per element instead of 10 on its 23 N = 768 launches)")
    ap.add_argument("--no-fwd-resid3", action="store_true", help="A/B knob: Mona adapters hand their output to the next frozen block as fp32 + bf16 copy (rounds 1-4) instead of a three-byte tensor")
    ap.add_argument("--block-resid3", action="store_true", help="A/B knob (measured level, off by default): x1 / dx1 inside the frozen image blocks as three-byte tensors instead of fp32 + bf16 copy")
    ap.add_argument("--attn-bwd-cfg", type=int, default=0, help="A/B knob: kernel configuration of the bf16 attention backward (uia_attn_bwd_cfg: 0 = the library's choice, "
                    "1 = the lock-step kernel of rounds 1-3, 2 = barrier-free units, 5 = their persistent form)")
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


def _cpu_sample(fn, batch, steps, threads, what, budget=30.0):
    """`fn()` = one fwd+bwd of the oracle at micro-batch `batch`; bounded: stops once ~budget seconds of CPU work are spent."""
    import torch
    torch.set_num_threads(threads)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
        if sum(times) > budget:
            break
    timed = times[1:] if len(times) > 1 else times
    dt = sum(timed) / len(timed)
    return {"value": round(batch / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "host_logical_cpus": os.cpu_count(),
            "host_cpu_model": _cpu_model(), "kind": "port",
            "sample": f"{len(timed)} fwd+bwd step(s){' after 1 warm-up' if len(times) > 1 else ' (no warm-up: first step exceeded the budget)'} "
                      f"of the same model at micro-batch {batch}, fp32, {what}",
            "s_per_step": round(dt, 3)}


def _cpu_share():
    """CPUs this process may actually use: the scheduler affinity mask, cut by the cgroup quota (a GPU box hands a 1-GPU job a 16-CPU
    share of its 256 logical CPUs: 256 threads on that share ran one micro-batch-32 step in 298 s — profiles/r03_a_bench_default.json)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, int(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(state, variant, batch, steps, wide=True):
    """The oracle (CPU restatement of the reference path) timed on this host's cores: the reported CPU baseline.  Threads = the CPUs this
    process is ALLOWED to use (affinity mask and cgroup quota, capped at 32: beyond that micro-batch 8 only adds contention) — BASELINE.md §3
    says os.cpu_count(), which on the GPU boxes counts 256 logical CPUs of which a 1-GPU job may use 16.  wide: a second bounded sample at
    micro-batch 32 on the same threads, reported under "micro_batch_32"."""
    import torch
    from oracle import train_ref
    names = [k for k in state if "mona" in k]
    mona = dict(variant=variant, hw=(14, 14))
    threads = max(1, min(32, _cpu_share()))

    def make(bs):
        g = torch.Generator().manual_seed(1)
        images = torch.rand(bs, 3, 224, 224, generator=g)
        ids = torch.zeros(bs, 256, dtype=torch.long)
        for b in range(bs):
            n = 24 + (b * 13) % 105
            ids[b, 1:n - 1] = torch.randint(1000, 30000, (n - 2,), generator=g)
            ids[b, 0], ids[b, n - 1] = 2, 3
        return lambda: train_ref.grads_of(lambda Pq, im, tk: train_ref.biomedclip_loss(Pq, im, tk, mona=mona), state, names, [(images, ids)])

    out = _cpu_sample(make(batch), batch, steps, threads, "oracle/train_ref.py", budget=20.0)
    out["cpu_share"] = _cpu_share()
    if wide:
        out["micro_batch_32"] = _cpu_sample(make(32), 32, 1, threads, "oracle/train_ref.py", budget=10.0)
    return out


def gemm_roofline(prof, prof_serial, args, ms_per_step, ops, torch):
    """Roofline of the dominant GEMM instantiation from per-launch HIP events.  Launches are grouped by the kernel instantiation they run
    on (tile config, compile-time epilogue mask) — the same granularity as a row of rocprofv3's kernel_stats.csv — and the instantiation
    with the largest share of the step is reported; the whole GEMM family and a per-(kernel, M, N, K) table are given beside it."""
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3

    def group(events, key):
        acc = {}
        for e0, e1, M, N, K, _dt, cfg, nbytes, mask in events:
            d = acc.setdefault(key(cfg, mask), [0.0, 0.0, 0, 0.0])
            d[0] += e0.elapsed_time(e1) * 1e-3
            d[1] += 2.0 * M * N * K
            d[2] += 1
            d[3] += nbytes
        return acc

    def shape_table(events):
        acc = {}
        for e0, e1, M, N, K, _dt, cfg, nbytes, mask in events:
            d = acc.setdefault((cfg, mask if mask in ops._SPECIALISED else ops.EPI_GENERIC, M, N, K), [0.0, 0, nbytes])
            d[0] += e0.elapsed_time(e1) * 1e-3
            d[1] += 1
        rows = []
        for (cfg, mask, M, N, K), (tsec, n, nbytes) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
            tf = 2.0 * M * N * K * n / tsec * 1e-12
            rows.append({"kernel": ops.gemm_kernel_name(cfg, mask, dt)[0], "M": M, "N": N, "K": K,
                         "launches": n, "avg_us": round(tsec / n * 1e6, 1), "tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / peak, 4),
                         "algorithmic_GBps": round(nbytes * n / tsec * 1e-9), "frac_of_hbm_spec": round(nbytes * n / tsec * 1e-12 / 8.0, 3)})
        return rows

    per_kernel = lambda c, m: (c, m if m in ops._SPECIALISED else ops.EPI_GENERIC)
    multi = prof_serial is not prof
    by_k_inflight, fam_inflight = group(prof, per_kernel), group(prof, lambda c, m: c in (8, 12, 13, 14, 24))
    # more than one stream in the timed step: the kernel's rate comes from the serialised extra step (a launch's events then bracket that launch only)
    by_k, fam = group(prof_serial, per_kernel), group(prof_serial, lambda c, m: c in (8, 12, 13, 14, 24))
    if not by_k:
        return None
    dom = max(by_k, key=lambda k: by_k[k][0])
    tsec, flops, n, algo_bytes = by_k[dom]
    achieved = flops / tsec * 1e-12
    ts, fs, ns, _ = by_k_inflight.get(dom, (tsec, flops, n, algo_bytes))
    kname, kmangled = ops.gemm_kernel_name(dom[0], dom[1], dt)
    traffic, traffic_src, traffic_err = None, None, None
    tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    if not (args.config == "mona" and args.dtype == "bf16" and args.batch == 256):
        traffic_err = "PMC passes exist for the headline workload only (mona, bf16, 256 per GPU)"
    elif not os.path.exists(tpath):
        traffic_err = f"profiles/{TRAFFIC_FILE} is missing: run tools/pmc_traffic.sh on the GPU box and commit its summary.json under that name"
    else:
        # HBM bytes per launch of this kernel from rocprofv3 PMC passes over the same workload (tools/pmc_traffic.sh):
        # FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced
        # reads at 64 B, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B stores.
        pm = json.load(open(tpath))
        f, w = pm["FETCH_SIZE"].get(kmangled), pm["WRITE_SIZE"].get(kmangled)
        if f and w:
            traffic = round((2.0 * f["sum"] / f["launches"] + w["sum"] / w["launches"]) * 1024)
            traffic_src = (f"profiles/{os.path.basename(tpath)} (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, same "
                           "workload, measured in the round the file name carries; FETCH_SIZE doubled per the gfx950 note of the microarchitecture guide)")
        else:
            # a renamed instantiation (a new template parameter) must not turn into a silent null again (BENCH_r03)
            traffic_err = f"kernel {kmangled} is not in profiles/{TRAFFIC_FILE}: the instantiation was renamed after the PMC passes; re-run tools/pmc_traffic.sh"
    ft, ff, fn, _ = fam.get(True, (tsec, flops, n, 0.0))
    fts, ffs, fns, _ = fam_inflight.get(True, (ts, fs, ns, 0.0))

    def union_seconds(events):
        """Wall time during which at least one launch of the family was running: small M tails of a split launch run on a side stream BESIDE their main launch
        (ops.TAIL_SIDE_STREAM), so the SUM of the launch durations counts that time twice (ViT-L/14: 128-row tails of 80-130 us beside every main launch)."""
        fam_ev = [(e0, e1) for e0, e1, M, N, K, _dt, cfg, nb, mask in events if cfg in (8, 12, 13, 14, 24)]
        if not fam_ev:
            return None
        base = fam_ev[0][0]
        iv = sorted((base.elapsed_time(e0), base.elapsed_time(e1)) for e0, e1 in fam_ev)
        total, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
        for s_, e_ in iv[1:]:
            if s_ > cur_e:
                total += cur_e - cur_s
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        return (total + cur_e - cur_s) * 1e-3

    fu = union_seconds(prof_serial)
    # in-kernel clock under sustained GEMM load (s_memtime / s_memrealtime, profiles/r03_a_inkernel_clock.txt): 1.51-1.65 GHz on this pool's
    # devices against the 2.4 GHz the 2.5 PF datasheet peak is quoted at; `frac` stays against the datasheet peak
    load_clock = {"measured_GHz": [1.51, 1.65], "source": "profiles/r03_a_inkernel_clock.txt (test_gemm_stamps: s_memtime / s_memrealtime x 100 MHz per workgroup, after 6-12 k warm launches)",
                  "mfma_peak_at_that_clock_TFLOPs": [round(peak * 1.51 / 2.4), round(peak * 1.65 / 2.4)],
                  "frac_of_that_peak": [round(achieved / (peak * 1.65 / 2.4), 4), round(achieved / (peak * 1.51 / 2.4), 4)]} if args.dtype == "bf16" else None
    roof = {"bound": "mfma", "kernel": kname, "kernel_in_rocprof_csv": kmangled, "achieved": round(achieved, 1), "peak": peak,
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "bytes per launch (HBM, PMC)",
            "traffic_source": traffic_src, "traffic_error": traffic_err, "algorithmic_bytes_per_launch": round(algo_bytes / n), "launches_per_step": n,
            "avg_launch_us": round(tsec / n * 1e6, 2), "flop_per_launch_avg": round(flops / n),
            "share_of_step": round(tsec / (ms_per_step * 1e-3), 3), "load_clock": load_clock,
            "note": ("HIP events around every launch of this kernel, on the launch stream, during one extra step right after the timed region" +
                     (": that step runs both towers on ONE stream (the timed steps use more than one; a launch's events would then include time "
                      "shared with the other stream's kernels - those figures are under `in_timed_configuration`)" if multi else " of the same loop")),
            "gemm_family": {"kernels": "gemm_tn_ring_kernel<...,EPI> (+ gemm_tn_persist_kernel when selected), all epilogue masks",
                            "launches_per_step": fn, "achieved": round(ff / ft * 1e-12, 1), "frac": round(ff / ft * 1e-12 / peak, 4),
                            "ms_per_step": round(ft * 1e3, 3),
                            "busy_ms_per_step": None if fu is None else round(fu * 1e3, 3), "frac_over_busy_time": None if fu is None else round(ff / fu * 1e-12 / peak, 4),
                            "busy_note": "ms_per_step sums the launches' durations; busy_ms_per_step is the time at least one of them was running (M tails run beside their main launch on a side stream)"},
            "per_shape": shape_table(prof_serial)}
    if multi:
        roof["in_timed_configuration"] = {"achieved": round(fs / ts * 1e-12, 1), "frac": round(fs / ts * 1e-12 / peak, 4), "avg_launch_us": round(ts / ns * 1e6, 2),
                                          "family_achieved": round(ffs / fts * 1e-12, 1),
                                          "how": "the same events during a step with the streams of the timed region: durations include the other stream's kernels"}
    return roof


from uia_hip.telemetry import PowerSampler      # noqa: E402  (amdgpu hwmon files: board power and shader clock inside the timed region)


POWER = {}


def timed_loop(step, args, world, device, ops, torch):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize; returns (seconds MAX over ranks, per-rank list,
    last loss, events of one extra profiled step [, of a serialised one])."""
    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    multi = args.overlap_text or getattr(args, "streams", 1) > 1            # more than one stream: per-launch events also see the other stream's kernels
    mode = True if args.overlap_text else None                               # step(False) = everything on one stream
    loss = None
    for _ in range(args.warmup):
        loss = step(mode)
    barrier()
    sampler = PowerSampler(torch, device)
    sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(mode)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    POWER[args.config] = sampler.stop()
    # per-launch HIP events (on the launch stream) around every uia_gemm of ONE more step of the same loop, outside the timed region:
    # the event records cost host time the throughput figure should not carry; the step itself is identical to the timed ones
    ops.GEMM_PROFILE = []
    step(mode)
    torch.cuda.synchronize()
    prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    prof_serial = prof
    if multi:
        # one extra, untimed step with the two towers serialised on one stream: the kernels' rates without the other stream beside them
        ops.GEMM_PROFILE = []
        step(False)
        torch.cuda.synchronize()
        prof_serial, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    per_rank = [elapsed]
    if world > 1:
        t = torch.zeros(world, device=device, dtype=torch.float64)
        t[int(os.environ.get("RANK", 0))] = elapsed
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
        per_rank = [float(v) for v in t.tolist()]
        elapsed = max(per_rank)
    return elapsed, per_rank, float(loss), prof, prof_serial


def dist_fields(world, per_rank, steps, ops):
    """What the communicator itself saw (not the environment): a multi-GPU line must prove RCCL had N ranks."""
    ms = [e / steps * 1e3 for e in per_rank]
    return {"rccl_world": int(ops.comm_world()), "rccl_initialised": bool(ops.comm_world_initialised()), "env_world_size": world,
            "ms_per_step_per_rank": {"min": round(min(ms), 3), "max": round(max(ms), 3), "ranks": len(ms)}}


def main():
    args = parse()
    # stdout carries ONE JSON line and nothing else: file descriptor 1 is pointed at stderr for the whole run (RCCL prints a version banner from C when a
    # communicator is created, injectors print reference-compatible banners) and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    from uia_hip import functional as UF
    from uia_hip import ops
    from uia_hip.engine import FlatAdapterOptimizer, contrastive_step, init_data_parallel

    args.overlap_text = not args.no_overlap_text and args.streams == 1 and args.config == "mona"
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    if not args.batch:
        args.batch = 256 if args.config == "mona" else 128
    args.entry_point_result = None
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.upper().startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if profiled and not args.no_entry_point:
        # under rocprofv3 the profiler's preloaded library has initialised the GPU before this program started, and a child would inherit the tool: no child process
        args.entry_point_result = {"skipped": "running under rocprofv3: the entry-point child process is not started (pass --no-entry-point to silence)"}
    elif world == 1 and args.config == "mona" and not args.no_entry_point and not args.no_overlap_text and args.streams == 1:
        # FIRST, before this process has touched the GPU (a program started from a process with a live HIP runtime is not allowed on this pool, and the two would share the
        # device): the fine-tune CLI as a child process, run to completion; its figure is attached to the line below
        args.entry_point_result = entry_point_form(args, f"cuda:{local}")
    args.clipseg_entry_point_result = None
    if not profiled and world == 1 and not args.no_entry_point and ((args.config == "mona" and not args.no_secondary) or args.config == "clipseg"):
        args.clipseg_entry_point_result = entry_point_clipseg(args, f"cuda:{local}", 128 if args.config == "mona" else args.batch)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if not args.batch:
        args.batch = 256 if args.config == "mona" else 128
    UF.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    UF.set_unpad_text(args.unpad_text)
    UF.set_deferred_text_ln(not args.no_deferred_text_ln)
    UF.set_ln_fold(not args.no_ln_fold)
    ops.KBLOCK_W, ops.TAIL_SPLIT, ops.K64_CFG14 = not args.no_kblock_w, not args.no_tail_split, not args.no_k64_cfg14
    ops.TAIL_SPLIT_K = not args.no_tail_split_k
    ops.RING5 = args.ring5
    ops.QUAD = args.quad
    ops.QUADV = args.quadv
    ops.LORA_KEXT = not args.no_lora_kext
    ops.LORA_RANK3 = not args.no_lora_rank3
    ops.LORA_WGRAD_GROUP = not args.no_lora_wgrad_group
    if args.short_k_half_n >= 0:
        ops.SHORT_K_WIDE_HALF_N = args.short_k_half_n
    ops.SHORT_K_WIDE_HALF_STASH = not args.short_k_half_no_stash
    ops.HALF_HEIGHT_SHORT_K_ALWAYS = args.half_height_short_k_always
    if args.short_k_half_bytes >= 0:
        ops.SHORT_K_WIDE_HALF_BYTES = args.short_k_half_bytes
    ops.LN_LORA_DOWN = not args.no_ln_lora_down
    ops.MONA_PRE_FWD_T = not args.no_mona_pre_fwd_t
    ops.MONA_PRE_BWD_DU = not args.no_mona_pre_bwd_du
    ops.LORA_REGEN_DROP = not args.no_lora_regen_drop
    UF.DIRECT_TRAIN = not args.no_direct_train_grads
    ops.TAIL_SIDE_STREAM = not args.no_tail_side_stream
    ops.KBLOCK_ACT = not args.no_kblock_act
    ops.HALF_HEIGHT_SHORT_K = not args.no_half_height_short_k
    ops.MONA_FUSED = args.mona_fused
    ops.ATTN_BWD_CFG = args.attn_bwd_cfg
    UF.set_wgrad_side_stream(args.wgrad_side_stream)
    UF.set_fp32_heads(not args.bf16_heads)
    UF.set_text_resid3(not args.no_text_resid3)
    from uia_hip import engine as _engine
    _engine.GRAD_RESID3 = not args.no_grad_resid3
    if args.image_split >= 0:
        _engine.IMAGE_SPLIT = args.image_split
    _engine.IMAGE_SLICES = args.image_slices
    _engine.TEXT_SLICES = args.text_slices
    UF.set_block_resid3(args.block_resid3)
    UF.set_fwd_resid3(not args.no_fwd_resid3)
    ops.PERSIST_STORE_ONLY = args.persist_store_only
    ops.TILE_GROUP = {int(k): int(v) for k, v in (kv.split("=") for kv in args.tile_group.split(",") if kv)}
    try:
        out = {"mona": bench_mona, "clipseg": bench_clipseg, "vitl_lora": bench_vitl_lora}[args.config](args, rank, world, device)
    except Exception:
        # a rank that dies here (a failed RCCL init above all) must take the job down with a non-zero code on EVERY rank: its peers
        # are most likely blocked inside a collective, and torch.distributed.run only tears the group down when a worker exits non-zero
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(13)
    if rank == 0:
        out["power"] = POWER.get(args.config)                # rank 0's GPU during the timed steps (None where the driver's hwmon files are not readable)
    if rank == 0 and world == 1 and args.config == "mona" and not args.no_secondary:
        out["secondary"] = secondary_lines(args, device)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if ops.comm_world_initialised():
        ops.comm_destroy()
    if world > 1:
        torch.distributed.destroy_process_group()


def secondary_lines(args, device):
    """BASELINE configs[3] and the per-GPU shape of configs[4] inside the default line (the driver runs only the default command): 15 timed steps
    each, no CPU leg, the same keys as `python bench.py --config clipseg|vitl_lora` prints, cut to what a reader needs."""
    import copy
    import gc
    import torch
    from uia_hip import functional as UF
    lines = {}
    for cfg, fn in (("clipseg", bench_clipseg), ("vitl_lora", bench_vitl_lora)):
        a = copy.copy(args)
        a.config, a.batch, a.steps, a.warmup, a.no_cpu_baseline, a.overlap_text, a.streams = cfg, 128, 15, 5, True, False, 1      # (10 + 3 until round 6: one 100 ms hiccup inside ten steps once read as 78.8 ms for a 67.7 ms step)
        UF.clear_t_copies()
        gc.collect()
        torch.cuda.empty_cache()
        try:
            o = fn(a, 0, 1, device)
            r = o.get("roofline") or {}
            lines[cfg] = {"metric": o["metric"], "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"], "warmup": o["warmup"],
                          "dtype": o["dtype"], "data": o["data"], "config": o["config"], "loss": o["loss"], **({"entry_point": o["entry_point"]} if "entry_point" in o else {}),
                          "power": POWER.get(cfg),
                          "roofline": {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches_per_step", "avg_launch_us",
                                                             "share_of_step", "whole_step_frac_of_peak", "gflop_per_image_executed", "whole_step_frac_of_peak_executed", "executed_note", "gemm_family")}}
        except Exception as e:                      # a secondary line must never cost the headline
            lines[cfg] = {"error": f"{type(e).__name__}: {e}"}
    return lines


def entry_point_form(args, device):
    """The north-star boundary itself, timed: `python src/models/biomedclip/finetune.py --method mona --synthetic --batch_size 256 --accumulation_steps 1` as a CHILD
    process (its loader workers fork before it touches the GPU), three epochs of --entry-steps updates; the first epoch is warm-up, the figure is the wall time of the
    other two (first micro-batch enqueued -> device idle after the last update, engine.ContrastiveLoop / DevicePrefetcher) over their updates.  Same engine functions
    as the timed region above (contrastive_micro + guarded accumulate / update); on top of them the real loader: batches generated by worker processes, pinned, copied
    host-to-device on a copy stream, tokenised on the host."""
    import subprocess
    import tempfile
    script = os.path.join(ROOT, "nextgen-uia_amd", "src", "models", "biomedclip", "finetune.py")
    with tempfile.TemporaryDirectory() as td:
        stats = os.path.join(td, "stats.json")
        cmd = [sys.executable, script, "--method", "mona", "--mona_variant", args.variant, "--synthetic", "--synthetic_train", str(args.batch * args.entry_steps),
               "--synthetic_val", str(args.batch), "--batch_size", str(args.batch), "--accumulation_steps", "1", "--epochs", "3", "--patience", "99", "--dtype", args.dtype,
               "--exp", "bench_entry_point", "--device", str(device), "--stats_json", stats]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, cwd=td, capture_output=True, text=True, timeout=420)
        except subprocess.TimeoutExpired:
            return {"error": "the entry-point child process did not finish in 420 s", "command": " ".join(cmd[1:])}
        except OSError as e:
            return {"error": f"could not start the entry-point child process: {e}", "command": " ".join(cmd[1:])}
        wall = time.perf_counter() - t0
        if r.returncode != 0 or not os.path.exists(stats):
            return {"error": f"exit code {r.returncode}", "stderr_tail": r.stderr[-1500:], "command": " ".join(cmd[1:])}
        out = json.load(open(stats))
    steady = out["epochs"][1:]
    ms = sum(e["ms"] for e in steady) / max(1, sum(e["updates"] for e in steady))
    return {"ms_per_step": round(ms, 3), "value": round(args.batch / ms * 1e3, 2), "unit": "images/s", "vs_headline_ms": None,
            "epochs": [{k: (round(v, 1) if isinstance(v, float) else v) for k, v in e.items()} for e in out["epochs"]], "updates": out["updates"], "last_train_loss": round(out["last_train"], 5),
            "child_wall_s": round(wall, 1),
            "what": "src/models/biomedclip/finetune.py main() as a child process: synthetic pairs from loader workers -> pinned staging -> copy stream -> engine.ContrastiveLoop "
                    "(contrastive_micro + device-guarded accumulate / clip + AdamW, cosine LR on the device) ; epochs 2-3 of 3, wall time per optimiser update",
            "command": "python " + " ".join(os.path.relpath(c, ROOT) if c == script else c for c in cmd[1:-1]) + " <tmp>"}


def entry_point_clipseg(args, device, batch):
    """BASELINE configs[3]'s boundary, timed: `python src/models/clipseg/segmentation.py --dataset BUSI --synthetic --batch_size 128` as a CHILD process — the
    reference's loop (real loader workers -> shared ring -> copy stream -> engine.segmentation_step, cosine schedule, validation + test pass + best-Dice
    checkpoint at the last epoch, then test()); three epochs of --entry-steps iterations, the first is warm-up, the figure is the wall time of the other two
    over their iterations (validation is outside the epochs' clocks, as in the fine-tune form)."""
    import subprocess
    import tempfile
    script = os.path.join(ROOT, "nextgen-uia_amd", "src", "models", "clipseg", "segmentation.py")
    with tempfile.TemporaryDirectory() as td:
        stats = os.path.join(td, "stats.json")
        cmd = [sys.executable, script, "--dataset", "BUSI", "--synthetic", "--synthetic_train", str(batch * args.entry_steps), "--synthetic_val", str(batch),
               "--synthetic_test", str(batch), "--batch_size", str(batch), "--epochs", "3", "--dtype", args.dtype, "--exp", "bench_entry_point_clipseg",
               "--device", str(device), "--stats_json", stats]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, cwd=td, capture_output=True, text=True, timeout=420)
        except subprocess.TimeoutExpired:
            return {"error": "the entry-point child process did not finish in 420 s", "command": " ".join(cmd[1:])}
        except OSError as e:
            return {"error": f"could not start the entry-point child process: {e}", "command": " ".join(cmd[1:])}
        wall = time.perf_counter() - t0
        if r.returncode != 0 or not os.path.exists(stats):
            return {"error": f"exit code {r.returncode}", "stderr_tail": r.stderr[-1500:], "command": " ".join(cmd[1:])}
        out = json.load(open(stats))
    steady = out["epochs"][1:]
    ms = sum(e["ms"] for e in steady) / max(1, sum(e["updates"] for e in steady))
    return {"ms_per_step": round(ms, 3), "value": round(batch / ms * 1e3, 2), "unit": "images/s", "vs_line_ms": None,
            "epochs": [{k: (round(v, 1) if isinstance(v, float) else v) for k, v in e.items()} for e in out["epochs"]], "iters": out["iters"],
            "best_val_dice": out["best_val_dice"], "child_wall_s": round(wall, 1),
            "what": "src/models/clipseg/segmentation.py main() as a child process: synthetic image/mask pairs from loader workers -> shared-memory ring -> copy stream -> "
                    "engine.segmentation_step (the step this line times), cosine LR per iteration; epochs 2-3 of 3, wall time per iteration; validation, test pass, "
                    "best-Dice checkpoint and test() run after the last epoch, outside the clock",
            "command": "python " + " ".join(os.path.relpath(c, ROOT) if c == script else c for c in cmd[1:-1]) + " <tmp>"}


def _image_split_on(args):
    from uia_hip import engine
    return 0 < int(round(args.batch * engine.IMAGE_SPLIT)) < args.batch and args.batch >= 32


def bench_mona(args, rank, world, device):
    import torch
    from uia_hip import functional as UF
    from uia_hip import ops
    from uia_hip.engine import FlatAdapterOptimizer, contrastive_step, init_data_parallel
    from src.adapters import inject_mona_variant_to_open_clip
    from src.losses import InfoNCELoss
    from src.third_party.biomedclip.model import create_biomedclip

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
    init_data_parallel(opt, force_comm=True)                 # world 1 too: the one-rank RCCL communicator, so that the N = 1 line already times uia_allreduce_sum
    criterion = InfoNCELoss(0.07)
    images, ids = synthetic_batch(args.batch, rank, device)
    UF.set_dropout_seed(1234 + rank)

    step = lambda overlap: contrastive_step(model, criterion, opt, images, ids, overlap_text=bool(overlap) and args.streams == 1, global_loss=args.global_loss,
                                            streams=args.streams if overlap is not False else 1, inputs_ready=not args.no_text_ahead)
    elapsed, per_rank, final_loss, prof, prof_serial = timed_loop(step, args, world, device, ops, torch)
    multi = None
    if args.also_streams > 0:
        # Two other forms of the same step, 5 steps each, reported beside the headline and never as it:
        #   one_stream    — everything on one stream (the configuration the roofline block is measured on): what the streams are worth;
        #   unpadded_text — the frozen text tower computes only the valid tokens of each caption (UF.set_unpad_text: same features, LESS work than the reference
        #                   executes, which is why it is not the headline).
        def extra(fn):
            for _ in range(2):
                fn()
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                lv = fn()
            torch.cuda.synchronize()
            dte = time.perf_counter() - t0
            if world > 1:
                te = torch.tensor([dte], device=device, dtype=torch.float64)
                torch.distributed.all_reduce(te, op=torch.distributed.ReduceOp.MAX)
                dte = float(te[0])
            return {"steps": 5, "ms_per_step": round(dte / 5 * 1e3, 3), "value": round(world * args.batch * 5 / dte, 2), "loss": round(float(lv), 5)}

        multi = {"how": "5 untimed-in-`value` steps each, same model and batch",
                 "one_stream": extra(lambda: contrastive_step(model, criterion, opt, images, ids, overlap_text=False, global_loss=args.global_loss, image_split=0))}
        if not args.unpad_text:
            UF.set_unpad_text(True)
            try:
                multi["unpadded_text"] = extra(lambda: step(True))
                multi["unpadded_text"]["note"] = "valid caption tokens only through the text tower (opt-in --unpad-text): identical features, less work than the reference executes"
            finally:
                UF.set_unpad_text(False)
    if rank != 0:
        return None
    ms = elapsed / args.steps * 1e3
    value = world * args.batch * args.steps / elapsed
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
    roof = gemm_roofline(prof, prof_serial, args, ms, ops, torch)
    if roof is not None:
        # with --unpad-text the executed text-tower work is below the dense count GFLOP_PER_PAIR is quoted on: no fraction then
        roof["whole_step_frac_of_peak"] = None if args.unpad_text else round(value / world * GFLOP_PER_PAIR * 1e-3 / peak, 4)
        if not args.unpad_text:
            algo, execd = text_attention_flops(ids.cpu())
            gf_exec = GFLOP_PER_PAIR - (algo - execd) * 1e-9 / args.batch
            roof["text_attention_flops"] = {"algorithmic_GF_per_step": round(algo * 1e-9, 1), "executed_GF_per_step": round(execd * 1e-9, 1),
                                            "note": "fully padded 64-key chunks are skipped; every GEMM runs all 256 positions"}
            roof["gflop_per_pair_executed"] = round(gf_exec, 2)
            roof["whole_step_frac_of_peak_executed"] = round(value / world * gf_exec * 1e-3 / peak, 4)
    out = {"metric": "images/sec fwd+bwd BiomedCLIP+Mona bs=256", "value": round(value, 2), "unit": "images/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "BiomedCLIP ViT-B/16 + Mona fine-tune step (BASELINE configs[1]): image tower fwd+bwd with 12 Mona adapters, "
                                  "frozen BERT-base text tower fwd (all 256 positions through every GEMM; attention skips key tiles that are entirely padding), "
                                  "InfoNCE, clip+AdamW; random-init weights",
                      "mona_variant": args.variant, "batch_per_gpu": args.batch, "global_batch": args.batch * world, "image": "3x224x224",
                      "text_len": 256, "text_positions_computed": "valid tokens only (opt-in --unpad-text)" if args.unpad_text else "all 256",
                      "parallelism": f"dp{world}", "text_tower_stream": ("second HIP stream beside the image tower, whose two half-batch slices run on two streams (engine.IMAGE_SPLIT: engine.contrastive_micro, the function the fine-tune entry points call per loader batch — `entry_point` below is that CLI timed); roofline from one extra step with everything on one stream"
                                            if args.overlap_text else "same stream"),
                      "text_tower_start": ("as soon as enqueued (inputs resident, tower frozen: its stream does not wait for the previous step's backward / optimiser; every step computes its own text features)"
                                           if (args.overlap_text and args.streams == 1 and not args.no_text_ahead) else "behind the previous step"),
                      "image_tower_slices": (2 if (args.overlap_text and args.streams == 1 and _image_split_on(args)) else 1),
                      "hip_streams": (f"{args.streams}: the batch's towers run as {args.streams} slices on {args.streams} HIP streams, one InfoNCE over all pairs" if args.streams > 1 else 1),
                      "contrastive_batch": "global (opt-in)" if args.global_loss else "per-rank (reference-equivalent)", "mona_dropout": 0.1,
                      "bert_dropout_emulated": False,
                      "layernorm": "stand-alone kernels" if args.no_ln_fold else "folded into the neighbouring GEMMs (row sums in the producer epilogue, normalised accumulators in the consumer)",
                      "gflop_per_pair_algorithmic": GFLOP_PER_PAIR},
           "loss": round(final_loss, 5), "roofline": roof}
    out.update(dist_fields(world, per_rank, args.steps, ops))
    out["other_forms"] = multi
    ep = getattr(args, "entry_point_result", None)
    if ep is not None:
        if "ms_per_step" in ep:
            ep["vs_headline_ms"] = round(ep["ms_per_step"] / ms, 4)
        out["entry_point"] = ep
    out["cpu_baseline"] = cpu_baseline(cpu_state, args.variant, args.cpu_batch, args.cpu_steps, wide=not args.no_cpu_wide) if cpu_state is not None else None
    return out


def bench_clipseg(args, rank, world, device):
    """BASELINE configs[3]: CLIPSeg — frozen OpenAI ViT-B/16 (QuickGELU) taps 3/6/9 + text prompt -> FiLM decoder (1 127 009 trainable), DiceCE,
    bs 128 per GPU (reference: src/models/clipseg/segmentation.py:78-160, clipseg_adapter.py:73-98).  36.5 GF per image."""
    import torch
    from uia_hip import functional as UF
    from uia_hip import ops
    from uia_hip.engine import FlatAdapterOptimizer, init_data_parallel
    from src.models.clipseg import segmentation as S
    from src.losses.dice import DiceCELoss
    GF = 36.5
    # executed per image in the steady state: blocks 10-11 of the frozen ViT are never run (their outputs reach nothing: clipseg_adapter.py extract layers 3/6/9),
    # and the one prompt's text features are a cached constant of the run: 29.31 (ViT blocks 0-9 + patch embed) + 0.45 + 0.90 (decoder fwd + bwd)
    GF_EXEC = 29.31 + 0.45 + 0.90
    sargs = S.get_args(["--synthetic", "--batch_size", str(args.batch)])
    sargs.device = str(device)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = S.prepare_model(sargs)
    cpu_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()} if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    opt = FlatAdapterOptimizer([(n, p) for n, p in model.named_parameters() if p.requires_grad], lr=sargs.lr, betas=(sargs.beta1, sargs.beta2),
                               weight_decay=sargs.weight_decay, max_norm=0.0)          # the entry point's optimiser (reference clipseg/segmentation.py:121-126)
    init_data_parallel(opt, force_comm=True)                 # world 1 too: the one-rank RCCL communicator, so that the N = 1 line already times uia_allreduce_sum
    crit = S.criterion
    images, labels = S.synthetic_batch(args.batch, 224, 1 + rank, str(device))
    sargs.dataset = "BUSI"
    prompt = S.get_prompt(sargs).to(device).repeat(args.batch, 1)          # the reference's BUSI prompt, 68 tokens (src/models/clipseg/prompt_ids.json)
    from uia_hip.engine import segmentation_step

    def step(_overlap):
        return segmentation_step(model, crit, opt, images, labels, input_ids=prompt)[0]      # the function the entry point calls per loader batch

    elapsed, per_rank, final_loss, prof, prof_serial = timed_loop(step, args, world, device, ops, torch)
    if rank != 0:
        return None
    ms = elapsed / args.steps * 1e3
    value = world * args.batch * args.steps / elapsed
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
    roof = gemm_roofline(prof, prof_serial, args, ms, ops, torch)
    if roof is not None:
        roof["whole_step_frac_of_peak"] = round(value / world * GF * 1e-3 / peak, 4)
        roof["gflop_per_image_executed"] = round(GF_EXEC, 2)
        roof["whole_step_frac_of_peak_executed"] = round(value / world * GF_EXEC * 1e-3 / peak, 4)
        roof["executed_note"] = "ViT blocks 10-11 (5.8 GF) are skipped and the prompt tower (5.96 GF per distinct prompt) is cached: the algorithmic count includes both, the executed one neither"
    out = {"metric": "images/sec fwd+bwd CLIPSeg ViT-B/16 + FiLM decoder bs=128 (BASELINE configs[3]; secondary line)", "value": round(value, 2),
           "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "CLIPSeg segmentation step (BASELINE configs[3]): frozen OpenAI ViT-B/16 forward with taps 3/6/9 + prompt text tower, "
                                  "FiLM decoder fwd+bwd (1 127 009 trainable), DiceCE, AdamW; random-init weights, synthetic ellipse masks",
                      "batch_per_gpu": args.batch, "global_batch": args.batch * world, "image": "3x224x224", "parallelism": f"dp{world}",
                      "gflop_per_image_algorithmic": GF},
           "loss": round(final_loss, 5), "roofline": roof}
    out.update(dist_fields(world, per_rank, args.steps, ops))
    ep = getattr(args, "clipseg_entry_point_result", None)
    if ep is not None:
        if "ms_per_step" in ep:
            ep["vs_line_ms"] = round(ep["ms_per_step"] / ms, 4)
        out["entry_point"] = ep
    out["cpu_baseline"] = None
    if cpu_state is not None:
        from oracle import clipseg_ref, losses_ref
        bs = 4
        im, lab = images[:bs].cpu(), labels[:bs].cpu().float()
        pr = prompt[:bs].cpu()
        names = [k for k in cpu_state if k.startswith("decoder.")]

        def fn():
            leaves = {k: cpu_state[k].clone().requires_grad_(True) for k in names}
            Pq = dict(cpu_state)
            Pq.update(leaves)
            losses_ref.dice_ce(clipseg_ref.adapter_forward(im, pr, Pq, vit_heads=12, text_heads=8, extract_layers=(3, 6, 9)), lab).backward()
        out["cpu_baseline"] = _cpu_sample(fn, bs, 3, max(1, min(32, _cpu_share())), "oracle/clipseg_ref.py", budget=20.0)
    return out


def bench_vitl_lora(args, rank, world, device):
    """Per-GPU shape of BASELINE configs[4]: in-tree CLIP with a ViT-L/14 image tower (24 blocks, width 1024, 16 heads, 257 tokens) and the
    12-layer causal text tower, LoRA r = 16 (alpha 32, dropout 0.1) on q, k, v, o of the image tower through inject_lora_to_clip
    (reference src/adapters/lora.py:202-248), contrastive step at 128 pairs per GPU.  335 + 13.3 GF per pair."""
    import torch
    from uia_hip import functional as UF
    from uia_hip import ops
    from uia_hip.engine import FlatAdapterOptimizer, contrastive_step, init_data_parallel
    from src.adapters import inject_lora_to_clip
    from src.losses import InfoNCELoss
    from src.third_party.openai_clip.model import CLIP
    GF = 335.0 + 13.3
    torch.manual_seed(0)
    model = CLIP(768, 224, 24, 1024, 14, 77, 49408, 768, 12, 12)
    for p in model.parameters():
        p.requires_grad_(False)
    with contextlib.redirect_stdout(sys.stderr):
        model, n = inject_lora_to_clip(model, lora_r=16, lora_alpha=32, lora_dropout=0.1)
    for k, p in model.named_parameters():
        p.requires_grad_("lora" in k.lower())
    cpu_state = {k: v.detach().clone() for k, v in model.state_dict().items()} if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    trainable_names = [k for k, p in model.named_parameters() if p.requires_grad]
    model = model.to(device).train()
    opt = FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.95), weight_decay=0.01, max_norm=1.0)
    init_data_parallel(opt, force_comm=True)                 # world 1 too: the one-rank RCCL communicator, so that the N = 1 line already times uia_allreduce_sum
    g = torch.Generator().manual_seed(1 + rank)
    B = args.batch
    images = torch.rand(B, 3, 224, 224, generator=g)
    ids = torch.zeros(B, 77, dtype=torch.long)
    for b in range(B):
        L = int(torch.randint(8, 60, (1,), generator=g))
        ids[b, 0] = 49406
        ids[b, 1:L] = torch.randint(1000, 40000, (L - 1,), generator=g)
        ids[b, L] = 49407
    images_d, ids_d = images.to(device), ids.to(device)
    crit = InfoNCELoss(0.07)
    UF.set_dropout_seed(3 + rank)
    step = lambda overlap: contrastive_step(model, crit, opt, images_d, ids_d, overlap_text=overlap, inputs_ready=not args.no_text_ahead)
    elapsed, per_rank, final_loss, prof, prof_serial = timed_loop(step, args, world, device, ops, torch)
    if rank != 0:
        return None
    ms = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
    roof = gemm_roofline(prof, prof_serial, args, ms, ops, torch)
    if roof is not None:
        roof["whole_step_frac_of_peak"] = round(value / world * GF * 1e-3 / peak, 4)
        roof["gflop_per_image_executed"] = GF
        roof["whole_step_frac_of_peak_executed"] = roof["whole_step_frac_of_peak"]
        roof["executed_note"] = "nothing is skipped or cached in this step: LoRA in rank form is what SURVEY's 335 GF counts, the text tower runs all 77 positions every step"
    out = {"metric": "images/sec fwd+bwd ViT-L/14 + LoRA r=16, 128 pairs/GPU (per-GPU shape of BASELINE configs[4]; secondary line)", "value": round(value, 2),
           "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "contrastive fine-tune step on the in-tree CLIP with a ViT-L/14 image tower (24 blocks, width 1024, 257 tokens) + LoRA r=16 on "
                                  "q,k,v,o (3 145 728 trainable factor elements), frozen 12-layer causal text tower (77 tokens), InfoNCE, clip+AdamW; random-init weights",
                      "lora_layers": n, "batch_per_gpu": B, "global_batch": B * world, "image": "3x224x224", "parallelism": f"dp{world}",
                      "gflop_per_pair_algorithmic": GF, "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
           "loss": round(final_loss, 5), "roofline": roof}
    out.update(dist_fields(world, per_rank, args.steps, ops))
    out["cpu_baseline"] = None
    if cpu_state is not None:
        from oracle import losses_ref, text_ref, vit_ref
        bs = 2
        im, tk = images[:bs], ids[:bs]

        def fn():
            leaves = {k: cpu_state[k].clone().requires_grad_(True) for k in trainable_names}
            Pq = dict(cpu_state)
            Pq.update(leaves)
            fi = vit_ref.openai_vit_forward(im, Pq, heads=16, lora=dict(r=16, alpha=32))
            with torch.no_grad():
                ft = text_ref.openai_text_forward(tk, Pq, heads=12)
            losses_ref.info_nce(fi, ft, 0.07).backward()
        out["cpu_baseline"] = _cpu_sample(fn, bs, 2, max(1, min(32, _cpu_share())), "oracle/vit_ref.py + text_ref.py (ViT-L/14 + LoRA r=16)", budget=25.0)
    return out


if __name__ == "__main__":
    main()
