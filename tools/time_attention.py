"""Times uia_attn_fwd / uia_attn_bwd at the step's shapes: ViT-B (B 256, H 12, L 197, no mask), BERT (B 256, H 12, L 256, key padding 24-128)
and ViT-L/14 (L 257); the backward in every kernel configuration of uia_attn_bwd_cfg, interleaved rounds in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops

CFGS = [int(c) for c in os.environ.get("ATTN_CFGS", "1,2,5,7").split(",")]


def timed(f, n=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


SHAPES = os.environ.get("UIA_ATTN_SHAPES", "vit,vit-kb,bert,vit-l").split(",")
for name, L, mask, kb in (("vit", 197, None, False), ("vit-kb", 197, None, True), ("bert", 256, "keypad", False), ("vit-l", 257, None, False)):
    if name not in SHAPES:
        continue
    B, H, D = 256, 12, 768
    torch.manual_seed(0)
    qkv = (torch.randn(B * L, 3 * D, device="cuda") * 0.5).bfloat16()
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    out = ops.kb_empty(B * L, D, torch.bfloat16, "cuda") if kb else torch.empty(B * L, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, L, device="cuda")
    keylen = torch.randint(24, 129, (B,), device="cuda", dtype=torch.int32) if mask else None
    fwd = lambda: ops.attn_fwd(q, k, v, out, B, H, L, lse=lse, mask=mask, keylen=keylen)
    for _ in range(3):
        fwd()
    tf = timed(fwd)
    do = torch.randn(B * L, D, device="cuda").bfloat16()
    if kb:
        dkb = ops.kb_empty(B * L, 3 * D, torch.bfloat16, "cuda")
        dst = (dkb, None, None)
    else:
        dqkv = torch.empty_like(qkv)
        dst = (dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:])
    fl = 4.0 * B * H * L * L * 64
    times = {c: [] for c in CFGS}
    cfgs = [c for c in CFGS if not (c == 5 and L > 240)]
    for rnd in range(5):
        for c in cfgs:
            f = lambda: ops.attn_bwd(q, k, v, out, do, lse, *dst, B, H, L, mask=mask, keylen=keylen, cfg=c)
            if rnd == 0:
                f(); f()
            times[c].append(timed(f))
    print(f"{name:6s} L={L}: fwd {tf:7.1f} us ({fl / tf * 1e-6:6.0f} TF/s dense-equivalent)")
    for c in cfgs:
        t = sorted(times[c])
        print(f"        bwd cfg {c}: median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f}  ({2.5 * fl / t[len(t) // 2] * 1e-6:6.0f} TF/s on five products)")
