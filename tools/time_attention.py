"""Times uia_attn_fwd / uia_attn_bwd at the step's shapes: ViT-B (B 256, H 12, L 197, no mask) and BERT (B 256, H 12, L 256, key padding 24-128)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, L, mask in (("vit", 197, None), ("bert", 256, "keypad"), ("vit-l", 257, None)):
    B, H, D = 256, 12, 768
    torch.manual_seed(0)
    qkv = (torch.randn(B * L, 3 * D, device="cuda") * 0.5).bfloat16()
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    out = torch.empty(B * L, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, L, device="cuda")
    keylen = torch.randint(24, 129, (B,), device="cuda", dtype=torch.int32) if mask else None
    tf = timed(lambda: ops.attn_fwd(q, k, v, out, B, H, L, lse=lse, mask=mask, keylen=keylen))
    do = torch.randn_like(out)
    dqkv = torch.empty_like(qkv)
    tb = timed(lambda: ops.attn_bwd(q, k, v, out, do, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, L, mask=mask, keylen=keylen))
    fl = 4.0 * B * H * L * L * 64
    print(f"{name:6s} L={L}: fwd {tf:7.1f} us ({fl / tf * 1e-6:6.0f} TF/s dense-equivalent)   bwd {tb:7.1f} us ({2.5 * fl / tb * 1e-6:6.0f} TF/s)")
