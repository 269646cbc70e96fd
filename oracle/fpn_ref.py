"""Feature-pyramid task heads (TimmCLIPAdapter) — functional CPU restatement.  Test infrastructure only.

Follows /root/reference/src/third_party/timm/clip_adapter.py: extract_vit_features :59-116 (tokens after the blocks in
`extract_layers`), forward :118-160 (deep-to-shallow: drop CLS, reduce Linear, LayerNorm → Linear → GELU → Linear, summed;
[B, C, g, g]; seg head = bilinear Upsample(img_size, align_corners=False) then Conv1×1, cls head = global average pool →
Dropout(0.5) → Linear).  PINNED by tests/golden/fpn_adapter.npz, generated from the imported reference class over a
torch trunk with formula-filled weights (oracle/gen_golden.py::gen_fpn).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import vit_ref


def fill(shape, a, b, scale=1.0, fn=np.sin):
    n = int(np.prod(shape))
    return torch.from_numpy((scale * fn(a * np.arange(n, dtype=np.float64) + b)).astype(np.float32).reshape(shape))


def toy_trunk_params(D=768, depth=3, hidden=192, patch=8, img=32, prefix="visual.trunk.", seed=20260315):
    """Seeded timm-style trunk (the reference hard-codes feature_dim = 768, clip_adapter.py:28): regenerated identically by the
    fixture generator and by the tests, so its 8 M weights never need to be stored.  numpy's RandomState (MT19937 +
    standard_normal) is bit-reproducible across platforms and versions; Gaussian matrices keep the toy well conditioned
    (a sin-of-index fill gave near-low-rank weights whose bf16 gradients were dominated by cancellation)."""
    rs = np.random.RandomState(seed)
    nrm = lambda shape, std: torch.from_numpy((rs.standard_normal(int(np.prod(shape))) * std).astype(np.float32).reshape(shape))
    n = (img // patch) ** 2
    P = {prefix + "patch_embed.proj.weight": nrm((D, 3, patch, patch), 0.05), prefix + "patch_embed.proj.bias": nrm((D,), 0.02),
         prefix + "cls_token": nrm((1, 1, D), 0.1), prefix + "pos_embed": nrm((1, n + 1, D), 0.1),
         prefix + "norm.weight": 1.0 + nrm((D,), 0.1), prefix + "norm.bias": nrm((D,), 0.05)}
    for i in range(depth):
        b = f"{prefix}blocks.{i}."
        P[b + "norm1.weight"] = 1.0 + nrm((D,), 0.1); P[b + "norm1.bias"] = nrm((D,), 0.05)
        P[b + "attn.qkv.weight"] = nrm((3 * D, D), 0.03); P[b + "attn.qkv.bias"] = nrm((3 * D,), 0.02)
        P[b + "attn.proj.weight"] = nrm((D, D), 0.03); P[b + "attn.proj.bias"] = nrm((D,), 0.02)
        P[b + "norm2.weight"] = 1.0 + nrm((D,), 0.1); P[b + "norm2.bias"] = nrm((D,), 0.05)
        P[b + "mlp.fc1.weight"] = nrm((hidden, D), 0.03); P[b + "mlp.fc1.bias"] = nrm((hidden,), 0.02)
        P[b + "mlp.fc2.weight"] = nrm((D, hidden), 0.03); P[b + "mlp.fc2.bias"] = nrm((D,), 0.02)
    return P


def adapter_forward(images, P, A, task="seg", extract_layers=(0, 1, 2), heads=12, img_size=32, mona=None, drop_mask=None):
    """images [B,3,H,W]; P: trunk state dict (open_clip key names); A: adapter state dict (reduces.{i}.*, blocks.{i}.{0,1,3}.*,
    seg_head.1.*, cls_head.3.*).  drop_mask: optional [B, C] keep mask (already scaled semantics: kept entries ×2) for the cls head."""
    taps = {"layers": list(extract_layers), "acts": []}
    vit_ref.timm_vit_forward(images, P, heads=heads, mona=mona, taps=taps, return_tokens=True)
    B = images.shape[0]
    a = None
    L = len(extract_layers)
    for lvl in range(L - 1, -1, -1):                                   # deep → shallow (:121-124)
        act = taps["acts"][lvl][:, 1:, :]
        r = F.linear(act, A[f"reduces.{lvl}.weight"], A[f"reduces.{lvl}.bias"])
        C = r.shape[-1]
        h = F.layer_norm(r, (C,), A[f"blocks.{lvl}.0.weight"], A[f"blocks.{lvl}.0.bias"], 1e-5)
        h = F.gelu(F.linear(h, A[f"blocks.{lvl}.1.weight"], A[f"blocks.{lvl}.1.bias"]))
        h = F.linear(h, A[f"blocks.{lvl}.3.weight"], A[f"blocks.{lvl}.3.bias"])
        a = h if a is None else h + a
    g = int(math.sqrt(a.shape[1]))
    a = a.permute(0, 2, 1).reshape(B, -1, g, g)
    if task == "seg":
        up = F.interpolate(a, size=(img_size, img_size), mode="bilinear", align_corners=False)
        return F.conv2d(up, A["seg_head.1.weight"], A["seg_head.1.bias"])
    pooled = a.mean(dim=(2, 3))
    if drop_mask is not None:
        pooled = pooled * drop_mask * 2.0                              # Dropout(0.5), inverted scaling
    return F.linear(pooled, A["cls_head.3.weight"], A["cls_head.3.bias"])


def openai_adapter_forward(images, P, A, task="seg", extract_layers=(0, 1), heads=2, img_size=32, mona=None):
    """CLIPAdapter over the OpenAI-layout tower — /root/reference/src/third_party/openai_clip/clip_adapter.py: extract_vit_features :60-90 (outputs of the
    resblocks in `extract_layers`), forward :92-136 (as the timm class above), cls head :51-58 = pool -> Linear -> ReLU -> Dropout(0.1) -> Linear (eval: no
    dropout).  P: CLIP state dict ("visual." keys); A: adapter state dict (reduces.{i}.*, blocks.{i}.{0,1,3}.*, seg_head.1.*, cls_head.{2,5}.*).
    PINNED by tests/golden/clip_adapter_openai.npz, generated from the imported reference class (oracle/gen_golden_r05.py)."""
    _, taps = vit_ref.openai_vit_forward(images, P, heads=heads, mona=mona, taps=list(extract_layers))
    B = images.shape[0]
    a = None
    for lvl in range(len(extract_layers) - 1, -1, -1):
        act = taps[lvl][:, 1:, :]
        r = F.linear(act, A[f"reduces.{lvl}.weight"], A[f"reduces.{lvl}.bias"])
        C = r.shape[-1]
        h = F.layer_norm(r, (C,), A[f"blocks.{lvl}.0.weight"], A[f"blocks.{lvl}.0.bias"], 1e-5)
        h = F.gelu(F.linear(h, A[f"blocks.{lvl}.1.weight"], A[f"blocks.{lvl}.1.bias"]))
        h = F.linear(h, A[f"blocks.{lvl}.3.weight"], A[f"blocks.{lvl}.3.bias"])
        a = h if a is None else h + a
    g = int(math.sqrt(a.shape[1]))
    a = a.permute(0, 2, 1).reshape(B, -1, g, g)
    if task == "seg":
        up = F.interpolate(a, size=(img_size, img_size), mode="bilinear", align_corners=False)
        return F.conv2d(up, A["seg_head.1.weight"], A["seg_head.1.bias"])
    pooled = a.mean(dim=(2, 3))
    h = F.relu(F.linear(pooled, A["cls_head.2.weight"], A["cls_head.2.bias"]))
    return F.linear(h, A["cls_head.5.weight"], A["cls_head.5.bias"])
