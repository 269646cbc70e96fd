"""Image towers — functional CPU restatement.  Test infrastructure only.

(1) timm_vit_forward: the BiomedCLIP image tower = open_clip 3.2.0 ``TimmModel`` around timm
    1.0.20 ``vit_base_patch16_224`` (third-party, absent from /root/reference; call sites
    src/models/biomedclip/finetune.py:116-119,276; attributes used by src/adapters/mona.py:620-630
    and src/adapters/lora.py:284-313).  Published arithmetic restated per SURVEY Appendix A.1:
    Conv2d(3,D,k16,s16,bias) patch embed; cat(cls)+pos_embed; pre-LN blocks (eps 1e-6), fused qkv,
    softmax(q kᵀ d^-½) v, exact-erf GELU MLP; final LN; CLS pool; bias-free head.proj.
    PARITY UNPINNED BY THE REFERENCE — cross-checked against the installed transformers ViTModel
    (tests/golden/hf_vit_*.npz, oracle/gen_golden.py).
(2) openai_vit_forward: the in-tree OpenAI CLIP VisionTransformer,
    /root/reference/src/third_party/openai_clip/model.py:233-257 (block :177-202, QuickGELU :172-174,
    LayerNorm eps 1e-5 :163-169).  PINNED by golden vectors from the imported reference.

Both accept an optional Mona hook: after block i, x <- mona(x) (mona.py:562-571 / 667-676), and
an optional LoRA spec for the attention projections.
"""
import math
import torch
import torch.nn.functional as F

from . import mona_ref, lora_ref


def _sub(P, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in P.items() if k.startswith(prefix)}


def _attention(q, k, v, heads, mask=None):
    """q,k,v: [B, L, D] -> [B, L, D];  softmax(q kᵀ / sqrt(dh) + mask) v per head."""
    B, L, D = q.shape
    dh = D // heads
    q = q.view(B, L, heads, dh).transpose(1, 2)
    k = k.view(B, L, heads, dh).transpose(1, 2)
    v = v.view(B, L, heads, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2) * dh ** -0.5
    if mask is not None:
        s = s + mask
    o = torch.softmax(s, dim=-1) @ v
    return o.transpose(1, 2).reshape(B, L, D)


def _maybe_lora_linear(x, P, name, lora):
    if lora is not None and f"{name}.w_lora_A" in P:
        return lora_ref.linear_lora(x, P[f"{name}.weight"], P.get(f"{name}.bias"), P[f"{name}.w_lora_A"],
                                    P[f"{name}.w_lora_B"], lora["r"], lora["alpha"])
    return F.linear(x, P[f"{name}.weight"], P.get(f"{name}.bias"))


def timm_block(x, P, heads, eps=1e-6, lora=None, act="gelu"):
    """timm Block: x + attn(norm1 x); x + mlp(norm2 x).  P keys relative to 'blocks.{i}.'.
    act="quick_gelu": x·σ(1.702x), the `vit_*_clip_quickgelu_*` variants (same form as reference model.py:172-174)."""
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), P["norm1.weight"], P["norm1.bias"], eps)
    qkv = _maybe_lora_linear(h, P, "attn.qkv", lora)
    q, k, v = qkv.split(D, dim=-1)                      # reshape(B,N,3,H,dh): column = which*D + h*dh + d
    a = _attention(q, k, v, heads)
    x = x + _maybe_lora_linear(a, P, "attn.proj", lora)
    h = F.layer_norm(x, (D,), P["norm2.weight"], P["norm2.bias"], eps)
    h = F.linear(h, P["mlp.fc1.weight"], P["mlp.fc1.bias"])
    h = F.gelu(h) if act == "gelu" else h * torch.sigmoid(1.702 * h)
    return x + F.linear(h, P["mlp.fc2.weight"], P["mlp.fc2.bias"])


def timm_vit_tokens(images, P, prefix="visual.trunk."):
    W = P[prefix + "patch_embed.proj.weight"]
    x = F.conv2d(images, W, P.get(prefix + "patch_embed.proj.bias"), stride=W.shape[-1])
    x = x.flatten(2).transpose(1, 2)                                  # [B, 196, D]
    cls = P[prefix + "cls_token"].expand(x.shape[0], -1, -1)
    return torch.cat([cls, x], dim=1) + P[prefix + "pos_embed"]


def timm_vit_forward(images, P, heads=12, mona=None, lora=None, prefix="visual.trunk.", return_tokens=False, eps=1e-6, act="gelu", taps=None):
    """images [B,3,H,W] -> features [B, embed].  P: flat state dict with open_clip key names.

    mona: None or dict(variant=..., hw=(h,w), keep_masks=None|list, p_drop=0.1); Mona parameters
    are read from '<prefix>blocks.{i}.mona.clip_mona.<p>' (wrapper attribute, mona.py:52).
    """
    x = timm_vit_tokens(images, P, prefix)
    if prefix + "norm_pre.weight" in P:                               # timm `pre_norm=True` (vit_*_clip_* family); PARITY UNPINNED
        x = F.layer_norm(x, (x.shape[-1],), P[prefix + "norm_pre.weight"], P[prefix + "norm_pre.bias"], eps)
    depth = 1 + max(int(k[len(prefix) + 7:].split(".")[0]) for k in P if k.startswith(prefix + "blocks."))
    for i in range(depth):
        bp = _sub(P, f"{prefix}blocks.{i}.")
        x = timm_block(x, bp, heads, eps, lora, act)
        mp = _sub(bp, "mona.clip_mona.")
        if mona is not None and mp:
            km = None if mona.get("keep_masks") is None else mona["keep_masks"][i]
            x = mona_ref.forward(x, mp, mona["variant"], mona["hw"], keep_mask=km, p_drop=mona.get("p_drop", 0.1))
        if taps is not None and i in taps["layers"]:
            taps["acts"].append(x)          # tokens after block i (timm/clip_adapter.py:104-107)
    if return_tokens:
        return x
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), P[prefix + "norm.weight"], P[prefix + "norm.bias"], eps)
    head = prefix.replace("trunk.", "head.") + "proj.weight"           # visual.head.proj.weight
    return F.linear(x[:, 0], P[head])


# --------------------------------------------------------------------------- OpenAI CLIP ViT
def openai_block(x, P, heads, mask=None, lora=None):
    """ResidualAttentionBlock (model.py:199-202) on batch-first x [B,L,D]; P keys relative to the block."""
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), P["ln_1.weight"], P["ln_1.bias"], 1e-5)
    if lora is not None and "attn.q_proj.w_lora_A" in P:
        a = lora_ref.mha_lora(h.transpose(0, 1), _sub(P, "attn."), heads, lora["r"], lora["alpha"], attn_mask=mask).transpose(0, 1)
    else:
        qkv = F.linear(h, P["attn.in_proj_weight"], P["attn.in_proj_bias"])
        q, k, v = qkv.split(D, dim=-1)
        a = F.linear(_attention(q, k, v, heads, mask), P["attn.out_proj.weight"], P["attn.out_proj.bias"])
    x = x + a
    h = F.layer_norm(x, (D,), P["ln_2.weight"], P["ln_2.bias"], 1e-5)
    h = F.linear(h, P["mlp.c_fc.weight"], P["mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)                                   # QuickGELU :172-174
    return x + F.linear(h, P["mlp.c_proj.weight"], P["mlp.c_proj.bias"])


def openai_vit_forward(images, P, heads, mona=None, lora=None, prefix="visual.", taps=None):
    """model.py:233-257.  taps: iterable of block indices whose outputs are also returned
    (clipseg_adapter.py:63-68).  Mona params at '<prefix>transformer.resblocks.{i}.mona.<p>'."""
    W = P[prefix + "conv1.weight"]
    x = F.conv2d(images, W, None, stride=W.shape[-1]).flatten(2).transpose(1, 2)   # :234-236
    cls = P[prefix + "class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + P[prefix + "positional_embedding"]           # :237-245
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), P[prefix + "ln_pre.weight"], P[prefix + "ln_pre.bias"], 1e-5)
    bpfx = prefix + "transformer.resblocks."
    depth = 1 + max(int(k[len(bpfx):].split(".")[0]) for k in P if k.startswith(bpfx))
    tapped = []
    for i in range(depth):
        bp = _sub(P, f"{bpfx}{i}.")
        x = openai_block(x, bp, heads, None, lora)
        mp = _sub(bp, "mona.")
        if mona is not None and mp:
            km = None if mona.get("keep_masks") is None else mona["keep_masks"][i]
            x = mona_ref.forward(x, mp, mona["variant"], mona["hw"], keep_mask=km, p_drop=mona.get("p_drop", 0.1))
        if taps is not None and i in taps:
            tapped.append(x)
    feat = F.layer_norm(x[:, 0], (D,), P[prefix + "ln_post.weight"], P[prefix + "ln_post.bias"], 1e-5)
    feat = feat @ P[prefix + "proj"]                                               # :252-255
    return (feat, tapped) if taps is not None else feat
