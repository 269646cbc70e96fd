"""LoRA on the HIP path (drop-in for /root/reference/src/adapters/lora.py).

Same surface: ``LoRALayer`` (scaling = alpha / sqrt(r), :20-21), ``LinearLoRA(existing_linear, r, lora_alpha,
dropout_rate)`` with parameters ``weight``, ``bias``, ``w_lora_A [r,in]``, ``w_lora_B [out,r]`` (:29-36, A kaiming-uniform
a=√5, B zeros :39-44), ``PlainMultiheadAttentionLoRA(existing_mha, enable_lora, r, lora_alpha, dropout_rate)`` with
``q_proj / k_proj / v_proj / proj`` (:115-153) returning ``(out, None)``, and the two injectors with their
``(model, count)`` return.  Quirk kept: the copied ``bias`` of a LinearLoRA stays trainable (:64-65; SURVEY C-4).

Arithmetic: ``uia_hip.functional.LoraLinearFn`` evaluates y = x·Wᵀ + b + s·drop(x)·Aᵀ·Bᵀ in RANK form on the MFMA GEMM
(the reference materialises B·A and runs a second full-size GEMM, :46-51,:87); results are identical up to
fp32 summation order.
"""
import math

import torch
import torch.nn as nn

from uia_hip import functional as UF
from uia_hip import ops

__all__ = ["LoRALayer", "LinearLoRA", "PlainMultiheadAttentionLoRA", "inject_lora_to_clip", "inject_lora_to_biomedclip"]


class LoRALayer:
    def __init__(self, r, lora_alpha, dropout_rate=0):
        self.r, self.lora_alpha, self.dropout_rate = r, lora_alpha, dropout_rate
        if self.r > 0:
            self.scaling = self.lora_alpha / math.sqrt(self.r)
        self.merged = False
        self.params_with_lora = {}

    def register_lora_param(self):
        for pname, lname in self.params_with_lora.items():
            w = getattr(self, pname)
            assert w.dim() == 2
            self.register_parameter(f"{lname}_lora_A", nn.Parameter(w.new_zeros((self.r, w.shape[1]))))
            self.register_parameter(f"{lname}_lora_B", nn.Parameter(w.new_zeros((w.shape[0], self.r))))
            w.requires_grad = False

    def init_lora_param(self):
        for pname, lname in self.params_with_lora.items():
            if hasattr(self, f"{lname}_lora_A"):
                nn.init.kaiming_uniform_(getattr(self, f"{lname}_lora_A"), a=math.sqrt(5))
                nn.init.zeros_(getattr(self, f"{lname}_lora_B"))

    def merge_BA(self, param_name):
        lname = self.params_with_lora[param_name]
        return (getattr(self, f"{lname}_lora_B") @ getattr(self, f"{lname}_lora_A")).view(getattr(self, param_name).shape)


class LinearLoRA(nn.Linear, LoRALayer):
    def __init__(self, existing_linear, r=0, lora_alpha=1, dropout_rate=0.0):
        super().__init__(in_features=existing_linear.in_features, out_features=existing_linear.out_features,
                         bias=existing_linear.bias is not None, device=existing_linear.weight.device)
        self.load_state_dict(existing_linear.state_dict())
        LoRALayer.__init__(self, r=r, lora_alpha=lora_alpha, dropout_rate=dropout_rate)
        self.params_with_lora = {"weight": "w"}
        if r > 0:
            self.register_lora_param()
        self.init_lora_param()
        self.dropout = nn.Dropout(dropout_rate) if dropout_rate > 0 else None

    def _drop_p(self):
        return self.dropout.p if (self.training and self.dropout is not None and self.dropout.p > 0) else 0.0

    def apply_rows(self, x2d, resid32=None):
        """x2d: [M, in] in the compute dtype → [M, out] (T, or fp32 when the fp32 residual is fused)."""
        if self.r > 0:
            trainable = [p for p in (self.w_lora_A, self.w_lora_B, self.bias) if p is not None and p.requires_grad]
            direct = torch.is_grad_enabled() and len(trainable) >= 2 and all(UF._is_flat_grad(p) for p in trainable)
            return UF.LoraLinearFn.apply(x2d, self.weight, self.bias, self.w_lora_A, self.w_lora_B, self.scaling, self._drop_p(), resid32, direct)
        empty = x2d.new_zeros(0, x2d.shape[1], dtype=torch.float32)
        return UF.LoraLinearFn.apply(x2d, self.weight, self.bias, empty, empty.new_zeros(self.out_features, 0), 0.0, 0.0, resid32)

    def forward(self, x):
        """Generic entry: any leading shape, fp32 or compute-dtype input; output has the input's dtype."""
        dt = UF.compute_dtype()
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != dt:
            x2 = _ToCompute.apply(x2.contiguous())
        y = self.apply_rows(x2.contiguous())
        if x.dtype != dt:
            y = _FromCompute.apply(y, x.dtype)
        return y.reshape(*lead, self.out_features)


class _ToCompute(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        out = torch.empty(x.shape, device=x.device, dtype=UF.compute_dtype())
        ops.cast(x.float(), out)
        return out

    @staticmethod
    def backward(ctx, g):
        return g.float()


class _FromCompute(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = x.dtype
        return x.to(dt)

    @staticmethod
    def backward(ctx, g):
        out = torch.empty(g.shape, device=g.device, dtype=ctx.dt)
        ops.cast(g.contiguous().float(), out)
        return out, None


class PlainMultiheadAttentionLoRA(nn.Module):
    def __init__(self, existing_mha, enable_lora=("q", "k", "v", "o"), r=0, lora_alpha=1, dropout_rate=0.0):
        super().__init__()
        self.dropout = 0
        self.embed_dim, self.kdim, self.vdim = existing_mha.embed_dim, existing_mha.kdim, existing_mha.vdim
        self._qkv_same_embed_dim = existing_mha._qkv_same_embed_dim
        self.num_heads, self.batch_first, self.head_dim = existing_mha.num_heads, existing_mha.batch_first, existing_mha.head_dim
        has_b, dev, D = existing_mha.in_proj_bias is not None, existing_mha.in_proj_weight.device, self.embed_dim
        self.q_proj, self.k_proj, self.v_proj = (nn.Linear(D, D, bias=has_b, device=dev) for _ in range(3))
        self.proj = nn.Linear(D, D, bias=existing_mha.out_proj.bias is not None, device=dev)
        with torch.no_grad():
            w, b = existing_mha.in_proj_weight.data, (existing_mha.in_proj_bias.data if has_b else None)
            for i, lin in enumerate((self.q_proj, self.k_proj, self.v_proj)):
                lin.weight.copy_(w[i * D:(i + 1) * D])
                if b is not None:
                    lin.bias.copy_(b[i * D:(i + 1) * D])
            self.proj.weight.copy_(existing_mha.out_proj.weight.data)
            if self.proj.bias is not None:
                self.proj.bias.copy_(existing_mha.out_proj.bias.data)
        for item, name in (("q", "q_proj"), ("k", "k_proj"), ("v", "v_proj"), ("o", "proj")):
            if item in enable_lora:
                setattr(self, name, LinearLoRA(getattr(self, name), r=r, lora_alpha=lora_alpha, dropout_rate=dropout_rate))
            else:
                setattr(self, name, LinearLoRA(getattr(self, name), r=0))

    def rows_forward(self, h2d, B, L, mask, resid32=None):
        """h2d: [B*L, D] compute-dtype rows in (b, l) order → attention output rows (fp32 if resid32 is fused)."""
        q, k, v = self.q_proj.apply_rows(h2d), self.k_proj.apply_rows(h2d), self.v_proj.apply_rows(h2d)
        a = UF.AttentionFn.apply(q, k, v, B, self.num_heads, L, mask)
        return self.proj.apply_rows(a, resid32)

    def block_half(self, xb, ln, B, L, mask):
        """x + proj(attention(LN x)) for the fp32 residual stream xb [B, L, D] of an OpenAI-CLIP block whose attention this module replaced
        (model.py:195-201): one autograd node when q, k, v and the output projection all carry LoRA factors of one rank, scaling and dropout
        (how inject_lora_to_clip builds them); the composition of rows_forward otherwise."""
        ps = (self.q_proj, self.k_proj, self.v_proj, self.proj)
        # ... and a frozen LayerNorm: the fused node returns no gradient for ln.weight / ln.bias (a user who unfreezes the norms gets the composed path)
        uniform = (all(m.r > 0 for m in ps) and len({(m.r, m.scaling, m._drop_p()) for m in ps}) == 1
                   and len({m.bias is None for m in ps}) == 1 and self.head_dim == 64
                   and not ln.weight.requires_grad and not ln.bias.requires_grad)
        if not uniform:
            D = xb.shape[-1]
            h = UF.LayerNormFn.apply(xb, ln.weight, ln.bias, ln.eps).view(B * L, D)
            return self.rows_forward(h, B, L, mask, resid32=xb.view(B * L, D)).view(B, L, D)
        trainable = [p for m in ps for p in (m.w_lora_A, m.w_lora_B, m.bias) if p is not None and p.requires_grad]
        direct = (torch.is_grad_enabled() and all(m.w_lora_A.requires_grad and m.w_lora_B.requires_grad for m in ps)
                  and all(UF._is_flat_grad(p) for p in trainable))
        flat = [t for m in ps for t in (m.weight, m.bias, m.w_lora_A, m.w_lora_B)]
        return UF.LoraAttnHalfFn.apply(xb, ln.weight, ln.bias, ln.eps, self.num_heads, mask, ps[0].scaling, ps[0]._drop_p(), direct, *flat)

    def forward(self, query, key, value, key_padding_mask=None, need_weights=False, attn_mask=None, **kwargs):
        if not (query is key and key is value):
            raise NotImplementedError("PlainMultiheadAttentionLoRA on the HIP path supports self-attention only (how the reference uses it, model.py:197)")
        if key_padding_mask is not None:
            raise NotImplementedError("key_padding_mask is unused on the reference path")
        x = query if self.batch_first else query.transpose(0, 1)          # → [B, L, D]
        B, L, D = x.shape
        mask = _mask_kind(attn_mask, L)
        dt = UF.compute_dtype()
        x2 = x.contiguous().view(B * L, D)
        in_dtype = x2.dtype
        if in_dtype != dt:
            x2 = _ToCompute.apply(x2)
        out = self.rows_forward(x2, B, L, mask)
        if in_dtype != dt:
            out = _FromCompute.apply(out, in_dtype)
        out = out.view(B, L, D)
        return (out if self.batch_first else out.transpose(0, 1)), None


def _mask_kind(attn_mask, L):
    if attn_mask is None:
        return None
    if attn_mask.shape == (L, L) and torch.isinf(attn_mask[0, -1]) and float(attn_mask[-1, 0]) == 0.0:
        return "causal"
    raise NotImplementedError("only the causal additive mask of model.py:346-352 is supported")


def lora_block_forward(block, x):
    """timm-style Block whose attn.qkv / attn.proj were replaced by LinearLoRA (inject_lora_to_biomedclip)."""
    B, N, D = x.shape
    attn = block.attn
    h = UF.LayerNormFn.apply(x, block.norm1.weight, block.norm1.bias, block.norm1.eps).view(B * N, D)
    qkv = attn.qkv.apply_rows(h) if isinstance(attn.qkv, LinearLoRA) else LinearLoRA(attn.qkv, r=0).apply_rows(h)
    a = UF.AttentionFn.apply(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, attn.num_heads, N, None)
    x2d = x.contiguous().view(B * N, D)
    proj = attn.proj if isinstance(attn.proj, LinearLoRA) else LinearLoRA(attn.proj, r=0)
    x1 = proj.apply_rows(a, x2d).view(B, N, D)
    spec = UF.BlockSpec(attn.num_heads, block.norm1.eps, "gelu", None, None, None, (block.norm2.weight, block.norm2.bias),
                        (block.mlp.fc1.weight, block.mlp.fc1.bias), (block.mlp.fc2.weight, block.mlp.fc2.bias))
    return UF.MlpHalfFn.apply(x1, spec)


def inject_lora_to_clip(model, lora_r=16, lora_alpha=32, lora_dropout=0.1, num_layers=None):
    """Replace nn.MultiheadAttention of the first `num_layers` vision resblocks (reference :202-248)."""
    count = 0
    visual = getattr(model, "visual", None)
    if visual is not None and hasattr(visual, "transformer") and hasattr(visual.transformer, "resblocks"):
        blocks = visual.transformer.resblocks
        n = len(blocks) if num_layers is None else min(num_layers, len(blocks))
        for i in range(n):
            blk = blocks[i]
            if hasattr(blk, "attn") and isinstance(blk.attn, nn.MultiheadAttention):
                blk.attn = PlainMultiheadAttentionLoRA(blk.attn, enable_lora=["q", "k", "v", "o"], r=lora_r, lora_alpha=lora_alpha,
                                                       dropout_rate=lora_dropout)
                count += 1
    print(f"✓ Injected LoRA adapters to {count} layers (CLIP vision encoder)")
    return model, count


def inject_lora_to_biomedclip(model, lora_r=16, lora_alpha=32, lora_dropout=0.1, num_layers=None, tune_text_encoder=False):
    """Replace attn.qkv / attn.proj of the timm trunk blocks (and optionally BERT q/k/v/o) (reference :251-370)."""
    count = 0
    visual = getattr(model, "visual", None)
    if visual is not None and hasattr(visual, "trunk") and hasattr(visual.trunk, "blocks"):
        blocks = visual.trunk.blocks
        n = len(blocks) if num_layers is None else min(num_layers, len(blocks))
        for i in range(n):
            attn = getattr(blocks[i], "attn", None)
            if attn is None:
                continue
            for name in ("qkv", "proj"):
                lin = getattr(attn, name, None)
                if isinstance(lin, nn.Linear):
                    setattr(attn, name, LinearLoRA(lin, r=lora_r, lora_alpha=lora_alpha, dropout_rate=lora_dropout))
            count += 1
    if tune_text_encoder:
        text = getattr(model, "text", None)
        tr = getattr(text, "transformer", None)
        layers = getattr(getattr(tr, "encoder", None), "layer", None)
        if layers is not None:
            n = len(layers) if num_layers is None else min(num_layers, len(layers))
            for i in range(n):
                att = getattr(layers[i], "attention", None)
                sa = getattr(att, "self", None)
                if sa is None:
                    continue
                for name in ("query", "key", "value"):
                    lin = getattr(sa, name, None)
                    if isinstance(lin, nn.Linear):
                        setattr(sa, name, LinearLoRA(lin, r=lora_r, lora_alpha=lora_alpha, dropout_rate=lora_dropout))
                dense = getattr(getattr(att, "output", None), "dense", None)
                if isinstance(dense, nn.Linear):
                    att.output.dense = LinearLoRA(dense, r=lora_r, lora_alpha=lora_alpha, dropout_rate=lora_dropout)
                count += 1
    print(f"✓ Injected LoRA adapters to {count} layers (BiomedCLIP vision encoder)")
    return model, count
