"""Mona adapter variants backed by the MI355X HIP kernels (drop-in for the reference module of
the same name: /root/reference/src/adapters/mona.py).

Same public surface — class names, constructor signatures ``Cls(in_dim, bottleneck_dim=64)``,
``forward(x, hw_shapes=None)`` on sequence-first ``[N, B, D]`` input, parameter / state-dict names
(``project1``, ``project2``, ``adapter_conv.{conv1,conv2,conv3,projector,freq_filter,noise_estimator.1,
noise_estimator.3}``, ``norm``, ``gamma``, ``gammax``), ``BatchFirstMonaWrapper.clip_mona`` and the two
injectors with their return tuples — but the arithmetic runs in ``libuia_hip.so``
(``uia_hip.functional.MonaFn``): LN·γ + x·γx → project1 → fused per-image spatial op → GELU → dropout →
project2 + residual, forward and backward.  There is no PyTorch fallback: CPU tensors raise.

Parameters are held by small holder modules so that ``named_parameters()`` yields the reference's
names in the reference's order (checkpoints are matched by name: src/models/biomedclip/zero_shot.py:139-147).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from uia_hip import functional as UF

__all__ = ["BatchFirstMonaWrapper", "BaselineMona", "BaselineMonaOp", "NoiseAwareMona", "NoiseAwareMonaOp", "FreqEnhancedMona",
           "FreqEnhancedMonaOp", "HybridNoiseFreqMona", "HybridNoiseFreqMonaOp", "inject_mona_variant_to_clip",
           "inject_mona_variant_to_open_clip"]


class BatchFirstMonaWrapper(nn.Module):
    """[B,N,D] callers (open_clip-style towers) around a sequence-first Mona (reference :38-67).
    The two permutes cancel inside the fused op, so no data moves."""

    def __init__(self, mona_adapter):
        super().__init__()
        self.clip_mona = mona_adapter

    def forward(self, x, hw_shapes=None):
        return self.clip_mona(x.permute(1, 0, 2), hw_shapes).permute(1, 0, 2)


class _MonaOpParams(nn.Module):
    """Parameter holder for the spatial operator (names/shapes/initialisation of reference :78-83,
    :162-176, :271-277, :384-399).  It has no forward of its own on the product path: the operator is
    fused into the adapter kernel."""

    has_freq = False
    has_noise = False

    def __init__(self, in_features):
        super().__init__()
        c = in_features
        self.conv1 = nn.Conv2d(c, c, kernel_size=3, padding=1, groups=c)
        self.conv2 = nn.Conv2d(c, c, kernel_size=5, padding=2, groups=c)
        self.conv3 = nn.Conv2d(c, c, kernel_size=7, padding=3, groups=c)
        self.projector = nn.Conv2d(c, c, kernel_size=1)
        if self.has_freq:
            self.freq_filter = nn.Parameter(torch.ones(c))
        if self.has_noise:
            self.noise_estimator = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(c, c // 4, 1), nn.ReLU(inplace=True),
                                                 nn.Conv2d(c // 4, 3, 1), nn.Softmax(dim=1))

    def forward(self, x):
        raise RuntimeError("the Mona spatial operator only runs fused inside its adapter (libuia_hip.so); "
                           "call the enclosing *Mona module")


class BaselineMonaOp(_MonaOpParams):
    pass


class NoiseAwareMonaOp(_MonaOpParams):
    has_noise = True


class FreqEnhancedMonaOp(_MonaOpParams):
    has_freq = True


class HybridNoiseFreqMonaOp(_MonaOpParams):
    has_freq = True
    has_noise = True


class _MonaBase(nn.Module):
    variant = None
    op_cls = None

    def __init__(self, in_dim, bottleneck_dim=64):
        super().__init__()
        self.project1 = nn.Linear(in_dim, bottleneck_dim)
        self.project2 = nn.Linear(bottleneck_dim, in_dim)
        self.dropout = nn.Dropout(p=0.1)
        self.adapter_conv = self.op_cls(bottleneck_dim)
        self.norm = nn.LayerNorm(in_dim)
        self.gamma = nn.Parameter(torch.ones(in_dim) * 1e-6)
        self.gammax = nn.Parameter(torch.ones(in_dim))
        self.keep_mask = None      # optional uint8 [B,N,bottleneck] override of the dropout draw (parity tests)

    def _params(self):
        return {k: p for k, p in self.named_parameters()}

    def forward(self, x, hw_shapes=None):
        """x: [N, B, D] sequence-first (reference :115-151); returns [N, B, D]."""
        xb = x.permute(1, 0, 2)
        n = xb.shape[1]
        if hw_shapes is None:
            # reference :140-144: no CLS token — ALL n tokens form an int(sqrt(n))-square grid and go through the spatial operator.
            # The fused kernels are laid out for [CLS ; h*w tokens], so a zero row stands in for the CLS slot and is dropped from
            # the result: its output is never used, hence its upstream gradient is zero and it adds exactly nothing to any
            # parameter gradient.  (Neither injector takes this path; it serves direct callers of the module.)
            h = w = int(math.isqrt(n))
            if h * w != n:
                raise RuntimeError(f"shape '[{xb.shape[0]}, {h}, {w}, {self.project1.out_features}]' is invalid for {n} tokens: "
                                   "hw_shapes=None needs a square token grid (reference mona.py:141-142)")
            mask = self.keep_mask
            if mask is not None:
                mask = torch.cat([mask.new_ones(mask.shape[0], 1, mask.shape[2]), mask], dim=1).contiguous()
            xp = torch.cat([xb.new_zeros(xb.shape[0], 1, xb.shape[2]), xb], dim=1)
            y = UF.mona_apply(xp.contiguous(), self._params(), self.variant, (h, w), self.dropout.p, self.training, mask)
            return y[:, 1:, :].permute(1, 0, 2)
        y = UF.mona_apply(xb.contiguous(), self._params(), self.variant, hw_shapes, self.dropout.p, self.training, self.keep_mask)
        return y.permute(1, 0, 2)


class BaselineMona(_MonaBase):
    variant, op_cls = "baseline", BaselineMonaOp


class NoiseAwareMona(_MonaBase):
    variant, op_cls = "noise_aware", NoiseAwareMonaOp


class FreqEnhancedMona(_MonaBase):
    variant, op_cls = "freq_enhanced", FreqEnhancedMonaOp


class HybridNoiseFreqMona(_MonaBase):
    variant, op_cls = "hybrid", HybridNoiseFreqMonaOp


_VARIANTS = {"baseline": BaselineMona, "noise_aware": NoiseAwareMona, "freq_enhanced": FreqEnhancedMona, "hybrid": HybridNoiseFreqMona}


def _variant_class(variant):
    if variant not in _VARIANTS:      # includes the reference's CLI choice "fractional", which it also rejects (:520-521)
        raise ValueError(f"Unknown variant: {variant}. Choose from {list(_VARIANTS.keys())}")
    return _VARIANTS[variant]


def _attach(blocks, count, make_adapter, hw_shapes):
    """Store the adapter on the block and wrap the INSTANCE's forward (reference :556-571 / :661-676)."""
    for i in range(count):
        block = blocks[i]
        block.mona = make_adapter()
        orig_forward = block.forward

        def forward_with_mona(x, _orig=orig_forward, _block=block, **kwargs):
            return _block.mona(_orig(x, **kwargs), hw_shapes)

        block.forward = forward_with_mona
    return count


def inject_mona_variant_to_clip(model, variant="hybrid", bottleneck_dim=64, num_layers=None):
    """OpenAI-CLIP layout (sequence-first blocks under model.visual.transformer.resblocks); reference :495-575."""
    cls = _variant_class(variant)
    count = 0
    visual = getattr(model, "visual", None)
    if visual is not None and hasattr(visual, "transformer"):
        tr = visual.transformer
        if hasattr(visual, "input_resolution"):
            res = visual.input_resolution
        elif hasattr(visual, "image_size"):
            res = visual.image_size[0] if isinstance(visual.image_size, tuple) else visual.image_size
        else:
            raise AttributeError("Model does not have input_resolution or image_size attribute")
        grid = res // visual.conv1.kernel_size[0]
        if hasattr(tr, "resblocks"):
            n = len(tr.resblocks) if num_layers is None else min(num_layers, len(tr.resblocks))
            dev = visual.conv1.weight.device
            count = _attach(tr.resblocks, n, lambda: cls(tr.width, bottleneck_dim).to(dev), (grid, grid))
    print(f"✓ Injected {variant} MONA adapters to {count} layers (OpenAI CLIP vision encoder)")
    return model, count


def inject_mona_variant_to_open_clip(model, variant="hybrid", bottleneck_dim=64, num_layers=None):
    """open_clip layouts: visual.trunk.blocks (BiomedCLIP / UniMed) or visual.transformer.resblocks (MetaCLIP); reference :578-680."""
    cls = _variant_class(variant)
    count = 0
    visual = getattr(model, "visual", None)
    if visual is not None:
        blocks = dim = hw = None
        if hasattr(visual, "trunk"):
            trunk = visual.trunk
            dim = trunk.embed_dim
            g = int(np.sqrt(trunk.patch_embed.num_patches))
            hw = (g, g)
            blocks = getattr(trunk, "blocks", None)
        elif hasattr(visual, "transformer"):
            tr = visual.transformer
            dim = tr.width
            if hasattr(visual, "grid_size"):
                hw = (visual.grid_size[0], visual.grid_size[0])
            elif hasattr(visual, "image_size") and hasattr(visual, "patch_size"):
                img = visual.image_size[0] if isinstance(visual.image_size, tuple) else visual.image_size
                pat = visual.patch_size[0] if isinstance(visual.patch_size, tuple) else visual.patch_size
                hw = (img // pat, img // pat)
            blocks = getattr(tr, "resblocks", None)
        if blocks is not None and dim is not None:
            n = len(blocks) if num_layers is None else min(num_layers, len(blocks))
            dev = next(visual.parameters()).device
            count = _attach(blocks, n, lambda: BatchFirstMonaWrapper(cls(dim, bottleneck_dim)).to(dev), hw)
    print(f"✓ Injected {variant} MONA adapters to {count} layers (open_clip vision encoder)")
    return model, count
