"""Mona adapter (4 variants) — functional CPU restatement.  Test infrastructure only.

Follows /root/reference/src/adapters/mona.py:
  BaselineMona.forward        :115-151   (op :85-93)
  NoiseAwareMona.forward      :217-253   (op :178-195, estimator :170-176)
  FreqEnhancedMona.forward    :319-362   (op :279-295)
  HybridNoiseFreqMona.forward :451-487   (op :401-424)
  BatchFirstMonaWrapper       :54-67     (two permutes that cancel for batch-first callers)

Restated as equations; tensors are batch-first [B, N, D] here — the
reference's [N,B,D] entry layout is a pure permute.

The frequency filter is kept in its literal rfft2 / irfft2 form (mona.py:284-286) so that the
oracle does not bake in the "per-channel scale" simplification the HIP kernel relies on; the
three depth-wise convolutions are likewise evaluated separately.
"""
import math
import torch
import torch.nn.functional as F

VARIANTS = ("baseline", "noise_aware", "freq_enhanced", "hybrid")


def param_names(variant):
    """Parameter names in the reference's named_parameters() order."""
    names = ["gamma", "gammax", "project1.weight", "project1.bias", "project2.weight", "project2.bias"]
    if variant in ("freq_enhanced", "hybrid"):
        names.append("adapter_conv.freq_filter")
    for c in ("conv1", "conv2", "conv3", "projector"):
        names += [f"adapter_conv.{c}.weight", f"adapter_conv.{c}.bias"]
    if variant in ("noise_aware", "hybrid"):
        names += [f"adapter_conv.noise_estimator.{i}.{p}" for i in (1, 3) for p in ("weight", "bias")]
    names += ["norm.weight", "norm.bias"]
    return names


def param_shapes(variant, dim, bott):
    """Shapes as created at mona.py:78-83,104-113,162-176,277."""
    q = bott // 4
    shapes = {
        "gamma": (dim,), "gammax": (dim,),
        "project1.weight": (bott, dim), "project1.bias": (bott,),
        "project2.weight": (dim, bott), "project2.bias": (dim,),
        "adapter_conv.freq_filter": (bott,),
        "adapter_conv.conv1.weight": (bott, 1, 3, 3), "adapter_conv.conv1.bias": (bott,),
        "adapter_conv.conv2.weight": (bott, 1, 5, 5), "adapter_conv.conv2.bias": (bott,),
        "adapter_conv.conv3.weight": (bott, 1, 7, 7), "adapter_conv.conv3.bias": (bott,),
        "adapter_conv.projector.weight": (bott, bott, 1, 1), "adapter_conv.projector.bias": (bott,),
        "adapter_conv.noise_estimator.1.weight": (q, bott, 1, 1), "adapter_conv.noise_estimator.1.bias": (q,),
        "adapter_conv.noise_estimator.3.weight": (3, q, 1, 1), "adapter_conv.noise_estimator.3.bias": (3,),
        "norm.weight": (dim,), "norm.bias": (dim,),
    }
    return {k: shapes[k] for k in param_names(variant)}


def spatial_op(t_img, P, variant):
    """The *MonaOp on a [B, b, h, w] tensor (mona.py:85-93 / 178-195 / 279-295 / 401-424)."""
    ident = t_img
    xf = t_img
    if variant in ("freq_enhanced", "hybrid"):
        h, w = t_img.shape[-2:]
        spec = torch.fft.rfft2(t_img, dim=(-2, -1))                       # :284
        spec = spec * P["adapter_conv.freq_filter"].view(1, -1, 1, 1)     # :285
        xf = torch.fft.irfft2(spec, s=(h, w), dim=(-2, -1))               # :286
    b = t_img.shape[1]
    c1 = F.conv2d(xf, P["adapter_conv.conv1.weight"], P["adapter_conv.conv1.bias"], padding=1, groups=b)
    c2 = F.conv2d(xf, P["adapter_conv.conv2.weight"], P["adapter_conv.conv2.bias"], padding=2, groups=b)
    c3 = F.conv2d(xf, P["adapter_conv.conv3.weight"], P["adapter_conv.conv3.bias"], padding=3, groups=b)
    if variant in ("noise_aware", "hybrid"):
        pool = xf.mean(dim=(-2, -1), keepdim=True)                        # AdaptiveAvgPool2d(1) :171
        hid = F.relu(F.conv2d(pool, P["adapter_conv.noise_estimator.1.weight"], P["adapter_conv.noise_estimator.1.bias"]))
        wts = F.softmax(F.conv2d(hid, P["adapter_conv.noise_estimator.3.weight"], P["adapter_conv.noise_estimator.3.bias"]), dim=1)
        mix = c1 * wts[:, 0:1] + c2 * wts[:, 1:2] + c3 * wts[:, 2:3]      # :192 / :421
    else:
        mix = (c1 + c2 + c3) / 3.0                                        # :90 / :292
    c = mix + ident                                                       # identity is the UNfiltered input
    return c + F.conv2d(c, P["adapter_conv.projector.weight"], P["adapter_conv.projector.bias"])


def forward(x, P, variant, hw, keep_mask=None, p_drop=0.1, training=False):
    """x: [B, N, D] batch-first, N = 1 + h*w (CLS first).  Returns [B, N, D].

    keep_mask: optional {0,1} tensor [B, N, b]; when given (or training=False) the dropout of
    mona.py:147/249/357/483 is evaluated deterministically: d = g * keep / (1 - p_drop).
    """
    assert variant in VARIANTS
    D = x.shape[-1]
    B, N, _ = x.shape
    n = F.layer_norm(x, (D,), P["norm.weight"], P["norm.bias"], 1e-5)
    u = n * P["gamma"] + x * P["gammax"]                                  # :125
    t = F.linear(u, P["project1.weight"], P["project1.bias"])             # :127
    bott = t.shape[-1]
    if hw is None:                                                        # :140-144: no CLS token, ALL n tokens form a square grid
        h = w = int(math.sqrt(N))
        img = t.reshape(B, h, w, bott).permute(0, 3, 1, 2)                # (raises, like the reference, unless n is a perfect square)
        img = spatial_op(img, P, variant)
        z = img.permute(0, 2, 3, 1).reshape(B, N, bott)
    else:
        h, w = hw
        assert N == 1 + h * w
        cls_tok, spat = t[:, :1], t[:, 1:]
        img = spat.reshape(B, h, w, bott).permute(0, 3, 1, 2)             # :135
        img = spatial_op(img, P, variant)
        spat = img.permute(0, 2, 3, 1).reshape(B, h * w, bott)            # :137
        z = torch.cat([cls_tok, spat], dim=1)                             # :139
    g = F.gelu(z)                                                         # exact erf GELU :146
    if keep_mask is not None:
        g = g * keep_mask / (1.0 - p_drop)
    elif training:
        g = F.dropout(g, p_drop, True)
    y = F.linear(g, P["project2.weight"], P["project2.bias"])             # :148
    return x + y                                                          # :150


def merged_stencil(P, variant):
    """(Not used by forward.) The single 7x7 depth-wise stencil + bias that equals the /3 average
    of the three convolutions, for the kernel-side unit tests of that identity."""
    b = P["adapter_conv.conv1.weight"].shape[0]
    k = torch.zeros(b, 7, 7)
    k += P["adapter_conv.conv3.weight"][:, 0]
    k[:, 1:6, 1:6] += P["adapter_conv.conv2.weight"][:, 0]
    k[:, 2:5, 2:5] += P["adapter_conv.conv1.weight"][:, 0]
    bias = P["adapter_conv.conv1.bias"] + P["adapter_conv.conv2.bias"] + P["adapter_conv.conv3.bias"]
    return k / 3.0, bias / 3.0


def reference_init(variant, dim, bott, generator=None):
    """Parameters initialised like the reference constructors (mona.py:104-113: gamma=1e-6,
    gammax=1, freq_filter=1, nn.Linear / nn.Conv2d defaults = kaiming_uniform(a=sqrt(5)))."""
    P = {}
    for name, shp in param_shapes(variant, dim, bott).items():
        if name == "gamma":
            P[name] = torch.full(shp, 1e-6)
        elif name in ("gammax", "adapter_conv.freq_filter", "norm.weight"):
            P[name] = torch.ones(shp)
        elif name == "norm.bias":
            P[name] = torch.zeros(shp)
        else:
            if name.endswith("weight"):
                fan_in = math.prod(shp[1:])
            else:
                wshape = param_shapes(variant, dim, bott)[name[:-4] + "weight"]
                fan_in = math.prod(wshape[1:])
            bound = 1.0 / math.sqrt(fan_in)
            P[name] = (torch.rand(shp, generator=generator) * 2 - 1) * bound
    return P
