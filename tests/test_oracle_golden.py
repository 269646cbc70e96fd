"""Pin the CPU oracle (oracle/*.py) to the golden vectors captured from the imported reference
(oracle/gen_golden.py) and to the known-answer values of SURVEY.md Appendix B.  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import lora_ref, losses_ref, mona_ref, text_ref, vit_ref

VARIANTS = ("baseline", "noise_aware", "freq_enhanced", "hybrid")


def fill(shape, a, b, scale=1.0, fn=np.sin):
    n = int(np.prod(shape))
    return torch.from_numpy((scale * fn(a * np.arange(n, dtype=np.float64) + b)).astype(np.float32).reshape(shape))


def params_of(g, prefix="p."):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def close(a, b, rtol=2e-5, atol=2e-6):
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max())
    assert err <= atol + rtol * scale, f"max abs err {err:.3e} vs scale {scale:.3e}"


# ----------------------------------------------------------------------------- Appendix B KATs
def test_kat1_infonce():
    I, T = fill((4, 8), 0.37, 1.0), fill((4, 8), 0.23, 2.0, 1.0, np.cos)
    assert abs(float(losses_ref.info_nce(I, T, 0.07)) - 6.23591423) < 2e-6


KAT2 = {  # variant: (#params, sum y, sum|y|, sum|dx|, sum|dtheta|)
    "baseline": (1440, 6.6501746, 692.82996, 1385.7930, 336.73874),
    "noise_aware": (1467, 6.6485100, 692.83026, 1385.7944, 337.98044),
    "freq_enhanced": (1448, 6.7672405, 692.94092, 1385.9700, 342.51481),
    "hybrid": (1475, 6.7654457, 692.92352, 1385.9294, 343.83231),
}


@pytest.mark.parametrize("variant", VARIANTS)
def test_kat2_mona(variant):
    names = mona_ref.param_names(variant)
    shapes = mona_ref.param_shapes(variant, 32, 8)
    P = {k: fill(shapes[k], 0.37, float(n), 0.1).requires_grad_(True) for n, k in enumerate(names)}
    nparam, sy, say, sdx, sdp = KAT2[variant]
    assert sum(p.numel() for p in P.values()) == nparam
    x = fill((17, 2, 32), 0.11, 0.0).permute(1, 0, 2).contiguous().requires_grad_(True)   # -> batch-first
    y = mona_ref.forward(x, P, variant, (4, 4))
    y.square().sum().backward()
    assert abs(float(y.sum()) - sy) < 2e-4
    assert abs(float(y.abs().sum()) - say) < 2e-3
    assert abs(float(x.grad.abs().sum()) - sdx) < 5e-3
    assert abs(sum(float(p.grad.abs().sum()) for p in P.values()) - sdp) < 5e-3


def test_kat3_lora():
    W, b = fill((6, 8), 0.37, 0.0, 0.1), fill((6,), 0.37, 1.0, 0.1)
    A, B = fill((2, 8), 0.37, 2.0, 0.1), fill((6, 2), 0.37, 3.0, 0.1)
    assert abs(lora_ref.scaling(2, 4) - 2.8284271) < 1e-6
    y = lora_ref.linear_lora(fill((3, 8), 0.11, 0.0), W, b, A, B, 2, 4)
    assert abs(float(y.sum()) - 1.86544621) < 2e-6
    ref0 = torch.tensor([0.29604045, -0.13922282, 0.34197903, -0.12498981, 0.28950864, -0.15598631])
    close(y[0], ref0, 1e-6, 1e-6)


# ----------------------------------------------------------------------------- golden fixtures
@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("drop", [False, True])
def test_mona_matches_reference(golden, variant, drop):
    g = golden(f"mona_{variant}" + ("_drop" if drop else ""))
    P = {k: v.clone().requires_grad_(True) for k, v in params_of(g).items()}
    assert list(P) == mona_ref.param_names(variant)
    x = g["x_nbd"].permute(1, 0, 2).contiguous().requires_grad_(True)
    keep = g["keep_bnb"] if drop else None
    y = mona_ref.forward(x, P, variant, (4, 4), keep_mask=keep, p_drop=0.1)
    dy = g["dy_nbd"].permute(1, 0, 2) if drop else 2 * y.detach()
    y.backward(dy)
    close(y, g["y_nbd"].permute(1, 0, 2))
    close(x.grad, g["dx_nbd"].permute(1, 0, 2))
    for k in P:
        close(P[k].grad, g["g." + k], 5e-5, 1e-5)


@pytest.mark.parametrize("variant", VARIANTS)
def test_mona_without_hw_shapes_matches_reference(golden, variant):
    """forward(x) with hw_shapes=None (mona.py:140-144): every token goes through the spatial operator on a sqrt(n) grid."""
    g = golden(f"mona_{variant}_nohw")
    P = {k: v.clone().requires_grad_(True) for k, v in params_of(g).items()}
    x = g["x_nbd"].permute(1, 0, 2).contiguous().requires_grad_(True)
    y = mona_ref.forward(x, P, variant, None)
    y.backward(g["dy_nbd"].permute(1, 0, 2))
    close(y, g["y_nbd"].permute(1, 0, 2))
    close(x.grad, g["dx_nbd"].permute(1, 0, 2))
    for k in P:
        close(P[k].grad, g["g." + k], 5e-5, 1e-5)


def test_mona_merged_stencil_identity(golden):
    """rfft2*f_c*irfft2 == f_c scale, and (DW3+DW5+DW7)/3 == one merged 7x7."""
    g = golden("mona_freq_enhanced")
    P = params_of(g)
    t = fill((3, 8, 4, 4), 0.19, 0.4)
    want = mona_ref.spatial_op(t, P, "freq_enhanced")
    k, kb = mona_ref.merged_stencil(P, "freq_enhanced")
    xf = t * P["adapter_conv.freq_filter"].view(1, -1, 1, 1)
    c = torch.nn.functional.conv2d(xf, k[:, None], kb, padding=3, groups=8) + t
    got = c + torch.nn.functional.conv2d(c, P["adapter_conv.projector.weight"], P["adapter_conv.projector.bias"])
    close(got, want, 1e-5, 1e-6)


def test_lora_linear_matches_reference(golden):
    g = golden("lora_linear")
    x = g["x"].clone().requires_grad_(True)
    A, B, b = (g[k].clone().requires_grad_(True) for k in ("A", "B", "b"))
    y = lora_ref.linear_lora(x, g["W"], b, A, B, 2, 4)
    y.square().sum().backward()
    assert abs(lora_ref.scaling(2, 4) - float(g["scaling"])) < 1e-7
    close(y, g["y"]); close(x.grad, g["dx"]); close(A.grad, g["dA"]); close(B.grad, g["dB"]); close(b.grad, g["db"])


def test_lora_mha_matches_reference(golden):
    g = golden("lora_mha")
    P = {k: v.clone().requires_grad_("lora" in k) for k, v in params_of(g).items()}
    x = g["x_lbd"].clone().requires_grad_(True)
    y = lora_ref.mha_lora(x, P, 2, 4, 8)
    y.square().sum().backward()
    close(y, g["y_lbd"]); close(x.grad, g["dx_lbd"])
    for k in P:
        if "lora" in k:
            close(P[k].grad, g["g." + k], 5e-5, 1e-5)


@pytest.mark.parametrize("name,temp", [("infonce_kat1", 0.07), ("infonce_b6", 0.2)])
def test_infonce_matches_reference(golden, name, temp):
    g = golden(name)
    I, T = g["I"].clone().requires_grad_(True), g["T"].clone().requires_grad_(True)
    loss = losses_ref.info_nce(I, T, temp)
    loss.backward()
    close(loss, g["loss"]); close(I.grad, g["dI"]); close(T.grad, g["dT"])


def test_openai_clip_towers_match_reference(golden):
    g = golden("openai_clip_base")
    P = params_of(g)
    close(vit_ref.openai_vit_forward(g["images"], P, heads=2), g["image_features"], 1e-4, 1e-5)
    close(text_ref.openai_text_forward(g["ids"], P, heads=2), g["text_features"], 1e-4, 1e-5)


@pytest.mark.parametrize("variant", VARIANTS)
def test_openai_clip_mona_matches_reference(golden, variant):
    base, g = golden("openai_clip_base"), golden(f"openai_clip_mona_{variant}")
    P = params_of(base)
    mp = {k: v.clone().requires_grad_(True) for k, v in params_of(g).items()}
    assert int(g["count"]) == 2
    P.update(mp)
    fi = vit_ref.openai_vit_forward(base["images"], P, heads=2, mona=dict(variant=variant, hw=(4, 4)))
    ft = text_ref.openai_text_forward(base["ids"], P, heads=2)
    loss = losses_ref.info_nce(fi, ft, 0.07)
    loss.backward()
    close(fi, g["image_features"], 1e-4, 1e-5)
    close(loss, g["loss"], 1e-5, 1e-6)
    for k in mp:
        close(mp[k].grad, g["g." + k], 2e-4, 1e-6)


def test_openai_clip_lora_matches_reference(golden):
    base, g = golden("openai_clip_base"), golden("openai_clip_lora")
    P = params_of(base)
    P = {k: v for k, v in P.items() if not (k.startswith("visual.") and ".attn." in k)}
    lp = {k: v.clone().requires_grad_("lora" in k) for k, v in params_of(g).items()}
    P.update(lp)
    fi = vit_ref.openai_vit_forward(base["images"], P, heads=2, lora=dict(r=4, alpha=8))
    fi.square().sum().backward()
    close(fi, g["image_features"], 1e-4, 1e-5)
    for k in lp:
        if "lora" in k:
            close(lp[k].grad, g["g." + k], 2e-4, 1e-6)


# ----------------------------------------------------------------------------- third-party cross-checks
def test_timm_vit_recipe_matches_hf_vit(golden):
    """Appendix A.1 recipe == installed transformers ViTModel with mapped weights."""
    g = golden("hf_vit_tiny")
    hf = params_of(g, "hf.")
    P = {"visual.trunk.cls_token": hf["embeddings.cls_token"], "visual.trunk.pos_embed": hf["embeddings.position_embeddings"],
         "visual.trunk.patch_embed.proj.weight": hf["embeddings.patch_embeddings.projection.weight"],
         "visual.trunk.patch_embed.proj.bias": hf["embeddings.patch_embeddings.projection.bias"],
         "visual.trunk.norm.weight": hf["layernorm.weight"], "visual.trunk.norm.bias": hf["layernorm.bias"]}
    lay = "encoder.layer." if any(k.startswith("encoder.layer.") for k in hf) else "layers."
    for i in range(2):
        s, d = f"{lay}{i}.", f"visual.trunk.blocks.{i}."
        def pick(*cands):
            for c in cands:
                if s + c in hf:
                    return hf[s + c]
            raise KeyError(cands)
        for wb in ("weight", "bias"):
            q = pick(f"attention.attention.query.{wb}", f"attention.q_proj.{wb}")
            k = pick(f"attention.attention.key.{wb}", f"attention.k_proj.{wb}")
            v = pick(f"attention.attention.value.{wb}", f"attention.v_proj.{wb}")
            P[d + f"attn.qkv.{wb}"] = torch.cat([q, k, v], 0)
            P[d + f"attn.proj.{wb}"] = pick(f"attention.output.dense.{wb}", f"attention.o_proj.{wb}")
            P[d + f"norm1.{wb}"] = pick(f"layernorm_before.{wb}")
            P[d + f"norm2.{wb}"] = pick(f"layernorm_after.{wb}")
            P[d + f"mlp.fc1.{wb}"] = pick(f"intermediate.dense.{wb}", f"mlp.fc1.{wb}")
            P[d + f"mlp.fc2.{wb}"] = pick(f"output.dense.{wb}", f"mlp.fc2.{wb}")
    x = vit_ref.timm_vit_forward(g["images"], P, heads=2, return_tokens=True)
    x = torch.nn.functional.layer_norm(x, (32,), P["visual.trunk.norm.weight"], P["visual.trunk.norm.bias"], 1e-6)
    close(x, g["tokens_ln"], 1e-4, 1e-5)


def test_bert_recipe_matches_hf_bert(golden):
    """Appendix A.2 recipe == installed transformers BertModel on non-pad rows; and truncating a
    caption to its real length leaves the CLS vector unchanged (basis of the varlen text path)."""
    g = golden("hf_bert_tiny")
    P = {"text.transformer." + k: v for k, v in params_of(g, "hf.").items()}
    ids = g["ids"]
    hs = text_ref.bert_hidden(ids, P, heads=2)
    keep = ids != 0
    close(hs[keep], g["last_hidden_state"][keep], 1e-4, 1e-5)
    L0 = int(keep[0].sum())
    hs0 = text_ref.bert_hidden(ids[:1, :L0], P, heads=2)
    close(hs0[0, 0], g["last_hidden_state"][0, 0], 1e-4, 1e-5)


def test_clipseg_adapter_matches_reference(golden):
    """reference CLIPSegAdapter.forward (clipseg_adapter.py:73-98) around the installed HF CLIPSegDecoder: logits [B,2,H,W]
    and every decoder gradient (the CLIP backbone is frozen, :100-110)."""
    from oracle import clipseg_ref
    g = golden("clipseg_adapter")
    P = {k: v.clone() for k, v in params_of(g).items()}
    names = [k for k in P if k.startswith("decoder.")]
    leaves = {k: P[k].clone().requires_grad_(True) for k in names}
    Pq = dict(P); Pq.update(leaves)
    out = clipseg_ref.adapter_forward(g["images"], g["ids"], Pq, vit_heads=1, text_heads=2, extract_layers=(0, 1, 2))
    assert tuple(out.shape) == (2, 2, 64, 64)
    (out * g["dlogits"]).sum().backward()
    close(out, g["logits"], 1e-4, 1e-5)
    assert sorted("g." + k for k in names) == sorted(k for k in g if k.startswith("g."))
    for k in names:
        close(leaves[k].grad, g["g." + k], 2e-3, 1e-6)     # fp32 summation order differs from HF's eager attention / conv kernels


@pytest.mark.parametrize("task", ["seg", "cls"])
def test_fpn_adapter_matches_reference(golden, task):
    """reference TimmCLIPAdapter.forward (timm/clip_adapter.py:118-160), seg and cls heads, over a torch trunk with
    formula-filled weights that the generator and this test rebuild identically (oracle/fpn_ref.py::toy_trunk_params):
    outputs and every adapter-parameter gradient."""
    from oracle import fpn_ref
    g = golden("fpn_adapter")
    P = fpn_ref.toy_trunk_params()
    A = {k[2:]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith("A.")}
    out = fpn_ref.adapter_forward(g["images"], P, A, task=task)
    assert tuple(out.shape) == ((3, 2, 32, 32) if task == "seg" else (3, 2))
    (out * g[f"{task}.dy"]).sum().backward()
    close(out, g[f"{task}.y"], 1e-4, 1e-5)
    gnames = [k[len(task) + 3:] for k in g if k.startswith(f"{task}.g.")]
    assert gnames and all(n.startswith(("reduces.", "blocks.", "seg_head." if task == "seg" else "cls_head.")) for n in gnames)
    for n in gnames:
        close(A[n].grad, g[f"{task}.g.{n}"], 2e-3, 1e-6)


# ------------------------------------------------------------------------------------------------ MONAI DiceCE / Dice (f3)
DICE_CASES = ("A_empty_gt", "B_empty_pred", "C_three_class", "D_all_foreground")


@pytest.mark.parametrize("case", DICE_CASES)
def test_dicece_restatement_vs_independent_float64_vectors(golden, case):
    """oracle/losses_ref.dice_ce (torch, vectorised, autograd) against tests/golden/dicece_cases.npz — MONAI's published DiceCELoss
    algorithm restated a SECOND time, independently, in float64 numpy with per-pixel loops and a finite-difference gradient
    (oracle/gen_dice_golden.py).  MONAI itself is absent from the image: the pin is this cross-check plus the hand cases below."""
    from oracle import losses_ref
    z = golden("dicece_cases")
    logits = z[case + "_logits"].clone().requires_grad_(True)
    label = z[case + "_label"][:, None].float()
    loss = losses_ref.dice_ce(logits, label)
    loss.backward()
    assert abs(float(loss) - float(z[case + "_loss"])) < 2e-6 * abs(float(z[case + "_loss"]))
    g = z[case + "_grad"].float()
    assert float((logits.grad - g).abs().max()) < 2e-5 * float(g.abs().max()) + 1e-7
    if case + "_dice" in z:
        d, want = losses_ref.dice_metric(logits.detach(), label), z[case + "_dice"]
        assert torch.equal(torch.isnan(d), torch.isnan(want)) and torch.allclose(torch.nan_to_num(d), torch.nan_to_num(want), atol=1e-12)


def test_dicece_hand_cases():
    """Closed forms: a perfect, saturated prediction has loss -> 0; a uniform prediction on a 50/50 binary image has
    dice = 1 - (2*(n/4))/(n/4 + n/4 + n/2)... evaluated by hand: p = 1/2 everywhere, t one-hot."""
    from oracle import losses_ref
    H = W = 4
    label = torch.zeros(1, 1, H, W)
    label[0, 0, :2] = 1                                                  # 8 foreground, 8 background pixels
    sat = torch.stack([(1 - label[0, 0]) * 40 - 20, label[0, 0] * 40 - 20])[None]
    assert float(losses_ref.dice_ce(sat, label)) < 1e-6
    uni = torch.zeros(1, 2, H, W)
    # per channel: I = 8*0.5 = 4, sum p^2 = 16*0.25 = 4, sum t^2 = 8  ->  dice_c = 1 - 8/12 = 1/3; CE = ln 2
    want = 1.0 / 3.0 + math.log(2.0)
    assert abs(float(losses_ref.dice_ce(uni, label)) - want) < 1e-6
    # metric: prediction = all background on an image with foreground -> 0; empty ground truth -> NaN
    d = losses_ref.dice_metric(torch.cat([sat * -1, sat]), torch.cat([label, torch.zeros_like(label)]))
    assert float(d[0]) == 0.0 and math.isnan(float(d[1]))


# ----------------------------------------------------------------------------- round-5 fixtures (oracle/gen_golden_r05.py)
@pytest.mark.parametrize("task", ["seg", "cls"])
def test_clip_adapter_openai_matches_reference(golden, task):
    """oracle/fpn_ref.openai_adapter_forward against the reference's CLIPAdapter (src/third_party/openai_clip/clip_adapter.py:6-165): output and every gradient."""
    from oracle import fpn_ref
    base, g = golden("openai_clip_base"), golden("clip_adapter_openai")
    P = params_of(base)
    P.update({k: v.clone().requires_grad_(True) for k, v in params_of(g).items()})
    A = {k[2:]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith("A.")}
    y = fpn_ref.openai_adapter_forward(g["images"], P, A, task=task, extract_layers=(0, 1), heads=2, img_size=32, mona=dict(variant="freq_enhanced", hw=(4, 4)))
    (y * g[f"{task}.dy"]).sum().backward()
    close(y, g[f"{task}.y"], 1e-4, 1e-5)
    gmax = max(float(v.abs().max()) for k, v in g.items() if k.startswith(f"{task}.g."))
    for k, v in g.items():
        if k.startswith(f"{task}.g."):
            name = k[len(task) + 3:]
            got = P[name[len("clip_model."):]].grad if name.startswith("clip_model.") else A[name].grad
            assert float((got - v).abs().max()) < 1e-4 * max(float(v.abs().max()), 1e-2 * gmax), name


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("case", ["eval", "drop", "nohw"])
def test_mona_at_bottleneck_64_matches_reference(golden, variant, case):
    """The fixtures the HIP path is run on directly (tests/test_golden_gpu.py) also pin the oracle at that geometry (width 128, bottleneck 64)."""
    g = golden(f"ref_mona_{variant}_d128")
    P = {k: v.clone().requires_grad_(True) for k, v in params_of(g).items()}
    x = g[f"{case}.x"].permute(1, 0, 2).contiguous().requires_grad_(True)
    y = mona_ref.forward(x, P, variant, None if case == "nohw" else (4, 4), keep_mask=g["drop.keep"] if case == "drop" else None, p_drop=0.1)
    y.backward(g[f"{case}.dy"].permute(1, 0, 2))
    close(y, g[f"{case}.y"].permute(1, 0, 2))
    close(x.grad, g[f"{case}.dx"].permute(1, 0, 2))
    for k in P:
        close(P[k].grad, g[f"{case}.g." + k], 1e-4, 1e-5)
