"""Generate tests/golden/*.npz from the IMPORTED reference.  Runs only in the build container
(needs /root/reference); the fixtures it writes are committed, the reference code never is.

    python oracle/gen_golden.py            # rewrites tests/golden/

Each fixture holds inputs, deterministic formula-filled parameters, and the reference's outputs
and autograd gradients, at toy dimensions (KBs).  fill(shape,a,b,scale,fn) = scale*fn(a*arange+b)
evaluated in float64 and cast to fp32, so fixtures are reproducible bit for bit.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def fill(shape, a, b, scale=1.0, fn=np.sin):
    n = int(np.prod(shape))
    return torch.from_numpy((scale * fn(a * np.arange(n, dtype=np.float64) + b)).astype(np.float32).reshape(shape))


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def np_dict(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    np.savez(os.path.join(OUT, name + ".npz"), **np_dict(arrays))
    print("wrote", name, sum(np.asarray(v).size for v in np_dict(arrays).values()), "values")


def fill_module(mod, a=0.37, scale=0.1):
    with torch.no_grad():
        for n, (_, p) in enumerate(mod.named_parameters()):
            p.copy_(fill(tuple(p.shape), a, float(n), scale))


def gen_mona(mona):
    classes = {"baseline": mona.BaselineMona, "noise_aware": mona.NoiseAwareMona,
               "freq_enhanced": mona.FreqEnhancedMona, "hybrid": mona.HybridNoiseFreqMona}
    for variant, cls in classes.items():
        torch.manual_seed(0)
        m = cls(32, 8).eval()
        fill_module(m)
        # KAT2 input (sequence-first [N,B,D]) plus a second, larger-amplitude case with a dropout mask
        x = fill((17, 2, 32), 0.11, 0.0).requires_grad_(True)
        y = m(x, (4, 4))
        (y.square().sum()).backward()
        arrays = {"x_nbd": x, "y_nbd": y, "dx_nbd": x.grad}
        for k, p in m.named_parameters():
            arrays["p." + k] = p
            arrays["g." + k] = p.grad
        save(f"mona_{variant}", **arrays)

        # training-mode case: deterministic keep mask injected by patching the module's dropout
        m.zero_grad()
        keep = (fill((2, 17, 8), 0.77, 1.0) > -0.6).float()      # ~80 % kept, batch-first [B,N,b]

        class FixedDrop(torch.nn.Module):
            def forward(self, g):
                return g * keep / 0.9
        m.dropout = FixedDrop()
        x2 = (fill((17, 2, 32), 0.23, 1.0) * 3.0).requires_grad_(True)
        y2 = m(x2, (4, 4))
        (y2 * fill((17, 2, 32), 0.05, 2.0)).sum().backward()
        arrays = {"x_nbd": x2, "y_nbd": y2, "dx_nbd": x2.grad, "keep_bnb": keep, "dy_nbd": fill((17, 2, 32), 0.05, 2.0)}
        for k, p in m.named_parameters():
            arrays["p." + k] = p
            arrays["g." + k] = p.grad
        save(f"mona_{variant}_drop", **arrays)

        # hw_shapes=None (mona.py:140-144): no CLS token, all 16 tokens form a 4x4 grid
        m2 = cls(32, 8).eval()
        fill_module(m2)
        x3 = fill((16, 2, 32), 0.13, 0.5).requires_grad_(True)
        y3 = m2(x3)
        (y3 * fill((16, 2, 32), 0.07, 1.0)).sum().backward()
        arrays = {"x_nbd": x3, "y_nbd": y3, "dx_nbd": x3.grad, "dy_nbd": fill((16, 2, 32), 0.07, 1.0)}
        for k, p in m2.named_parameters():
            arrays["p." + k] = p
            arrays["g." + k] = p.grad
        save(f"mona_{variant}_nohw", **arrays)


def gen_lora(lora):
    lin = torch.nn.Linear(8, 6)
    with torch.no_grad():
        lin.weight.copy_(fill((6, 8), 0.37, 0.0, 0.1))
        lin.bias.copy_(fill((6,), 0.37, 1.0, 0.1))
    ll = lora.LinearLoRA(lin, r=2, lora_alpha=4, dropout_rate=0.0)
    with torch.no_grad():
        ll.w_lora_A.copy_(fill((2, 8), 0.37, 2.0, 0.1))
        ll.w_lora_B.copy_(fill((6, 2), 0.37, 3.0, 0.1))
    x = fill((3, 8), 0.11, 0.0).requires_grad_(True)
    y = ll(x)
    y.square().sum().backward()
    save("lora_linear", x=x, y=y, dx=x.grad, W=ll.weight, b=ll.bias, A=ll.w_lora_A, B=ll.w_lora_B,
         dA=ll.w_lora_A.grad, dB=ll.w_lora_B.grad, db=ll.bias.grad, scaling=np.float32(ll.scaling))

    mha = torch.nn.MultiheadAttention(16, 2)
    fill_module(mha, 0.29, 0.2)
    pm = lora.PlainMultiheadAttentionLoRA(mha, enable_lora=["q", "k", "v", "o"], r=4, lora_alpha=8, dropout_rate=0.0)
    with torch.no_grad():
        for n, (k, p) in enumerate(pm.named_parameters()):
            if "lora" in k:
                p.copy_(fill(tuple(p.shape), 0.41, float(n), 0.2))
    x = fill((5, 3, 16), 0.13, 0.5).requires_grad_(True)
    y, _ = pm(x, x, x)
    y.square().sum().backward()
    arrays = {"x_lbd": x, "y_lbd": y, "dx_lbd": x.grad}
    for k, p in pm.named_parameters():
        arrays["p." + k] = p
        if p.grad is not None:
            arrays["g." + k] = p.grad
    save("lora_mha", **arrays)


def gen_infonce(losses):
    crit = losses.InfoNCELoss(0.07)
    I = fill((4, 8), 0.37, 1.0).requires_grad_(True)
    T = fill((4, 8), 0.23, 2.0, 1.0, np.cos).requires_grad_(True)
    loss = crit(I, T)
    loss.backward()
    save("infonce_kat1", I=I, T=T, loss=loss, dI=I.grad, dT=T.grad)
    I = (fill((6, 16), 0.91, 0.3) * 2).requires_grad_(True)
    T = (fill((6, 16), 0.53, 1.1) * 0.5).requires_grad_(True)
    loss = losses.InfoNCELoss(0.2)(I, T)
    loss.backward()
    save("infonce_b6", I=I, T=T, loss=loss, dI=I.grad, dT=T.grad, temperature=np.float32(0.2))


def gen_openai_clip(model_mod, mona, lora):
    torch.manual_seed(0)
    clip = model_mod.CLIP(16, 32, 2, 128, 8, 8, 50, 64, 2, 2).float().eval()   # embed16 res32 2 layers vision w128 (2 heads) patch8; text ctx8 w64 2 heads
    fill_module(clip, 0.31, 0.08)
    with torch.no_grad():   # keep LN gains near 1 so activations stay O(1)
        for k, p in clip.named_parameters():
            if k.endswith(("ln_1.weight", "ln_2.weight", "ln_pre.weight", "ln_post.weight", "ln_final.weight")):
                p.add_(1.0)
    img = fill((3, 3, 32, 32), 0.017, 0.0) * 0.5 + 0.5
    ids = torch.tensor([[49, 3, 7, 11, 2, 0, 0, 0], [5, 49, 1, 1, 1, 1, 1, 1], [4, 9, 8, 7, 6, 5, 3, 49]])
    with torch.no_grad():
        fi, ft = clip.encode_image(img), clip.encode_text(ids)
    arrays = {"images": img, "ids": ids, "image_features": fi, "text_features": ft}
    for k, v in clip.state_dict().items():
        arrays["p." + k] = v
    save("openai_clip_base", **arrays)

    # + Mona (all four variants share the same backbone fill)
    for variant in ("baseline", "noise_aware", "freq_enhanced", "hybrid"):
        torch.manual_seed(0)
        c2 = model_mod.CLIP(16, 32, 2, 128, 8, 8, 50, 64, 2, 2).float().eval()
        c2.load_state_dict(clip.state_dict())
        for p in c2.parameters():
            p.requires_grad_(False)
        c2, cnt = mona.inject_mona_variant_to_clip(c2, variant=variant, bottleneck_dim=8)
        c2.eval()   # the injected adapters are built in training mode; parity vectors are dropout-free
        mp = [(k, p) for k, p in c2.named_parameters() if "mona" in k]
        with torch.no_grad():
            for n, (k, p) in enumerate(mp):
                p.copy_(fill(tuple(p.shape), 0.37, float(n), 0.1))
                if k.endswith(("norm.weight", "gammax")):
                    p.add_(1.0)
        for k, p in mp:
            p.requires_grad_(True)
        fi = c2.encode_image(img)
        with torch.no_grad():
            ft = c2.encode_text(ids)
        from_losses = load_by_path("ref_losses", "src/losses/losses.py")
        loss = from_losses.InfoNCELoss(0.07)(fi, ft)
        loss.backward()
        arrays = {"image_features": fi, "loss": loss, "count": np.int32(cnt)}
        for k, p in mp:
            arrays["p." + k] = p
            arrays["g." + k] = p.grad
        save(f"openai_clip_mona_{variant}", **arrays)

    # + LoRA r=4 on q,k,v,o
    c3 = model_mod.CLIP(16, 32, 2, 128, 8, 8, 50, 64, 2, 2).float().eval()
    c3.load_state_dict(clip.state_dict())
    for p in c3.parameters():
        p.requires_grad_(False)
    c3, cnt = lora.inject_lora_to_clip(c3, lora_r=4, lora_alpha=8, lora_dropout=0.0)
    c3.eval()
    lp = [(k, p) for k, p in c3.named_parameters() if "lora" in k]
    with torch.no_grad():
        for n, (k, p) in enumerate(lp):
            p.copy_(fill(tuple(p.shape), 0.43, float(n), 0.15))
    for k, p in lp:
        p.requires_grad_(True)
    fi = c3.encode_image(img)
    fi.square().sum().backward()
    arrays = {"image_features": fi, "count": np.int32(cnt)}
    for k, p in c3.named_parameters():
        if "attn" in k and "visual" in k:
            arrays["p." + k] = p
        if "lora" in k:
            arrays["g." + k] = p.grad
    save("openai_clip_lora", **arrays)


def gen_hf_crosschecks():
    """Third-party towers: capture the installed transformers implementations (the reference pins
    transformers 4.57.1; 5.15.0 is what is installed here) for the timm-ViT and BERT recipes."""
    from transformers import BertConfig, BertModel, ViTConfig, ViTModel
    torch.manual_seed(0)
    vcfg = ViTConfig(hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64, image_size=32,
                     patch_size=8, layer_norm_eps=1e-6, hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    vit = ViTModel(vcfg, add_pooling_layer=False).eval()
    fill_module(vit, 0.27, 0.1)
    with torch.no_grad():
        for k, p in vit.named_parameters():
            if "layernorm" in k and k.endswith("weight"):
                p.add_(1.0)
    img = fill((2, 3, 32, 32), 0.019, 0.3) * 0.5 + 0.5
    with torch.no_grad():
        out = vit(pixel_values=img).last_hidden_state            # includes final layernorm
    arrays = {"images": img, "tokens_ln": out}
    for k, v in vit.state_dict().items():
        arrays["hf." + k] = v
    save("hf_vit_tiny", **arrays)

    bcfg = BertConfig(vocab_size=60, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                      max_position_embeddings=16, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-12)
    bert = BertModel(bcfg, add_pooling_layer=False).eval()
    fill_module(bert, 0.33, 0.12)
    with torch.no_grad():
        for k, p in bert.named_parameters():
            if "LayerNorm.weight" in k:
                p.add_(1.0)
    ids = torch.tensor([[2, 17, 33, 41, 3, 0, 0, 0, 0, 0, 0, 0], [2, 9, 8, 7, 6, 5, 4, 11, 12, 13, 14, 3], [2, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]])
    with torch.no_grad():
        hs = bert(input_ids=ids, attention_mask=(ids != 0).long()).last_hidden_state
    arrays = {"ids": ids, "last_hidden_state": hs}
    for k, v in bert.state_dict().items():
        arrays["hf." + k] = v
    save("hf_bert_tiny", **arrays)


def gen_clipseg(model_mod):
    """Reference CLIPSegAdapter (built with __new__: its __init__ downloads weights) around the installed HF decoder."""
    adapter_mod = load_by_path("ref_clipseg_adapter", "src/third_party/openai_clip/clipseg_adapter.py")
    from transformers import CLIPSegConfig
    from transformers.models.clipseg.modeling_clipseg import CLIPSegDecoder
    torch.manual_seed(0)
    clip = model_mod.CLIP(64, 64, 3, 64, 16, 8, 50, 64, 2, 2).float().eval()         # image 64, patch 16 -> 4x4 grid; 3 vision layers, width 64 (1 head)
    fill_module(clip, 0.31, 0.08)
    with torch.no_grad():
        for k, p in clip.named_parameters():
            if k.endswith(("ln_1.weight", "ln_2.weight", "ln_pre.weight", "ln_post.weight", "ln_final.weight")):
                p.add_(1.0)
    cfg = CLIPSegConfig(use_complex_transposed_convolution=True, reduce_dim=64, extract_layers=[0, 1, 2], conditional_layer=0,
                        decoder_num_attention_heads=4, decoder_intermediate_size=128, projection_dim=64,
                        vision_config={"patch_size": 16, "image_size": 64, "hidden_size": 64, "num_attention_heads": 1, "intermediate_size": 256, "num_hidden_layers": 3})
    dec = CLIPSegDecoder(cfg).eval()
    fill_module(dec, 0.29, 0.15)
    with torch.no_grad():
        for k, p in dec.named_parameters():
            if "layer_norm" in k and k.endswith("weight"):
                p.add_(1.0)
    ad = adapter_mod.CLIPSegAdapter.__new__(adapter_mod.CLIPSegAdapter)
    torch.nn.Module.__init__(ad)
    ad.clip_model, ad.decoder, ad.extract_layers = clip, dec, cfg.extract_layers
    ad.freeze_clip_backbone()
    img = fill((2, 3, 64, 64), 0.013, 0.2) * 0.5 + 0.5
    ids = torch.tensor([[49, 3, 7, 11, 2, 0, 0, 0], [5, 9, 49, 1, 1, 1, 1, 1]])
    out = ad(img, input_ids=ids)
    (out * fill(tuple(out.shape), 0.007, 0.4)).sum().backward()
    arrays = {"images": img, "ids": ids, "logits": out, "dlogits": fill(tuple(out.shape), 0.007, 0.4)}
    for k, v in ad.state_dict().items():
        arrays["p." + k] = v
    for k, p in ad.named_parameters():
        if p.grad is not None:
            arrays["g." + k] = p.grad
    assert all(k.startswith("decoder.") for k, p in ad.named_parameters() if p.requires_grad)
    save("clipseg_adapter", **arrays)


def gen_fpn():
    """TimmCLIPAdapter (seg and cls heads) of the imported reference over a torch trunk with formula-filled weights."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import fpn_ref, vit_ref
    ca = load_by_path("ref_timm_clip_adapter", "src/third_party/timm/clip_adapter.py")
    import torch.nn as nn
    P = fpn_ref.toy_trunk_params()
    pre = "visual.trunk."

    class Block(nn.Module):                     # timm Block arithmetic via the (separately pinned) functional restatement
        def __init__(self, i):
            super().__init__()
            self.i = i

        def forward(self, x):
            return vit_ref.timm_block(x, vit_ref._sub(P, f"{pre}blocks.{self.i}."), heads=12)

    class PatchEmbed(nn.Module):
        def forward(self, x):
            W = P[pre + "patch_embed.proj.weight"]
            return torch.nn.functional.conv2d(x, W, P[pre + "patch_embed.proj.bias"], stride=W.shape[-1]).flatten(2).transpose(1, 2)

    trunk = nn.Module()
    trunk.patch_embed, trunk.pos_drop, trunk.norm = PatchEmbed(), nn.Identity(), nn.LayerNorm(768, eps=1e-6)
    trunk.cls_token, trunk.pos_embed = nn.Parameter(P[pre + "cls_token"].clone()), nn.Parameter(P[pre + "pos_embed"].clone())
    trunk.blocks = nn.ModuleList([Block(i) for i in range(3)])
    with torch.no_grad():
        trunk.norm.weight.copy_(P[pre + "norm.weight"]); trunk.norm.bias.copy_(P[pre + "norm.bias"])
    clip = nn.Module(); clip.visual = nn.Module(); clip.visual.trunk = trunk
    images = torch.from_numpy(np.random.RandomState(11).uniform(0, 1, (3, 3, 32, 32)).astype(np.float32))
    out = {}
    for task in ("seg", "cls"):
        ad = ca.TimmCLIPAdapter(clip, extract_layers=[0, 1, 2], reduce_dim=64, num_classes=2, img_size=32, patch_size=8, task=task)
        rs = np.random.RandomState(7)                                    # seeded Gaussian adapter weights (stored in the fixture)
        with torch.no_grad():
            for k, p_ in torch.nn.ModuleList([ad.reduces, ad.blocks, ad.seg_head, ad.cls_head]).named_parameters():
                p_.copy_(torch.from_numpy((rs.standard_normal(p_.numel()) * 0.08).astype(np.float32).reshape(tuple(p_.shape))))
            for blk in ad.blocks:                                        # keep LayerNorm scales near one
                blk[0].weight.add_(1.0)
        ad.eval()
        ad.freeze_clip_backbone()
        A = {k: v.detach().clone() for k, v in ad.state_dict().items() if not k.startswith("clip_model.")}
        y = ad(images)
        dy = torch.from_numpy(np.random.RandomState(13).standard_normal(tuple(y.shape)).astype(np.float32))
        (y * dy).sum().backward()
        yo = fpn_ref.adapter_forward(images, P, A, task=task)
        assert float((yo - y).abs().max()) < 1e-4 * float(y.abs().max()), "oracle restatement deviates from the reference"
        if task == "seg":
            out.update({f"A.{k}": v for k, v in A.items()})
        out[f"{task}.y"] = y.detach(); out[f"{task}.dy"] = dy
        for k, p in ad.named_parameters():
            if p.requires_grad and p.grad is not None and not k.startswith("clip_model."):
                out[f"{task}.g.{k}"] = p.grad.detach()
    save("fpn_adapter", images=images, **out)


def main():
    sys.path.insert(0, REF)
    mona = load_by_path("ref_mona", "src/adapters/mona.py")
    lora = load_by_path("ref_lora", "src/adapters/lora.py")
    losses = load_by_path("ref_losses", "src/losses/losses.py")
    model_mod = load_by_path("ref_clip_model", "src/third_party/openai_clip/model.py")
    gen_mona(mona)
    gen_lora(lora)
    gen_infonce(losses)
    gen_openai_clip(model_mod, mona, lora)
    gen_hf_crosschecks()
    gen_fpn()
    gen_clipseg(model_mod)


if __name__ == "__main__":
    main()
