"""-m gpu: the HIP path run DIRECTLY on the vectors captured from the imported reference (tests/golden/*.npz, written by oracle/gen_golden.py and
oracle/gen_golden_r05.py) — no oracle in between: reference -> fixture -> HIP.  fp32 mode, 1e-3 relative (max |delta| / max |ref| per tensor), the
north_star's bound for logits; gradients get the same bound with a floor at 1e-3 of the largest gradient tensor of the case (tensors whose true gradient
is a near-cancelling sum).

Fixtures whose geometry the kernels do not take (Mona bottleneck 8 / width 32, LoRA on 8-wide linears: mona_*.npz, lora_*.npz, openai_clip_mona_*.npz of
round 1) pin the ORACLE (tests/test_oracle_golden.py); their round-5 twins at bottleneck 64 / widths that are multiples of 64 (ref_*_d128 / _k128 / _b64) are
the same reference classes at shapes the HIP path runs."""
import pytest
import torch

pytestmark = pytest.mark.gpu
VARIANTS = ("baseline", "noise_aware", "freq_enhanced", "hybrid")
TOL = 1e-3


def dev():
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _fp32_mode():
    from uia_hip import functional as UF
    UF.set_compute_dtype(torch.float32)
    yield
    UF.set_compute_dtype(torch.bfloat16)
    UF.clear_t_copies()


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def params_of(g, prefix="p."):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def check_grads(named_params, want, floor_scale=None):
    """every tensor in `want` (name -> reference gradient) against the product's .grad; floor: 1e-3 of the largest reference gradient of the case"""
    gmax = max(float(v.abs().max()) for v in want.values()) if floor_scale is None else floor_scale
    named = dict(named_params)
    assert want, "no gradients in the fixture"
    for k, w in want.items():
        got = named[k].grad
        assert got is not None, f"{k}: no gradient"
        err = float((got.detach().float().cpu() - w).abs().max())
        assert err < TOL * float(w.abs().max()) or err < TOL * gmax, (k, err, float(w.abs().max()), gmax)


# ------------------------------------------------------------------------------------------------ Mona (a1-a5)
@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("case", ["eval", "drop", "nohw"])
def test_mona_module_on_reference_vectors(golden, variant, case):
    """src/adapters/mona.py classes (reference mona.py:75-487; sequence-first [N, B, D] API, hw_shapes (4, 4) or None, the keep mask in place of the dropout draw)."""
    from src.adapters import mona as M
    g = golden(f"ref_mona_{variant}_d128")
    mod = M._VARIANTS[variant](128, 64)
    mod.load_state_dict(params_of(g))
    mod = mod.to(dev())
    mod.train(case == "drop")
    mod.keep_mask = g["drop.keep"].to(torch.uint8).to(dev()).contiguous() if case == "drop" else None
    x = g[f"{case}.x"].to(dev()).requires_grad_(True)
    y = mod(x, (4, 4)) if case != "nohw" else mod(x)
    y.backward(g[f"{case}.dy"].to(dev()))
    assert rel(y, g[f"{case}.y"]) < TOL
    assert rel(x.grad, g[f"{case}.dx"]) < TOL
    check_grads(mod.named_parameters(), params_of(g, f"{case}.g."))


# ------------------------------------------------------------------------------------------------ LoRA (a7, a8)
def test_linear_lora_on_reference_vectors(golden):
    """LinearLoRA (reference lora.py:57-90) in rank form: y, dx, dA, dB and the (trainable, quirk C-4) bias gradient."""
    from src.adapters.lora import LinearLoRA
    g = golden("ref_lora_linear_k128")
    lin = torch.nn.Linear(128, 192)
    with torch.no_grad():
        lin.weight.copy_(g["W"])
        lin.bias.copy_(g["b"])
    ll = LinearLoRA(lin, r=4, lora_alpha=8, dropout_rate=0.0)
    with torch.no_grad():
        ll.w_lora_A.copy_(g["A"])
        ll.w_lora_B.copy_(g["B"])
    assert abs(ll.scaling - float(g["scaling"])) < 1e-7
    ll = ll.to(dev()).eval()
    x = g["x"].to(dev()).requires_grad_(True)
    y = ll(x)
    y.backward(g["dy"].to(dev()))
    assert rel(y, g["y"]) < TOL and rel(x.grad, g["dx"]) < TOL
    check_grads(ll.named_parameters(), {"w_lora_A": g["dA"], "w_lora_B": g["dB"], "bias": g["db"]})


def test_mha_lora_on_reference_vectors(golden):
    """PlainMultiheadAttentionLoRA (reference lora.py:96-199) over an nn.MultiheadAttention: sequence-first self-attention, (out, None)."""
    from src.adapters.lora import PlainMultiheadAttentionLoRA
    g = golden("ref_lora_mha_d128")
    pm = PlainMultiheadAttentionLoRA(torch.nn.MultiheadAttention(128, 2), enable_lora=["q", "k", "v", "o"], r=4, lora_alpha=8, dropout_rate=0.0)
    pm.load_state_dict(params_of(g))
    pm = pm.to(dev()).eval()
    x = g["x_lbd"].to(dev()).requires_grad_(True)
    y, w = pm(x, x, x)
    assert w is None
    y.backward(g["dy_lbd"].to(dev()))
    assert rel(y, g["y_lbd"]) < TOL and rel(x.grad, g["dx_lbd"]) < TOL
    check_grads(pm.named_parameters(), {k: v for k, v in params_of(g, "g.").items()})


# ------------------------------------------------------------------------------------------------ InfoNCE (a10)
@pytest.mark.parametrize("name,temp", [("infonce_kat1", 0.07), ("infonce_b6", 0.2)])
def test_infonce_on_reference_vectors(golden, name, temp):
    from src.losses import InfoNCELoss
    g = golden(name)
    I, T = g["I"].to(dev()).requires_grad_(True), g["T"].to(dev()).requires_grad_(True)
    loss = InfoNCELoss(temp)(I, T)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < TOL * abs(float(g["loss"]))
    assert rel(I.grad, g["dI"]) < TOL and rel(T.grad, g["dT"]) < TOL


# ------------------------------------------------------------------------------------------------ OpenAI CLIP towers (a11, a12) and their adapters (a6, a9)
def _base_clip(golden):
    from src.third_party.openai_clip.model import CLIP
    base = golden("openai_clip_base")
    clip = CLIP(16, 32, 2, 128, 8, 8, 50, 64, 2, 2).float().eval()
    clip.load_state_dict(params_of(base))
    for p in clip.parameters():
        p.requires_grad_(False)
    return clip, base


def test_openai_clip_towers_on_reference_vectors(golden):
    clip, base = _base_clip(golden)
    clip = clip.to(dev())
    assert rel(clip.encode_image(base["images"].to(dev())), base["image_features"]) < TOL
    assert rel(clip.encode_text(base["ids"].to(dev())), base["text_features"]) < TOL


@pytest.mark.parametrize("variant", VARIANTS)
def test_openai_clip_mona_on_reference_vectors(golden, variant):
    """inject_mona_variant_to_clip(variant, bottleneck 64, num_layers=1) (reference mona.py:495-575) + InfoNCE: features, loss and every adapter gradient."""
    from src.adapters import inject_mona_variant_to_clip
    from src.losses import InfoNCELoss
    clip, base = _base_clip(golden)
    g = golden(f"ref_openai_clip_mona_{variant}_b64")
    clip, n = inject_mona_variant_to_clip(clip, variant=variant, bottleneck_dim=64, num_layers=1)
    assert n == int(g["count"])
    sd = clip.state_dict()
    P = params_of(g)
    assert sorted(k for k in sd if "mona" in k) == sorted(P)              # the checkpoint wire format: same names as the reference's injector produced
    sd.update(P)
    clip.load_state_dict(sd)
    for k, p in clip.named_parameters():
        p.requires_grad_("mona" in k)
    clip = clip.to(dev()).eval()
    fi = clip.encode_image(base["images"].to(dev()))
    with torch.no_grad():
        ft = clip.encode_text(base["ids"].to(dev()))
    loss = InfoNCELoss(0.07)(fi, ft)
    loss.backward()
    assert rel(fi, g["image_features"]) < TOL
    assert abs(float(loss) - float(g["loss"])) < TOL * max(1.0, abs(float(g["loss"])))
    check_grads(clip.named_parameters(), params_of(g, "g."))


def test_openai_clip_lora_on_reference_vectors(golden):
    """inject_lora_to_clip(r=4, alpha=8) (reference lora.py:202-248): image features and the LoRA factor gradients of d(sum f^2)."""
    from src.adapters import inject_lora_to_clip
    clip, base = _base_clip(golden)
    g = golden("openai_clip_lora")
    clip, n = inject_lora_to_clip(clip, lora_r=4, lora_alpha=8, lora_dropout=0.0)
    assert n == int(g["count"])
    sd = clip.state_dict()
    sd.update(params_of(g))
    clip.load_state_dict(sd)
    for k, p in clip.named_parameters():
        p.requires_grad_("lora" in k)
    clip = clip.to(dev()).eval()
    fi = clip.encode_image(base["images"].to(dev()))
    fi.square().sum().backward()
    assert rel(fi, g["image_features"]) < TOL
    check_grads(clip.named_parameters(), params_of(g, "g."))


# ------------------------------------------------------------------------------------------------ CLIPSeg (a14, a15)
def test_clipseg_adapter_on_reference_vectors(golden):
    """CLIPSegAdapter.forward (reference clipseg_adapter.py:73-98 around the transformers decoder) on the reference's own logits and decoder gradients."""
    from src.third_party.openai_clip.model import CLIP
    from src.third_party.openai_clip.clipseg_adapter import CLIPSegAdapter, CLIPSegDecoder
    g = golden("clipseg_adapter")
    clip = CLIP(64, 64, 3, 64, 16, 8, 50, 64, 2, 2).float().eval()
    dec = CLIPSegDecoder(vision_hidden=64, projection_dim=64, extract_layers=(0, 1, 2), intermediate=128, patch_size=16)
    model = CLIPSegAdapter(clip, decoder=dec)
    P = params_of(g)
    assert sorted(P) == sorted(model.state_dict())                          # same names as the reference adapter + HF decoder
    model.load_state_dict(P)
    model.freeze_clip_backbone()
    model = model.to(dev()).eval()
    out = model(g["images"].to(dev()), input_ids=g["ids"].to(dev()))
    (out * g["dlogits"].to(dev())).sum().backward()
    assert tuple(out.shape) == tuple(g["logits"].shape) and rel(out, g["logits"]) < TOL
    check_grads(model.named_parameters(), params_of(g, "g."))


# ------------------------------------------------------------------------------------------------ CLIPAdapter (f1)
@pytest.mark.parametrize("task", ["seg", "cls"])
def test_clip_adapter_openai_layout_on_reference_vectors(golden, task):
    """CLIPAdapter (reference src/third_party/openai_clip/clip_adapter.py:6-165) over the OpenAI-layout tower with freq_enhanced Mona adapters left trainable by
    freeze_clip_backbone(): head output, every head gradient and the adapter gradients THROUGH the tapped backbone."""
    from src.adapters import inject_mona_variant_to_clip
    from src.third_party.openai_clip.clip_adapter import CLIPAdapter
    clip, base = _base_clip(golden)
    g = golden("clip_adapter_openai")
    clip, n = inject_mona_variant_to_clip(clip, variant="freq_enhanced", bottleneck_dim=64, num_layers=1)
    assert n == int(g["count"])
    sd = clip.state_dict()
    sd.update(params_of(g))
    clip.load_state_dict(sd)
    ad = CLIPAdapter(clip, extract_layers=[0, 1], reduce_dim=64, num_classes=2, img_size=32, patch_size=8, task=task)
    sda = ad.state_dict()
    sda.update({k[2:]: v for k, v in g.items() if k.startswith("A.")})
    ad.load_state_dict(sda)
    ad.eval()
    ad.freeze_clip_backbone()
    trainable = {k for k, p in ad.named_parameters() if p.requires_grad}
    assert set(params_of(g, f"{task}.g.")) <= trainable and all("mona" in k for k in trainable if k.startswith("clip_model."))
    ad = ad.to(dev())
    y = ad(g["images"].to(dev()))
    (y * g[f"{task}.dy"].to(dev())).sum().backward()
    assert tuple(y.shape) == tuple(g[f"{task}.y"].shape) and rel(y, g[f"{task}.y"]) < TOL
    check_grads(ad.named_parameters(), params_of(g, f"{task}.g."))
