"""CPU: host-side logic of the drop-in surface — names (the checkpoint wire format), injector behaviour, parameter counts,
error behaviour, and the absence of any CPU fallback on the product path."""
import math
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOY = dict(embed_dim=128, vision_cfg=dict(img_size=32, patch_size=8, embed_dim=128, depth=3, num_heads=2),
           text_cfg=dict(vocab_size=120, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                         max_position_embeddings=40))
VARIANTS = ("baseline", "noise_aware", "freq_enhanced", "hybrid")


def test_adapters_package_exports():
    import src.adapters as A
    for name in ("BaselineMona", "NoiseAwareMona", "FreqEnhancedMona", "FreqEnhancedMonaOp", "inject_mona_variant_to_clip",
                 "inject_mona_variant_to_open_clip", "inject_lora_to_clip", "inject_lora_to_biomedclip"):
        assert hasattr(A, name)                                       # reference __init__.py:21-34, minus the names that do not exist
    for missing in ("FractionalMona", "SimplePromptTuner", "create_simple_prompt_tuner"):
        with pytest.raises(NotImplementedError):
            getattr(A, missing)


@pytest.mark.parametrize("variant", VARIANTS)
def test_mona_parameter_names_match_reference(golden, variant):
    """named_parameters() order and shapes == the reference module's (captured in the golden fixture)."""
    from src.adapters import mona as M
    g = golden(f"mona_{variant}")
    want = [k[2:] for k in g if k.startswith("p.")]
    mod = M._VARIANTS[variant](32, 8)
    got = [k for k, _ in mod.named_parameters()]
    assert got == want
    for k, p in mod.named_parameters():
        assert tuple(p.shape) == tuple(g["p." + k].shape), k
    assert float(mod.gamma[0]) == pytest.approx(1e-6) and float(mod.gammax[0]) == 1.0 and mod.dropout.p == 0.1


@pytest.mark.parametrize("variant,count", [("baseline", 1342464), ("freq_enhanced", 1343232), ("hybrid", 1356324)])
def test_mona_trainable_count_full_size(variant, count):
    """trainable elements of 12 Mona layers (b=64) on ViT-B/16 — the all-reduce payload."""
    from src.adapters import mona as M
    per_layer = sum(p.numel() for p in M._VARIANTS[variant](768, 64).parameters())
    assert 12 * per_layer == count


def test_inject_mona_open_clip_layout_and_names():
    from src.adapters import inject_mona_variant_to_open_clip, BatchFirstMonaWrapper
    from src.third_party.biomedclip.model import create_biomedclip
    model = create_biomedclip(config=TOY)
    text_before = {k: v.clone() for k, v in model.text.state_dict().items()}
    model, n = inject_mona_variant_to_open_clip(model, variant="freq_enhanced", bottleneck_dim=64, num_layers=2)
    assert n == 2
    blocks = model.visual.trunk.blocks
    assert isinstance(blocks[0].mona, BatchFirstMonaWrapper) and not hasattr(blocks[2], "mona")
    names = [k for k, _ in model.named_parameters() if "mona" in k]
    assert names[0] == "visual.trunk.blocks.0.mona.clip_mona.gamma"
    assert "visual.trunk.blocks.1.mona.clip_mona.adapter_conv.freq_filter" in names
    assert "forward" in blocks[0].__dict__ and "forward" not in blocks[2].__dict__   # instance-level patch, as the reference does
    for k, v in model.text.state_dict().items():
        assert torch.equal(v, text_before[k])                                         # the text tower is never touched
    with pytest.raises(ValueError):
        inject_mona_variant_to_open_clip(model, variant="fractional")                 # accepted by the CLI, rejected by the injector (reference :604-605)


def test_inject_mona_clip_layout_and_names():
    from src.adapters import inject_mona_variant_to_clip
    from src.third_party.openai_clip.model import CLIP
    model = CLIP(64, 32, 2, 128, 8, 16, 100, 128, 2, 2)
    model, n = inject_mona_variant_to_clip(model, variant="hybrid", bottleneck_dim=64)
    assert n == 2
    names = [k for k, _ in model.named_parameters() if "mona" in k]
    assert names[0] == "visual.transformer.resblocks.0.mona.gamma"                    # no wrapper on the sequence-first layout
    assert any(k.endswith("mona.adapter_conv.noise_estimator.3.bias") for k in names)
    with pytest.raises(ValueError):
        inject_mona_variant_to_clip(model, variant="nope")


def test_lora_modules_names_init_and_quirks():
    from src.adapters.lora import LinearLoRA, PlainMultiheadAttentionLoRA, inject_lora_to_biomedclip, inject_lora_to_clip
    from src.third_party.biomedclip.model import create_biomedclip
    from src.third_party.openai_clip.model import CLIP
    lin = torch.nn.Linear(8, 6)
    ll = LinearLoRA(lin, r=2, lora_alpha=4, dropout_rate=0.1)
    assert [k for k, _ in ll.named_parameters()] == ["weight", "bias", "w_lora_A", "w_lora_B"]
    assert tuple(ll.w_lora_A.shape) == (2, 8) and tuple(ll.w_lora_B.shape) == (6, 2)
    assert ll.scaling == pytest.approx(4 / math.sqrt(2)) and float(ll.w_lora_B.abs().max()) == 0.0 and float(ll.w_lora_A.abs().max()) > 0
    assert not ll.weight.requires_grad and ll.bias.requires_grad                      # reference quirk: the copied bias stays trainable (SURVEY C-4)
    assert torch.equal(ll.weight, lin.weight) and torch.allclose(ll.merge_BA("weight"), torch.zeros(6, 8))

    model = create_biomedclip(config=TOY)
    model, n = inject_lora_to_biomedclip(model, lora_r=4, lora_alpha=8, lora_dropout=0.1, num_layers=2)
    assert n == 2 and isinstance(model.visual.trunk.blocks[1].attn.qkv, LinearLoRA) and not isinstance(model.visual.trunk.blocks[2].attn.qkv, LinearLoRA)
    assert "visual.trunk.blocks.0.attn.qkv.w_lora_A" in dict(model.named_parameters())
    model, n = inject_lora_to_biomedclip(create_biomedclip(config=TOY), lora_r=4, tune_text_encoder=True)
    assert n == 3 + 2 and isinstance(model.text.transformer.encoder.layer[0].attention.self.query, LinearLoRA)

    clip = CLIP(64, 32, 2, 128, 8, 16, 100, 128, 2, 2)
    w = clip.visual.transformer.resblocks[0].attn.in_proj_weight.detach().clone()
    clip, n = inject_lora_to_clip(clip, lora_r=4, lora_alpha=8)
    attn = clip.visual.transformer.resblocks[0].attn
    assert n == 2 and isinstance(attn, PlainMultiheadAttentionLoRA)
    assert torch.equal(attn.k_proj.weight, w[128:256]) and isinstance(clip.transformer.resblocks[0].attn, torch.nn.MultiheadAttention)
    assert sorted(k for k, _ in attn.named_parameters())[:4] == ["k_proj.bias", "k_proj.w_lora_A", "k_proj.w_lora_B", "k_proj.weight"]


def test_no_cpu_fallback_on_product_path():
    """CPU tensors must be refused loudly: the hot path exists only in libuia_hip.so."""
    from uia_hip._lib import UiaError
    from src.adapters import FreqEnhancedMona
    from src.losses import InfoNCELoss
    m = FreqEnhancedMona(128, 64)
    with pytest.raises(UiaError):
        m(torch.randn(17, 2, 128), (4, 4))
    with pytest.raises(UiaError):
        InfoNCELoss()(torch.randn(4, 8), torch.randn(4, 8))
    with pytest.raises(RuntimeError):
        m.adapter_conv(torch.randn(2, 64, 4, 4))                                      # the spatial op never runs un-fused


def test_product_does_not_import_oracle():
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nextgen-uia_amd")
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"


def test_cosine_lr_matches_torch_scheduler():
    from uia_hip.engine import cosine_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=50, eta_min=1e-8)
    for t in range(1, 51):
        opt.step()
        sched.step()
        assert opt.param_groups[0]["lr"] == pytest.approx(cosine_lr(1e-4, 1e-8, t, 50), rel=1e-6, abs=1e-12)


def test_synthetic_tokenizer_shape_and_padding():
    from src.third_party.biomedclip.model import SyntheticTokenizer
    tok = SyntheticTokenizer(256)
    ids = tok(["breast ultrasound image with a benign tumor", "x"])
    assert tuple(ids.shape) == (2, 256) and ids.dtype == torch.long
    assert int(ids[0, 0]) == 2 and int(ids[0, 8]) == 3 and int(ids[0, 9:].abs().sum()) == 0 and int((ids[1] != 0).sum()) == 3
    assert torch.equal(ids, tok(["breast ultrasound image with a benign tumor", "x"]))


def test_finetune_cli_matches_reference_flags():
    """Flag names and defaults of reference biomedclip/finetune.py:32-111 (the CLI is part of the drop-in contract)."""
    from src.models.biomedclip.finetune import get_args
    a = get_args([])
    ref = dict(img_size=224, num_workers=8, exp="biomedclip_finetune", in_channels=3, ckpt=None, method="full", tune_text_encoder=False,
               tune_layers="all", mona_variant="freq_enhanced", mona_bottleneck=64, mona_layers=None, lora_r=16, lora_alpha=32,
               lora_dropout=0.1, lora_layers=None, temperature=0.07, seed=1, epochs=32, batch_size=64, lr=1e-4, lr_min=1e-8,
               weight_decay=0.01, beta1_adam=0.9, beta2_adam=0.95, patience=10, accumulation_steps=4, grad_clip=1.0,
               strong_augs=False, weak_augs=False)
    for k, v in ref.items():
        assert getattr(a, k) == v, k
    assert get_args(["--mona_variant", "fractional"]).mona_variant == "fractional"     # accepted here, rejected by the injector


def test_model_config_parser_accepts_dict_calls_and_nothing_executable():
    from src.utils.tools import parse_config
    cfg = parse_config("dict(embed_dim=128, vision_cfg=dict(img_size=32, act='quick_gelu', pre_norm=True), layers=[1, 2], t=(3, 4.5))")
    assert cfg == {"embed_dim": 128, "vision_cfg": {"img_size": 32, "act": "quick_gelu", "pre_norm": True}, "layers": [1, 2], "t": (3, 4.5)}
    assert parse_config("{'a': 1}") == {"a": 1}
    for bad in ("__import__('os').system('true')", "dict(a=open('x'))", "dict(**{'a': 1})", "[x for x in (1,)]"):
        with pytest.raises((ValueError, TypeError, SyntaxError)):
            parse_config(bad)


def test_gemm_tail_split_policy():
    """ops.tail_split_rows: which launches get their M tail re-cut into half-height tiles (pure host scheduling)."""
    from uia_hip import ops
    # ViT-B/16 at bs 256: 197 row panels x 3 column panels = 591 = 2*256 + 79 -> the main launch keeps 170 whole panels (510 tiles)
    assert ops.tail_split_rows(50432, 768, 256) == 170 * 256
    # N = 3072: 2364 = 9*256 + 60 -> 192 panels in the main launch, 5 panels (60 tiles -> 120 half tiles) in the tail
    assert ops.tail_split_rows(50432, 3072, 256) == 192 * 256
    # N = 2304: 1773 = 6*256 + 237: the last round is 93 % full, no split
    assert ops.tail_split_rows(50432, 2304, 256) == 50432
    # BERT rows (65 536): every N of the model is a whole number of rounds
    for n in (768, 2304, 3072):
        assert ops.tail_split_rows(65536, n, 256) == 65536
    # fewer tiles than CUs, or a single row panel: nothing to split
    assert ops.tail_split_rows(4096, 768, 256) == 4096 and ops.tail_split_rows(256, 768, 2) == 256
    for M, N, ncu in ((50432, 768, 256), (50432, 3072, 256), (32896, 1024, 256), (12345, 4096, 304)):
        m = ops.tail_split_rows(M, N, ncu)
        assert m % 256 == 0 or m == M
        if m < M:                                                       # main = whole rounds; the tail's half tiles fit one round
            tn = -(-N // 256)
            assert (m // 256 * tn) <= (-(-M // 256) * tn) // ncu * ncu
            assert -(-(M - m) // 128) * tn <= ncu + tn * 2


def test_packed_weight_kblocked_layout():
    """PackedW.kblocked(): element (n, k) of the row-major weight sits at [k // g][n][k % g], g = 64 bytes of elements."""
    import torch
    from uia_hip import ops
    for dt, g in ((torch.bfloat16, 32), (torch.float32, 16)):
        w = torch.arange(6 * 2 * g, dtype=torch.float32).view(6, 2 * g).to(dt)
        pw = ops.PackedW(w)
        kb = pw.kblocked()
        assert tuple(kb.shape) == (2, 6, g) and kb.is_contiguous() and pw.kblocked() is kb and pw.shape == w.shape and pw.dtype == dt
        for n, k in ((0, 0), (5, g - 1), (3, g), (2, 2 * g - 1)):
            assert float(kb[k // g, n, k % g]) == float(w[n, k])


def test_kblocked_activation_policy_and_views():
    """ops.kb_ok: an activation travels K-blocked only when BOTH the GEMM that writes it and the GEMM that reads it run on the ring
    kernels (M > 2048, more than 64 columns on either side, more than one 64-byte column block); ops.KBlocked: the wrapper's views share
    storage with the row-major buffer they were laid over, and element (m, k) sits at [k // g, m, k % g]."""
    from uia_hip import ops
    bf, f32 = torch.bfloat16, torch.float32
    assert ops.kb_ok(50432, 3072, 768, bf) and ops.kb_ok(65536, 768, 3072, bf) and ops.kb_ok(4096, 128, 128, f32)
    assert not ops.kb_ok(788, 3072, 768, bf)          # B = 4 full-geometry parity tests: small-M tile config
    assert not ops.kb_ok(50432, 64, 768, bf)          # the adapters' down-projection reads its rows straight from HBM (tile cfg 16)
    assert not ops.kb_ok(50432, 768, 64, bf)          # single K step: nothing to stream
    assert not ops.kb_ok(50432, 0, 768, bf)           # no GEMM consumer (a LayerNorm kernel writes / reads it row-major)
    saved, ops.KBLOCK_ACT = ops.KBLOCK_ACT, False
    try:
        assert not ops.kb_ok(50432, 3072, 768, bf)
    finally:
        ops.KBLOCK_ACT = saved
    M, K = 6, 96
    buf = torch.arange(M * K, dtype=torch.float32).view(M, K).to(bf)
    kb = ops.KBlocked.over(buf)
    assert ops.is_kb(kb) and not ops.is_kb(buf) and not ops.is_kb(buf.view(3, 2, 96))
    assert (kb.rows, kb.cols, kb.dtype, kb.t.shape) == (M, K, bf, (3, M, 32)) and kb.data_ptr() == buf.data_ptr()
    assert kb.as_rows().data_ptr() == buf.data_ptr() and kb.as_rows().shape == (M, K)
    logical = torch.randn(M, K).to(bf)
    kb.t.copy_(logical.view(M, K // 32, 32).permute(1, 0, 2))
    for m, k in ((0, 0), (5, 95), (2, 33), (3, 64)):
        assert kb.t[k // 32, m, k % 32] == logical[m, k]
    sl = kb.row_range(2, 5)
    assert sl.rows == 3 and sl.t.stride(0) == M * 32 and sl.t.data_ptr() == buf.data_ptr() + 2 * 32 * 2


def test_gemm_schedule_mirrors_are_pure_host_logic():
    """The host-side mirrors of the launcher's tile choice (uia_hip.ops): which kernel a shape runs on, where the M tail is cut, when that tail is
    split over K.  No GPU involved."""
    from uia_hip import ops
    assert ops.auto_tile_cfg(50432, 768, 768, 2) == 8 and ops.auto_tile_cfg(50432, 768, 64, 2) == 14
    assert ops.auto_tile_cfg(50432, 64, 768, 2) == 16                       # N = 64 stream kernel (epilogue not known yet: optimistic)
    assert ops.auto_tile_cfg(50432, 64, 768, 2, ops.EPI_BIAS | ops.EPI_OUTT) == 16
    assert ops.auto_tile_cfg(25216, 64, 2048, 2) == 14                      # W image beyond the stream kernel's LDS: 3-deep ring
    assert ops.auto_tile_cfg(256, 640, 768, 4) == 21 and ops.auto_tile_cfg(256, 640, 768, 2) == 3 and ops.auto_tile_cfg(2048, 4096, 768, 4) == 3
    assert ops.tail_split_rows(50432, 768, 256) == 43520 and ops.tail_split_rows(65536, 768, 256) == 65536
    assert ops.tail_split_rows(32896, 1024, 256) == 32768 and ops.tail_split_rows(32896, 4096, 256) == 32768
    assert ops.tail_k_slices(128, 1024, 4096, 2, 256) == 8 and ops.tail_k_slices(128, 1024, 3072, 2, 256) == 6
    assert ops.tail_k_slices(128, 4096, 1024, 2, 256) == 0 and ops.tail_k_slices(6912, 768, 3072, 2, 256) == 0 and ops.tail_k_slices(128, 1024, 4096, 4, 256) == 0
    assert ops.big_tile_cfg(768, 3072, 2) == 8                              # the 5-deep ring is opt-in
    m = ops.EPI_QUICK | ops.EPI_BIAS | ops.EPI_GELU | ops.EPI_OUTT
    assert m in ops._SPECIALISED and (m | ops.EPI_LNFOLD) in ops._SPECIALISED and (ops.EPI_QUICK | ops.EPI_DGELU | ops.EPI_OUTT) in ops._SPECIALISED
    assert ops.gemm_kernel_name(8, 977, __import__("torch").bfloat16)[1].endswith("Li977ELi0ELb0ELb0EE")
    # the register-staged four-wave kernels (csrc/gemm_quadv.hip) are opt-in: the knob moves the 256 x 256 bf16 launches, nothing else; masks they do not instantiate name the
    # run-time epilogue, and the three-in-flight form (28) instantiates the plain store only
    bf = __import__("torch").bfloat16
    try:
        ops.QUADV = 27
        assert ops.big_tile_cfg(768, 3072, 2) == 27 and ops.big_tile_cfg(768, 3072, 4) == 8
        ops.QUADV = 29
        assert ops.big_tile_cfg(2304, 768, 2) == 29
    finally:
        ops.QUADV = False
    assert ops.big_tile_cfg(768, 3072, 2) == 8
    assert ops.gemm_kernel_name(27, ops.EPI_OUTT, bf) == ("gemm_tn_quadv_kernel<128,2,0>", "gemm_tn_quadv_kernelILi128ELi2ELi0EE")
    assert ops.gemm_kernel_name(28, ops.EPI_BIAS | ops.EPI_OUTT, bf)[0] == "gemm_tn_quadv_kernel<-1,3,0>"
    assert ops.gemm_kernel_name(29, ops.EPI_BIAS | ops.EPI_OUTT, bf) == ("gemm_tn_quadvp_kernel<129>", "gemm_tn_quadvp_kernelILi129EE")
    assert ops.gemm_kernel_name(29, ops.EPI_BIAS | ops.EPI_RESIDT | ops.EPI_OUT32, bf)[0] == "gemm_tn_quadvp_kernel<-1>"
    assert set((27, 28, 29)) <= set(ops.RING_CFGS)


def test_committed_traffic_file_names_the_kernels_the_headline_step_runs():
    """bench.py's roofline.traffic comes from profiles/<bench.TRAFFIC_FILE> (tools/pmc_traffic.sh), keyed by the mangled instantiation name.  VERDICT
    r03: the round-3 file was generated before the last rename and bench printed traffic: null.  Host-side check that every 256 x 256 ring
    instantiation the headline step (mona, bf16, LayerNorm folded, three-byte text residual) can put on top is a key of the committed file, so a
    rename of a template argument or a new epilogue mask fails here, on the CPU, and not as a silent null on the GPU box."""
    import importlib.util
    import json
    import torch
    from uia_hip import ops
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    data = json.load(open(os.path.join(ROOT, "profiles", bench.TRAFFIC_FILE)))
    assert set(data) == {"FETCH_SIZE", "WRITE_SIZE"}
    headline = [ops.EPI_BIAS | ops.EPI_RESID_LO | ops.EPI_RESID_LN | ops.EPI_OUTT | ops.EPI_OUT_LO | ops.EPI_ROWSUM,     # text tower sub-layer sums (three-byte tensors)
                ops.EPI_BIAS | ops.EPI_OUTT | ops.EPI_LNFOLD,                                                          # QKV with the folded LayerNorm
                ops.EPI_BIAS | ops.EPI_GELU | ops.EPI_OUTT | ops.EPI_LNFOLD,                                           # fc1, frozen tower
                ops.EPI_BIAS | ops.EPI_GELU | ops.EPI_AUX_OUT | ops.EPI_OUTT | ops.EPI_LNFOLD,                         # fc1 with the stash
                ops.EPI_DGELU | ops.EPI_OUTT, ops.EPI_OUTT, ops.EPI_BIAS | ops.EPI_RESID | ops.EPI_OUT32]              # data gradients, fc2 / proj
    for mask in headline:
        # 256 x 256 tiles (cfg 8), or — fc1 and its GELU' data gradient under ops.SHORT_K_WIDE_HALF_N — the half-height tiles of cfg 14
        frags = [ops.gemm_kernel_name(cfg, mask, torch.bfloat16)[1] for cfg in (8, 14)]
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            hit = [f for f in frags if f in data[counter]]
            assert hit, (mask, frags, sorted(data[counter])[:3])
            assert all(data[counter][f]["launches"] > 0 and data[counter][f]["sum"] > 0 for f in hit)
    # the kernel bench.py names as dominant must be there under the name bench.py derives
    dom = ops.gemm_kernel_name(8, headline[0], torch.bfloat16)[1]
    assert dom in data["FETCH_SIZE"] and dom in data["WRITE_SIZE"]


def test_three_byte_gradient_tokens_travel_through_views_and_poison_other_readers():
    """functional.publish_grad3 / grad3_of: the token autograd carries for a three-byte residual gradient is one NaN expanded to the gradient's shape — found again by
    address behind any chain of views, consumed by the first lookup, and NaN to every reader that is not one of the two Functions (a topology the hand-off does not cover
    fails loudly).  _g3_partner_feeds only accepts a MonaFn / VitBlockFn output seen through view nodes."""
    import torch
    from uia_hip import functional as UF
    UF.clear_t_copies()
    hi, lo = torch.zeros(6, 4, dtype=torch.bfloat16), torch.zeros(6, 4, dtype=torch.int8)
    tok = UF.publish_grad3((2, 3, 4), torch.device("cpu"), hi, lo)
    assert tuple(tok.shape) == (2, 3, 4) and tok.stride() == (0, 0, 0) and bool(torch.isnan(tok).all())
    seen = tok.permute(1, 0, 2).permute(1, 0, 2)
    assert UF.grad3_of(torch.zeros(2, 3, 4)) is None                      # an ordinary gradient
    got = UF.grad3_of(seen)
    assert got is not None and got[0] is hi and got[1] is lo
    assert UF.grad3_of(seen) is None                                      # consumed
    one = UF.publish_grad3((1, 6, 4), torch.device("cpu"), hi, lo)          # a one-image slice: the size-1 dimension keeps a non-zero stride
    assert UF.grad3_of(one.permute(1, 0, 2)) is not None
    tok2 = UF.publish_grad3((2, 3, 4), torch.device("cpu"), hi, lo)
    assert tok2.data_ptr() != tok.data_ptr()                              # consecutive hand-offs do not share an address
    assert bool(torch.isnan(tok2 + 1.0).all())
    UF.clear_t_copies()
    assert UF.grad3_of(tok2) is None

    class VitBlockFn(torch.autograd.Function):                           # same class name as the product's Function: its backward node is "VitBlockFnBackward"
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            return g * 2

    x = torch.randn(2, 3, 4, requires_grad=True)
    y = VitBlockFn.apply(x)
    UF.set_grad_resid3(True)
    try:
        assert not UF._g3_partner_feeds(y.permute(1, 0, 2))               # outside a tower's own block loop (UF.linear_chain): a tapped output could have a second consumer
        with UF.linear_chain():
            assert UF._g3_partner_feeds(y.permute(1, 0, 2).permute(1, 0, 2))
            assert not UF._g3_partner_feeds(y + 0.0)                      # arithmetic in between would read the token
            assert not UF._g3_partner_feeds(x)
        assert not UF._g3_partner_feeds(y)                                # the scope has ended
        UF.set_grad_resid3(False)
        assert not UF._g3_partner_feeds(y)
    finally:
        UF.set_grad_resid3(False)


def test_zero_shot_prompt_ensembles_are_the_references_ten_per_class():
    """src/models/zero_shot_prompt.py exposes the reference's names (reference zero_shot.py:22 imports them) with ten prompts per class and set (reference
    zero_shot_prompt.py:2-54), and picks the ensemble by dataset name as reference zero_shot.py:168-173 does."""
    import pytest
    from src.models.zero_shot_prompt import BREAST_PROMPTS_ENSEMBLE, LN_PROMPTS_ENSEMBLE, ensemble_for
    for ens, word in ((LN_PROMPTS_ENSEMBLE, "lymph node"), (BREAST_PROMPTS_ENSEMBLE, "nodule")):
        assert sorted(ens) == ["benign", "malignant"]
        for c in ens:
            assert len(ens[c]) == 10 and len(set(ens[c])) == 10 and all(p.startswith(f"A {c} {word}") for p in ens[c])
    assert ensemble_for("BUSI") is BREAST_PROMPTS_ENSEMBLE and ensemble_for("LN-INT") is LN_PROMPTS_ENSEMBLE and ensemble_for("busi_external") is BREAST_PROMPTS_ENSEMBLE
    with pytest.raises(ValueError):
        ensemble_for("thyroid")


def test_clip_adapter_openai_layout_names_and_freeze_rule():
    """CLIPAdapter (reference src/third_party/openai_clip/clip_adapter.py:6-165): state-dict keys of the heads as in the reference (`cls_head.2/.5`, not the timm class's
    `cls_head.3`), feature_dim from the tower's width, and freeze_clip_backbone() keeping ONLY "mona" names trainable in the backbone (:138-150)."""
    import torch
    from src.adapters import inject_mona_variant_to_clip
    from src.third_party.openai_clip.clip_adapter import CLIPAdapter
    from src.third_party.openai_clip.model import CLIP
    clip = CLIP(16, 32, 2, 128, 8, 8, 50, 64, 2, 2)
    clip, n = inject_mona_variant_to_clip(clip, variant="freq_enhanced", bottleneck_dim=8)
    for task in ("seg", "cls"):
        ad = CLIPAdapter(clip, extract_layers=[0, 1], reduce_dim=64, num_classes=2, img_size=32, patch_size=8, task=task)
        keys = {k for k in ad.state_dict() if not k.startswith("clip_model.")}
        want = {f"reduces.{i}.{p}" for i in (0, 1) for p in ("weight", "bias")} | {f"blocks.{i}.{j}.{p}" for i in (0, 1) for j in (0, 1, 3) for p in ("weight", "bias")}
        want |= {"seg_head.1.weight", "seg_head.1.bias", "cls_head.2.weight", "cls_head.2.bias", "cls_head.5.weight", "cls_head.5.bias"}
        assert keys == want and ad.feature_dim == 128
        ad.freeze_clip_backbone()
        backbone = {k for k, p in ad.clip_model.named_parameters() if p.requires_grad}
        assert backbone and all("mona" in k for k in backbone)
        heads = {k for k, p in ad.named_parameters() if p.requires_grad and not k.startswith("clip_model.")}
        assert {k.split(".")[0] for k in heads} >= {"reduces", "blocks", "seg_head" if task == "seg" else "cls_head"}      # (the other head is never frozen by the reference either)
    import pytest
    with pytest.raises(ValueError):
        CLIPAdapter(clip, task="det").freeze_clip_backbone()


def test_shared_batch_ring_carries_every_batch_through_reused_slots():
    """datasets.finetune.SharedBatchRing with real loader worker processes (CPU only: no pinning): every batch of two epochs arrives in order as (marker, slot, captions),
    the slot holds exactly the samples default_collate would have stacked and the token ids the tokenizer gives for those captions, and the ring of 2·workers + 4 slots is
    recycled (more batches than slots) without a worker ever writing into a slot the consumer has not released."""
    import types
    import torch
    from src.datasets import finetune as F
    from src.third_party.biomedclip.model import SyntheticTokenizer
    args = types.SimpleNamespace(synthetic=True, synthetic_train=22 * 4, synthetic_val=8, img_size=16, seed=3, batch_size=4, num_workers=2, data_pt=None)
    tok = SyntheticTokenizer(32)
    dm = F.DataModule(args, rank=0, world=1, tokenizer=tok)
    loader = dm.train_dataloader()
    ring = loader.collate_fn
    assert isinstance(ring, F.SharedBatchRing) and ring.slots == 8 and len(loader) == 22
    dm.start_workers()
    try:
        it = loader.__dict__.pop("_uia_first_iter")
        for epoch in range(2):
            seen = []
            held = []
            for batch in (it if epoch == 0 else iter(loader)):
                assert batch[0] == ring.MARK
                slot, texts = batch[1], batch[2]
                im, ids = ring.images[slot].clone(), ring.ids[slot].clone()
                assert torch.equal(ids, tok(list(texts)))
                seen.append((im, list(texts)))
                held.append(slot)
                if len(held) > 2:                                # the consumer keeps up to three slots (copies in flight) before handing one back
                    ring.release(held.pop(0))
            for s in held:
                ring.release(s)
            assert len(seen) == 22
            # every sample of the dataset exactly once per epoch (shuffled), each image beside its own caption
            ds = dm.train
            got = {t: im_b[j] for im_b, texts in seen for j, t in enumerate(texts)}
            for i in range(len(ds)):
                img, text = ds[i]
                assert text in got and torch.equal(got[text], img)
    finally:
        dm.shutdown()


def test_openai_clip_finetune_cli_matches_the_reference_flags():
    """src/models/clip/finetune.py keeps the reference's flags and defaults (reference src/models/clip/finetune.py:27-62)."""
    from src.models.clip import finetune
    a = finetune.get_args([])
    want = dict(img_size=224, num_workers=8, strong_augs=False, weak_augs=False, mona_variant="noise_aware", exp="clip_finetune", ckpt="ckpt/ViT-B-16.pt", in_channels=3,
                mona_bottleneck=64, mona_layers=None, temperature=0.07, seed=1, epochs=1000, batch_size=64, lr=1e-4, lr_min=1e-8, weight_decay=0.01, beta1_adam=0.9,
                beta2_adam=0.95, patience=10)
    for k, v in want.items():
        assert getattr(a, k) == v, k
    assert finetune._geometry(a) == (512, 224, 12, 768, 16, 77, 49408, 512, 8, 12)
    assert finetune._geometry(finetune.get_args(["--model_config", "(64, 32, 2, 128, 8, 16, 100, 128, 2, 2)"])) == (64, 32, 2, 128, 8, 16, 100, 128, 2, 2)
