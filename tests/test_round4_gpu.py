"""-m gpu, round 4: the barrier-free attention backward (csrc/attention_bwd.hip, attn_bwd_units_kernel; uia_attn_bwd_cfg 2 / 3 / 4) against
torch's fp32 softmax attention (autograd) on the same operands — every mask kind, ragged lengths, the production head counts — and its
K-blocked inputs / outputs against its own row-major launch, bit for bit.  Reference op: /root/reference/src/adapters/lora.py:188,
src/third_party/openai_clip/model.py:197 (autograd through SDPA / nn.MultiheadAttention)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
CFGS = [1, 2, 3, 4, 5, 6, 7]


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _reference(qkv, dout, B, H, L, mask, keylen):
    D = H * 64
    qf = qkv.float().view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(True)   # [3,B,H,L,64]
    s = qf[0] @ qf[1].transpose(-1, -2) / 8.0
    if mask == "causal":
        s = s + torch.full((L, L), float("-inf"), device=dev()).triu_(1)
    if mask == "keypad":
        ar = torch.arange(L, device=dev())
        s = s.masked_fill(ar[None, None, None, :] >= keylen[:, None, None, None], float("-inf"))
    ref = torch.softmax(s, -1) @ qf[2]
    ref.backward(dout.float().view(B, L, H, 64).permute(0, 2, 1, 3))
    return qf.grad.permute(1, 3, 0, 2, 4).reshape(B * L, 3 * D)


@pytest.mark.parametrize("cfg", CFGS)
@pytest.mark.parametrize("L,mask", [(197, "none"), (77, "causal"), (256, "keypad"), (17, "none"), (50, "keypad"), (257, "none"), (130, "causal"),
                                    (16, "none"), (2, "none"), (208, "keypad"), (272, "none"), (33, "causal")])
def test_attention_bwd_every_kernel_configuration(cfg, L, mask):
    from uia_hip import ops
    if cfg == 5 and L > 240:
        pytest.skip("the persistent kernel holds at most 240 tokens in LDS")
    torch.manual_seed(2 + L)
    B, H, D = 3, 4, 256
    qkv = (torch.randn(B * L, 3 * D, device=dev()) * 1.5).bfloat16()
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    keylen = torch.tensor([L, max(1, L // 3), max(2, L - 5)], device=dev(), dtype=torch.int32) if mask == "keypad" else None
    out = torch.empty(B * L, D, device=dev(), dtype=torch.bfloat16)
    lse = torch.empty(B, H, L, device=dev())
    ops.attn_fwd(q, k, v, out, B, H, L, lse=lse, mask=mask, keylen=keylen)
    dout = torch.randn(B * L, D, device=dev()).bfloat16()
    g = _reference(qkv, dout, B, H, L, mask, keylen)
    dqkv = torch.full((B * L, 3 * D), float("nan"), device=dev(), dtype=torch.bfloat16)     # every element must be written
    ops.attn_bwd(q, k, v, out, dout, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, L, mask=mask, keylen=keylen, cfg=cfg)
    assert bool(torch.isfinite(dqkv.float()).all())
    for i, name in enumerate("qkv"):
        assert rel(dqkv[:, i * D:(i + 1) * D], g[:, i * D:(i + 1) * D]) < 2.5e-2, name
    if mask == "keypad":        # padded keys receive exactly zero gradient
        for bi in range(B):
            kl = int(keylen[bi])
            assert float(dqkv[bi * L + kl:(bi + 1) * L, D:].float().abs().max() if kl < L else 0.0) == 0.0


@pytest.mark.parametrize("cfg", [2, 3, 4, 5])
@pytest.mark.parametrize("L,mask", [(197, "none"), (256, "keypad"), (257, "none")])
def test_attention_bwd_production_head_count(cfg, L, mask):
    """B = 8, H = 12 (ViT-B / BERT-base head count; 257 = ViT-L/14 tokens)"""
    from uia_hip import ops
    if cfg == 5 and L > 240:
        pytest.skip("the persistent kernel holds at most 240 tokens in LDS")
    torch.manual_seed(7 + L)
    B, H = 8, 12
    D = H * 64
    qkv = (torch.randn(B * L, 3 * D, device=dev()) * 1.2).bfloat16()
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    keylen = torch.randint(24, 129, (B,), device=dev(), dtype=torch.int32) if mask == "keypad" else None
    out = torch.empty(B * L, D, device=dev(), dtype=torch.bfloat16)
    lse = torch.empty(B, H, L, device=dev())
    ops.attn_fwd(q, k, v, out, B, H, L, lse=lse, mask=mask, keylen=keylen)
    dout = torch.randn(B * L, D, device=dev()).bfloat16()
    g = _reference(qkv, dout, B, H, L, mask, keylen)
    dqkv = torch.full((B * L, 3 * D), float("nan"), device=dev(), dtype=torch.bfloat16)
    ops.attn_bwd(q, k, v, out, dout, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, L, mask=mask, keylen=keylen, cfg=cfg)
    for i, name in enumerate("qkv"):
        assert rel(dqkv[:, i * D:(i + 1) * D], g[:, i * D:(i + 1) * D]) < 2.5e-2, name
    # the legacy kernel on the same operands: both are bf16 roundings of the same fp32 sums
    d1 = torch.empty_like(dqkv)
    ops.attn_bwd(q, k, v, out, dout, lse, d1[:, :D], d1[:, D:2 * D], d1[:, 2 * D:], B, H, L, mask=mask, keylen=keylen, cfg=1)
    assert rel(dqkv, d1) < 1.5e-2


@pytest.mark.parametrize("B,H,L,mask", [(43, 12, 197, "none"), (37, 7, 130, "keypad"), (65, 4, 77, "causal"), (33, 12, 197, "keypad")])
def test_attention_bwd_persistent_kernel_walks_several_heads(B, H, L, mask):
    """More heads than CUs: a workgroup of the persistent kernel (cfg 5) takes two or three heads in turn — staging of the next head under the
    sweeps of the current one, double-buffered statistics, per-head key lengths — and one workgroup fewer heads than the others where the
    count does not divide.  Against torch autograd AND against the one-head-per-workgroup kernel (cfg 2), which computes the same sums."""
    from uia_hip import ops
    torch.manual_seed(B + L)
    D = H * 64
    qkv = (torch.randn(B * L, 3 * D, device=dev()) * 1.2).bfloat16()
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    keylen = torch.randint(1, L + 1, (B,), device=dev(), dtype=torch.int32) if mask == "keypad" else None
    out = torch.empty(B * L, D, device=dev(), dtype=torch.bfloat16)
    lse = torch.empty(B, H, L, device=dev())
    ops.attn_fwd(q, k, v, out, B, H, L, lse=lse, mask=mask, keylen=keylen)
    dout = torch.randn(B * L, D, device=dev()).bfloat16()
    g = _reference(qkv, dout, B, H, L, mask, keylen)
    d5 = torch.full((B * L, 3 * D), float("nan"), device=dev(), dtype=torch.bfloat16)
    ops.attn_bwd(q, k, v, out, dout, lse, d5[:, :D], d5[:, D:2 * D], d5[:, 2 * D:], B, H, L, mask=mask, keylen=keylen, cfg=5)
    assert bool(torch.isfinite(d5.float()).all())
    for i, name in enumerate("qkv"):
        assert rel(d5[:, i * D:(i + 1) * D], g[:, i * D:(i + 1) * D]) < 2.5e-2, name
    d2 = torch.empty_like(d5)
    ops.attn_bwd(q, k, v, out, dout, lse, d2[:, :D], d2[:, D:2 * D], d2[:, 2 * D:], B, H, L, mask=mask, keylen=keylen, cfg=2)
    assert torch.equal(d5, d2)        # same fragments, same order of sums: only where the operands come from differs


@pytest.mark.parametrize("cfg", [2, 3, 4, 5])
@pytest.mark.parametrize("L,mask", [(197, None), (256, "keypad"), (50, "causal")])
def test_attention_bwd_kblocked_tensors(cfg, L, mask):
    """O read K-blocked, the fused dq / dk / dv written K-blocked: the same values as the row-major launch of the same kernel, bit for bit;
    and a second launch reproduces the first (no atomics, the unit queue only changes which wave computes what)"""
    from uia_hip import ops
    from tests.test_lnfold_gpu import _from_kb
    if cfg == 5 and L > 240:
        pytest.skip("the persistent kernel holds at most 240 tokens in LDS")
    B, H, D = 3, 4, 256
    g = torch.Generator().manual_seed(L)
    qkv = (torch.randn(B * L, 3 * D, generator=g) * 0.5).to(dev()).bfloat16()
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    keylen = torch.tensor([L, max(1, L // 3), max(1, L - 7)], dtype=torch.int32, device=dev()) if mask == "keypad" else None
    out = torch.empty(B * L, D, device=dev(), dtype=torch.bfloat16)
    lse = torch.empty(B, H, L, device=dev())
    ops.attn_fwd(q, k, v, out, B, H, L, lse=lse, mask=mask, keylen=keylen)
    big = ops.kb_empty(B * L + 40, D, torch.bfloat16, dev())
    okb = big.row_range(8, 8 + B * L)
    ops.attn_fwd(q, k, v, okb, B, H, L, lse=torch.empty_like(lse), mask=mask, keylen=keylen)
    do = torch.randn(B * L, D, generator=g).to(dev()).bfloat16()
    dqkv, again = torch.empty_like(qkv), torch.empty_like(qkv)
    ops.attn_bwd(q, k, v, out, do, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, L, mask=mask, keylen=keylen, cfg=cfg)
    ops.attn_bwd(q, k, v, out, do, lse, again[:, :D], again[:, D:2 * D], again[:, 2 * D:], B, H, L, mask=mask, keylen=keylen, cfg=cfg)
    dkb = ops.kb_empty(B * L, 3 * D, torch.bfloat16, dev())
    ops.attn_bwd(q, k, v, okb, do, lse, dkb, None, None, B, H, L, mask=mask, keylen=keylen, cfg=cfg)
    assert torch.equal(again, dqkv)
    assert torch.equal(_from_kb(dkb), dqkv)


def test_parity_at_a_quarter_of_the_benchmark_batch_vs_oracle():
    """configs[1] at B = 64 (12 608 image-token rows, 16 384 text positions: every launch on the kernels and tile configs the benchmark
    batch runs, ring GEMMs, folded LayerNorms, K-blocked activations), bf16, against oracle/train_ref's arithmetic on the host cores:
    features AND contrastive logits (the quantity BASELINE.json's north_star bounds) inside 1e-2, loss, whole gradient direction.
    (tools/parity_at_bench_batch.py holds the B = 256 run: profiles/r04_*_parity_bench_batch.json.)  ~20 s of CPU work."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("parity_at_bench_batch", os.path.join(root, "tools", "parity_at_bench_batch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.run_case(64, "freq_enhanced", False, 16, max(1, min(32, os.cpu_count() or 1)))
    assert r["image_features_rel"] < 1e-2 and r["text_features_rel"] < 1e-2, r
    assert r["logits_rel"] < 1e-2, r
    assert abs(r["loss"] - r["loss_ref"]) < 2e-3 * max(1.0, abs(r["loss_ref"])), r
    assert r["grad_cosine"] > 0.99 and r["grad_rel_l2"] < 0.15, r
    assert not r["ln_fold_guard_tripped"]
    assert r["grad_resid3"]                       # the step ran as engine.contrastive_step runs it: three-byte residual gradients between the backward Functions


def test_clipseg_prompt_features_cache_hits_and_invalidates():
    """One fixed prompt over the batch (reference src/models/clipseg/segmentation.py:142): the adapter keeps the frozen prompt tower's features
    per (token ids, tower weight versions, compute dtype).  Same logits as running the tower, bit for bit; another prompt, an in-place weight
    update or another compute dtype recompute."""
    from uia_hip import functional as UF
    from src.third_party.openai_clip.model import CLIP
    from src.third_party.openai_clip.clipseg_adapter import CLIPSegAdapter, CLIPSegDecoder
    UF.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(5)
    clip = CLIP(64, 64, 3, 128, 16, 16, 100, 128, 2, 2).eval()
    model = CLIPSegAdapter(clip, decoder=CLIPSegDecoder(vision_hidden=128, projection_dim=64, extract_layers=(0, 1, 2), intermediate=128, patch_size=16))
    model.freeze_clip_backbone()
    model = model.to(dev())
    images = torch.rand(4, 3, 64, 64, device=dev())
    ids = torch.randint(1, 90, (1, 16), device=dev())
    ids[0, 0], ids[0, 9] = 98, 99
    ids = ids.repeat(4, 1)
    calls = []
    enc = clip.encode_text
    clip.encode_text = lambda t: (calls.append(1), enc(t))[1]
    with torch.no_grad():
        a = model(images, input_ids=ids)
        b = model(images, input_ids=ids)
        assert len(calls) == 1 and torch.equal(a, b)
        model.cache_prompt_features = False
        c = model(images, input_ids=ids)
        model.cache_prompt_features = True
        assert len(calls) == 2 and torch.equal(a, c)
        ids2 = ids.clone(); ids2[:, 3] = 7
        model(images, input_ids=ids2)
        assert len(calls) == 3                                            # another prompt
        clip.token_embedding.weight.mul_(1.5)                             # an in-place update of a tower tensor
        d = model(images, input_ids=ids2)
        assert len(calls) == 4
        model(images, input_ids=ids2)
        assert len(calls) == 4
        assert not torch.equal(d[:1], a[:1])


def _kb(t):
    from uia_hip import ops
    g = 64 // t.element_size()
    M, K = t.shape
    return ops.KBlocked(t.view(M, K // g, g).permute(1, 0, 2).contiguous())


def test_three_byte_conversion_helpers_roundtrip():
    """float -> (bf16 hi, int8 lo) -> float keeps 15 mantissa bits (2^-16 relative), across exponent boundaries and signs"""
    from uia_hip import ops
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(4096, generator=g) * 10 ** torch.randint(-6, 6, (4096,), generator=g).float(),
                   torch.tensor([0.0, -0.0, 1.0, -1.0, 1.9999999, 2.0, -255.99998, 3.3895314e38, 1e-30])]).to(dev()).view(1, -1)
    x = torch.nn.functional.pad(x, (0, (-x.shape[1]) % 8))
    hi, lo = ops.float_to_three_byte(x)
    back = ops.three_byte_to_float(hi, lo)
    err = ((back - x).abs() / x.abs().clamp_min(1e-37)).max()
    assert float(err) <= 2.0 ** -15 and bool((back.sign() == x.sign()).all() | (x == 0).all())


@pytest.mark.parametrize("M,K,kb_hi,spec", [(4100, 768, True, True), (4100, 768, False, True), (65536, 768, True, True), (4100, 3072, True, False), (2560, 64, False, False)])
def test_gemm_three_byte_residual_in_and_out(M, K, kb_hi, spec):
    """uia_gemm_desc.resid_lo8 / out_lo8: the residual arrives as a bf16 hi plane (K-blocked or row-major) + low bytes and has the deferred
    LayerNorm applied, the result leaves as hi plane + low bytes (+ row sums).  Against torch on the reconstructed operands, and against the
    fp32-residual / fp32-output launch of the same GEMM (the compile-time mask of the text tower's sub-layer sums when `spec`, else the
    run-time epilogue: no bias, fp32 output beside the low bytes)."""
    from uia_hip import ops
    torch.manual_seed(M + K)
    N = 768
    a = torch.randn(M, K, device=dev()).bfloat16()
    w = (torch.randn(N, K, device=dev()) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device=dev()) if spec else None
    prev = torch.randn(M, N, device=dev()) * 3 + 0.5
    hi, lo = ops.float_to_three_byte(prev)
    prev3 = ops.three_byte_to_float(hi, lo)                                   # what the kernel must reconstruct, exactly
    assert float((prev3 - prev).abs().max() / prev.abs().max()) < 2.0 ** -15
    stats = torch.stack([prev3.mean(1), (prev3.var(1, unbiased=False) + 1e-12).rsqrt()], 1).contiguous()
    lw, lb = torch.randn(N, device=dev()), torch.randn(N, device=dev())
    ref = a.float() @ w.float().T + (bias if bias is not None else 0) + ((prev3 - stats[:, :1]) * stats[:, 1:]) * lw + lb
    out_t = ops.kb_empty(M, N, torch.bfloat16, dev()) if M > 2048 and spec else torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    # low bytes: row-major, or in 64-column blocks beside a K-blocked hi plane (whole 128-byte lines per 16-row pass)
    out_lo = ops.kb_empty(M, N, torch.int8, dev()) if kb_hi else torch.full((M, N), 99, device=dev(), dtype=torch.int8)
    sums = torch.zeros(M, 2, device=dev(), dtype=torch.int64) if spec else None
    o32 = None if spec else torch.empty(M, N, device=dev())
    ops.gemm(a, w, bias=bias, resid3=(_kb(hi), _kb(lo)) if kb_hi else (hi, lo), resid_ln=(stats, lw, lb), out_t=out_t, out_lo=out_lo, rowsum=sums, out32=o32)
    got = ops.three_byte_to_float(out_t, out_lo)
    assert rel(got, ref) < 3e-5                                               # fp32 accumulation order + 2^-16 of the format
    if o32 is not None:
        assert rel(o32, ref) < 2e-5 and float(((got - o32).abs() / o32.abs().clamp_min(1e-3)).max()) < 2.0 ** -14
    # the fp32 form of the same launch
    o_ref32 = torch.empty(M, N, device=dev())
    ops.gemm(a, w, bias=bias, resid=prev3.contiguous(), resid_ln=(stats, lw, lb), out32=o_ref32)
    assert float(((got - o_ref32).abs() / o_ref32.abs().clamp_min(1e-3)).max()) < 2.0 ** -14
    hi_rows = out_t.t.permute(1, 0, 2).reshape(M, N) if ops.is_kb(out_t) else out_t
    assert torch.equal(hi_rows, o_ref32.bfloat16())                           # the hi plane IS the rounded result (the next GEMM's A operand)
    if sums is not None:
        s = ops.rowsum_to_float(sums)
        assert rel(s[:, 0], o_ref32.sum(1)) < 1e-4 and rel(s[:, 1], (o_ref32 * o_ref32).sum(1)) < 1e-4


def test_gemm_three_byte_arguments_are_validated():
    from uia_hip import ops
    from uia_hip._lib import UiaError
    a = torch.zeros(4100, 128, device=dev(), dtype=torch.bfloat16)
    w = torch.zeros(256, 128, device=dev(), dtype=torch.bfloat16)
    hi, lo = torch.zeros(4100, 256, device=dev(), dtype=torch.bfloat16), torch.zeros(4100, 256, device=dev(), dtype=torch.int8)
    out = torch.empty(4100, 256, device=dev(), dtype=torch.bfloat16)
    with pytest.raises(UiaError):
        ops.gemm(a, w, resid3=(hi, lo), resid=torch.zeros(4100, 256, device=dev()), out_t=out)          # two residuals
    with pytest.raises(UiaError):
        ops.gemm(a, w, out32=torch.empty(4100, 256, device=dev()), out_lo=lo)                             # low bytes without a hi plane
    with pytest.raises(UiaError):
        ops.gemm(a[:256], w, resid3=(hi[:256], lo[:256]), out_t=out[:256])                               # small M: not a ring tile config
    with pytest.raises(UiaError):
        ops.gemm(a.float(), w.float(), resid3=(hi, lo), out_t=out.float())                               # fp32 operands


@pytest.mark.parametrize("kb", [False, True])
def test_layernorm_bwd_on_three_byte_tensors(kb):
    """uia_layernorm_bwd3: x and dres arrive as (bf16 hi plane, low bytes), dx leaves as (T copy, low bytes): the same result as the fp32 launch
    on the reconstructed operands — the T copies bit for bit, the reconstructed dx within 2^-15."""
    from uia_hip import ops
    torch.manual_seed(9)
    M, D = 4100, 768
    x = torch.randn(M, D, device=dev()) * 2 + 0.3
    dres = torch.randn(M, D, device=dev())
    dy = torch.randn(M, D, device=dev()).bfloat16()
    gamma = torch.randn(D, device=dev())
    xh, xl = ops.float_to_three_byte(x)
    rh, rl = ops.float_to_three_byte(dres)
    x3, r3 = ops.three_byte_to_float(xh, xl).contiguous(), ops.three_byte_to_float(rh, rl).contiguous()
    dx_ref, dxt_ref = torch.empty(M, D, device=dev()), torch.empty(M, D, device=dev(), dtype=torch.bfloat16)
    ops.layernorm_bwd(dy, x3, gamma, 1e-6, dres=r3, dx32=dx_ref, dx_t=dxt_ref)
    dxt, dlo = torch.empty_like(dxt_ref), torch.empty(M, D, device=dev(), dtype=torch.int8)
    ops.layernorm_bwd(dy, (_kb(xh) if kb else xh, xl), gamma, 1e-6, dres=(rh, rl), dx_t=dxt, dx_lo=dlo)
    assert torch.equal(dxt, dxt_ref)
    got = ops.three_byte_to_float(dxt, dlo)
    assert float(((got - dx_ref).abs() / dx_ref.abs().clamp_min(1e-3)).max()) < 2.0 ** -14
    # mixed: fp32 x, three-byte dres, fp32 + T output (the block's first LayerNorm)
    dx2, dxt2 = torch.empty(M, D, device=dev()), torch.empty(M, D, device=dev(), dtype=torch.bfloat16)
    ops.layernorm_bwd(dy, x3, gamma, 1e-6, dres=(rh, rl), dx32=dx2, dx_t=dxt2)
    assert torch.equal(dx2, dx_ref) and torch.equal(dxt2, dxt_ref)


def test_block_internal_three_byte_tensors_match_fp32_form():
    """UF.set_block_resid3 (opt-in): x1 / dx1 of a frozen pre-LN block as three-byte tensors between the GEMM epilogues and the LayerNorm backward —
    features and adapter gradients against the default (fp32 + T copy) on a batch that reaches the ring kernels."""
    from uia_hip import functional as UF
    from tests.test_lnfold_gpu import _toy
    from src.losses import InfoNCELoss
    model = _toy()
    g = torch.Generator().manual_seed(5)
    images = torch.rand(160, 3, 32, 32, generator=g).to(dev())            # 160 x 17 tokens = 2720 rows > 2048
    ids = torch.randint(4, 120, (160, 12), generator=g)
    ids[:, 0] = 2
    ids = ids.to(dev())

    def run(flag):
        UF.set_compute_dtype(torch.bfloat16)
        UF.set_block_resid3(flag)
        UF.set_dropout_seed(5)
        for p in model.parameters():
            p.grad = None
        fi, ft = model.encode_image(images), model.encode_text(ids)
        InfoNCELoss(0.07)(fi, ft).backward()
        UF.clear_t_copies()
        return fi.detach().clone(), torch.cat([p.grad.flatten() for p in model.parameters() if p.requires_grad]).clone()

    try:
        f0, g0 = run(False)
        f1, g1 = run(True)
    finally:
        UF.set_block_resid3(False)
    assert rel(f1, f0) < 5e-3                                              # 15 stored mantissa bits against 24; bf16 results re-roll their last bits on any 1e-5 change upstream
    assert float(torch.nn.functional.cosine_similarity(g1, g0, dim=0)) > 0.999


# ------------------------------------------------------------------------------------------------ four waves of 128 x 128 (tile cfg 25, csrc/gemm_quad.hip)
@pytest.mark.parametrize("M,N,K", [(65536, 768, 768), (4100, 2304, 3072), (2305, 776, 64), (257, 264, 128)])
def test_four_wave_gemm_equals_the_ring_kernel_bit_for_bit(M, N, K):
    """Tile cfg 25 runs the 256 x 256 tile on four waves of 128 x 128 (one per SIMD, accumulators pinned to the AGPRs, free-running K loop) and the
    LDS-patch epilogue once per 64-column half.  Every output element is the same chain of MFMA K steps followed by the same epilogue arithmetic
    as on the eight-wave ring kernel: plain store, K-blocked operands and result, three-byte residual in / three-byte result + row sums out, and
    the K extension must all agree with cfg 8 bit for bit (ragged M / N included)."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    dt = torch.bfloat16
    a = torch.randn(M, K, generator=g).to(dev()).to(dt)
    w = ops.PackedW((torch.randn(N, K, generator=g) * K ** -0.5).to(dev()).to(dt))
    bias = torch.randn(N, generator=g).to(dev())
    outs = {}
    for cfg in (8, 25):
        o = torch.full((M, N), float("nan"), device=dev(), dtype=dt)
        ops.gemm(a, w, bias=bias, act="gelu", out_t=o, tile_cfg=cfg)
        outs[cfg] = o
    assert torch.equal(outs[8], outs[25]) and not torch.isnan(outs[25].float()).any()
    assert rel(outs[25], torch.nn.functional.gelu(a.float() @ w.row.float().T + bias)) < 1e-2
    if N % 64 == 0 and K % 32 == 0 and M > 2048:
        # K-blocked A and result; three-byte residual (deferred LayerNorm) in, three-byte result + row sums out
        prev = torch.randn(M, N, generator=g).to(dev()) * 3 + 0.5
        hi, lo = ops.float_to_three_byte(prev)
        prev3 = ops.three_byte_to_float(hi, lo)
        stats = torch.stack([prev3.mean(1), (prev3.var(1, unbiased=False) + 1e-12).rsqrt()], 1).contiguous()
        lw, lb = torch.randn(N, generator=g).to(dev()), torch.randn(N, generator=g).to(dev())
        res = {}
        for cfg in (8, 25):
            out_t, out_lo = ops.kb_empty(M, N, dt, dev()), ops.kb_empty(M, N, torch.int8, dev())
            sums = torch.zeros(M, 2, device=dev(), dtype=torch.int64)
            ops.gemm(_kb(a), w, bias=bias, resid3=(_kb(hi), _kb(lo)), resid_ln=(stats, lw, lb), out_t=out_t, out_lo=out_lo, rowsum=sums, tile_cfg=cfg)
            res[cfg] = (out_t.t.clone(), out_lo.t.clone(), sums)
        for x, y in zip(res[8], res[25]):
            assert torch.equal(x, y)
    if K >= 96 and N % 256 == 0:
        # K extension: the last 64 columns of the K loop come from a second operand (LoRA rank update)
        from uia_hip import functional as UF
        D = K
        ws = [torch.nn.Parameter((torch.randn(N // 3 if N % 768 == 0 else N, D, generator=g) * D ** -0.5).to(dev()), requires_grad=False) for _ in range(3 if N % 768 == 0 else 1)]
        Bs = [torch.nn.Parameter((torch.randn(wi.shape[0], 16, generator=g) * 0.1).to(dev())) for wi in ws]
        ext = UF.WEIGHTS.get_lora_ext(tuple(ws), tuple(Bs), 2.0, dt)
        t_all = torch.zeros(len(ws), M, 64, device=dev(), dtype=dt)
        t_all[:, :, :16] = torch.randn(len(ws), M, 16, generator=g).to(dev()).to(dt)
        ys = {}
        for cfg in (8, 25):
            y = torch.full((M, N), float("nan"), device=dev(), dtype=dt)
            ops.gemm(a, ext, bias=bias, out_t=y, a2=(t_all, ws[0].shape[0]) if len(ws) > 1 else (t_all[0], 0), tile_cfg=cfg)
            ys[cfg] = y
        assert torch.equal(ys[8], ys[25]) and not torch.isnan(ys[25].float()).any()


# ------------------------------------------------------------------------------------------------ LoRA: three rank terms in one pass (csrc/lora_rank.hip)
@pytest.mark.parametrize("M,N,n,p", [(32896, 1024, 3, 0.25), (1000, 768, 3, 0.1), (517, 1024, 2, 0.0), (4100, 256, 1, 0.5)])
def test_lora_rank_update_equals_the_sum_of_masked_rank_terms(M, N, n, p):
    """uia_lora_rank_update: out += sum_i mask_i * (alpha * Q_i @ W_i.T) / (1 - p) with mask_i the mask uia_dropout draws for an [M, N] tensor from seed_i
    (reference lora.py:78-90: one nn.Dropout per LinearLoRA, so q, k and v of a block each have their own).  Against the fp32 rank terms pushed through
    uia_dropout with the same seeds; and, for one source, against the K = 64 stream launch it replaces (tile cfg 23, same mask)."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + n)
    dt = torch.bfloat16
    q_all = torch.randn(n, M, 64, generator=g).to(dev()).to(dt)
    ws = [(torch.randn(N, 64, generator=g) * 0.1).to(dev()).to(dt) for _ in range(n)]
    out0 = torch.randn(M, N, generator=g).to(dev()).to(dt)
    seeds = [1234567 + 77 * i for i in range(n)]
    alpha = 2.0
    want = out0.float()
    for i in range(n):
        u = alpha * (q_all[i].float() @ ws[i].float().T)
        if p > 0:
            ud = torch.empty_like(u)
            ops.dropout(u.contiguous(), ud, p, seeds[i])
            u = ud
        want = want + u
    out = out0.clone()
    ops.lora_rank_update(q_all, ws, out, alpha, p, seeds)
    torch.cuda.synchronize()
    assert not torch.isnan(out.float()).any()
    assert float((out.float() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max())           # one bf16 rounding of the sum
    if p > 0:                                                                                        # the masks are the seeds': dropped positions keep out0 exactly
        u0 = torch.empty(M, N, device=dev())
        ops.dropout(torch.ones(M, N, device=dev()), u0, p, seeds[0])
        if n == 1:
            assert torch.equal(out[u0 == 0], out0[u0 == 0])
    if n == 1:
        ref = out0.clone()
        ops.gemm(q_all[0], ws[0], alpha=alpha, resid_t=ref, out_t=ref, drop=("acc", p, seeds[0]) if p > 0 else None, tile_cfg=23)
        assert float((out.float() - ref.float()).abs().max()) <= 2.0 ** -7 * float(ref.float().abs().max())
    with pytest.raises(ops.UiaError):
        ops.lora_rank_update(q_all, ws, out[:, :N - 64], alpha, p, seeds)


# ------------------------------------------------------------------------------------------------ head dim 16 on the matrix cores (csrc/attention_dh16.hip)
@pytest.mark.parametrize("B,H,L", [(2, 4, 485), (3, 2, 16), (1, 4, 100), (2, 1, 512), (5, 4, 33), (2, 4, 1), (1, 1, 17)])
def test_head_dim_16_attention_on_mfma_vs_torch(B, H, L):
    """CLIPSeg decoder attention (four heads of 16, 485 tokens; clipseg_adapter.py:73-98): the bf16 forward and backward run on v_mfma_f32_16x16x16_bf16
    (csrc/attention_dh16.hip) instead of the scalar kernels.  Output, log-sum-exp and the three gradients against fp32 torch on the same bf16 inputs —
    the tolerance is the bf16 rounding of P / dS as MFMA operands; ragged lengths exercise the key mask of the last tile and the padding queries."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + H * 100 + L)
    D = H * 16
    dt = torch.bfloat16
    qkv = (torch.randn(B * L, 3 * D, generator=g) * 0.8).to(dev()).to(dt)
    do = torch.randn(B * L, D, generator=g).to(dev()).to(dt)
    out = torch.full((B * L, D), float("nan"), device=dev(), dtype=dt)
    lse = torch.full((B, H, L), float("nan"), device=dev())
    ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, B, H, L, lse=lse, dh=16)
    qf, kf, vf = [t.float().view(B, L, H, 16).permute(0, 2, 1, 3).detach().requires_grad_(True) for t in (qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:])]
    s = (qf @ kf.transpose(-1, -2)) * 0.25
    ref = torch.softmax(s, -1) @ vf
    ref_rows = ref.permute(0, 2, 1, 3).reshape(B * L, D)
    assert not torch.isnan(out.float()).any() and not torch.isnan(lse).any()
    assert rel(out, ref_rows) < 8e-3
    assert float((lse - torch.logsumexp(s, -1)).abs().max()) < 2e-3
    dqkv = torch.full((B * L, 3 * D), float("nan"), device=dev(), dtype=dt)
    ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, do, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, L, dh=16)
    ref.backward(do.float().view(B, L, H, 16).permute(0, 2, 1, 3))
    assert not torch.isnan(dqkv.float()).any()
    for i, t in enumerate((qf, kf, vf)):
        want = t.grad.permute(0, 2, 1, 3).reshape(B * L, D)
        assert rel(dqkv[:, i * D:(i + 1) * D], want) < 1.5e-2, ("dq", "dk", "dv")[i]


# ------------------------------------------------------------------------------------------------ LoRA input dropout regenerated by the dA launch
@pytest.mark.parametrize("M,K,p", [(32896, 1024, 0.1), (1000, 768, 0.25), (130, 64, 0.5)])
def test_wgrad_with_regenerated_dropout_equals_wgrad_on_the_dropped_rows(M, K, p):
    """uia_wgrad_drop: dA = s * q.T @ dropout(x) with the mask regenerated from the seed while x is staged (reference lora.py:82-87) must equal, bit for
    bit, uia_wgrad on the rows the N = 64 stream kernel writes out as its by-product (same mask, same rounding of the kept values, same MFMA order)."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(M + K)
    dt = torch.bfloat16
    x = torch.randn(M, K, generator=g).to(dev()).to(dt)
    q = torch.randn(M, 64, generator=g).to(dev()).to(dt)
    a_w = (torch.randn(64, K, generator=g) * 0.05).to(dev()).to(dt)
    seed = 987654321
    t, xd = torch.empty(M, 64, device=dev(), dtype=dt), torch.empty_like(x)
    if M > 2048:
        ops.gemm(x, a_w, out_t=t, drop=("a", p, seed, xd))                   # the forward's launch and its by-product
    else:
        ops.dropout(x, xd, p, seed)
    kept = (xd != 0).float().mean().item()
    assert abs(kept - (1 - p)) < 0.02
    for r in (16, 64):
        want = torch.zeros(r, K, device=dev())
        ops.wgrad(q, xd, want, alpha=2.0)
        got = torch.zeros(r, K, device=dev())
        ops.wgrad(q, x, got, alpha=2.0, drop=(p, seed))
        torch.cuda.synchronize()
        assert float(want.abs().max()) > 0
        # the M chunks of a launch meet in fp32 atomics: the order is not fixed, the sum is the same up to the association
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    with pytest.raises(ops.UiaError):
        ops.wgrad(q, x[:, :K // 2], torch.zeros(16, K // 2, device=dev()), drop=(p, seed))     # a column window of a wider tensor: not this wrapper's form


def test_lora_block_gradients_do_not_depend_on_where_the_dropout_mask_is_applied():
    """LoraAttnHalfFn with ops.LORA_REGEN_DROP (the forward keeps the un-dropped LayerNorm output once; the four dA launches regenerate their masks) against
    the form that writes the four dropped inputs out: same seeds -> same loss, same input gradient, the same factor gradients."""
    from uia_hip import functional as UF
    from uia_hip import ops
    from src.third_party.openai_clip.model import ResidualAttentionBlock
    from src.adapters.lora import PlainMultiheadAttentionLoRA
    UF.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(5)
    D, H, B, L = 256, 4, 40, 65
    blk = ResidualAttentionBlock(D, H).to(dev())
    for q in blk.parameters():
        q.requires_grad_(False)
    blk.attn = PlainMultiheadAttentionLoRA(blk.attn, r=16, lora_alpha=32, dropout_rate=0.2).to(dev())
    for k, q in blk.named_parameters():
        q.requires_grad_("lora" in k.lower())
        if "lora_B" in k:
            torch.nn.init.normal_(q, std=0.05)
    blk.train()
    x = torch.randn(L, B, D, device=dev(), requires_grad=True)
    outs = []
    for regen in (True, False):
        ops.LORA_REGEN_DROP = regen
        UF.set_dropout_seed(1234)
        for q in blk.parameters():
            q.grad = None
        x.grad = None
        y = blk(x)
        (y.float() ** 2).mean().backward()
        outs.append((y.detach().clone(), x.grad.clone(), {k: q.grad.clone() for k, q in blk.named_parameters() if q.grad is not None}))
    ops.LORA_REGEN_DROP = True
    (y0, gx0, g0), (y1, gx1, g1) = outs
    assert torch.equal(y0, y1) and torch.equal(gx0, gx1)
    for k in g0:
        assert float((g0[k] - g1[k]).abs().max()) <= 1e-5 * float(g1[k].abs().max() + 1e-12), k


# ------------------------------------------------------------------------------------------------ Mona: project1's data gradient inside the pre-norm backward
@pytest.mark.parametrize("M,kb", [(50432, True), (50432, False), (1000, False), (37, False)])
def test_mona_pre_bwd_with_the_k64_data_gradient_inside_equals_the_two_launches(M, kb):
    """uia_mona_pre_bwd_du computes du = dt·W1 per 16-row tile on the matrix cores instead of reading the [M, D] tensor a K = 64 GEMM launch wrote
    (reference mona.py:118-127: u = norm(x)·gamma + x·gammax; t = project1(u)).  Same products in the same order, same bf16 rounding of du: dx (fp32 and
    its T copy, row-major or K-blocked) must be bit-identical to the two-launch form, the four parameter gradients equal up to the order of their fp32 sums."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(M)
    D, dt = 768, torch.bfloat16
    x = torch.randn(M, D, generator=g).to(dev()) * 1.5 + 0.3
    dy = torch.randn(M, D, generator=g).to(dev())
    dtt = torch.randn(M, 64, generator=g).to(dev()).to(dt)
    w1 = (torch.randn(64, D, generator=g) * 0.05).to(dev())                  # project1.weight [64, D]
    w1t = w1.t().contiguous().to(dt)                                          # [D, 64]
    nw, nb = (1 + 0.1 * torch.randn(D, generator=g)).to(dev()), (0.1 * torch.randn(D, generator=g)).to(dev())
    gam, gamx = (0.5 * torch.randn(D, generator=g)).to(dev()), (1 + 0.1 * torch.randn(D, generator=g)).to(dev())
    kb = kb and M > 2048

    def run(fused):
        dx = torch.full((M, D), float("nan"), device=dev())
        dx_t = ops.kb_empty(M, D, dt, dev()) if kb else torch.full((M, D), float("nan"), device=dev(), dtype=dt)
        G = [torch.zeros(D, device=dev()) for _ in range(4)]
        if fused:
            ops.mona_pre_bwd(None, x, dy, nw, nb, gam, gamx, dx, dx_t, *G, dt_w1t=(dtt, w1t))
        else:
            du = torch.empty(M, D, device=dev(), dtype=dt)
            ops.gemm(dtt, w1t, out_t=du)
            ops.mona_pre_bwd(du, x, dy, nw, nb, gam, gamx, dx, dx_t, *G)
        torch.cuda.synchronize()
        return dx, (dx_t.t.clone() if kb else dx_t), G

    dx0, t0, G0 = run(False)
    dx1, t1, G1 = run(True)
    assert not torch.isnan(dx1).any() and torch.equal(dx0, dx1) and torch.equal(t0, t1)
    for a, b in zip(G0, G1):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max() + 1e-12)


@pytest.mark.parametrize("M,kb", [(50432, True), (4100, False), (37, False)])
def test_mona_pre_bwd_on_three_byte_residual_gradients(M, kb):
    """uia_mona_pre_bwd_du3: dy arrives as (bf16 hi plane, low bytes), dx leaves as (T copy — row-major or K-blocked —, low bytes).  Against uia_mona_pre_bwd_du on the
    decoded dy: the reconstructed dx within 2^-14 of the fp32 result, the T copy the bf16 rounding of it, the parameter gradients up to their sum order."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(M + 1)
    D, dt = 768, torch.bfloat16
    x = torch.randn(M, D, generator=g).to(dev()) * 1.5 + 0.3
    dy = torch.randn(M, D, generator=g).to(dev())
    dyh, dyl = ops.float_to_three_byte(dy)
    dy3 = ops.three_byte_to_float(dyh, dyl).contiguous()
    dtt = torch.randn(M, 64, generator=g).to(dev()).to(dt)
    w1t = (torch.randn(64, D, generator=g) * 0.05).to(dev()).t().contiguous().to(dt)
    nw, nb = (1 + 0.1 * torch.randn(D, generator=g)).to(dev()), (0.1 * torch.randn(D, generator=g)).to(dev())
    gam, gamx = (0.5 * torch.randn(D, generator=g)).to(dev()), (1 + 0.1 * torch.randn(D, generator=g)).to(dev())
    kb = kb and M > 2048
    new_t = lambda: ops.kb_empty(M, D, dt, dev()) if kb else torch.full((M, D), float("nan"), device=dev(), dtype=dt)
    dx0, t0, G0 = torch.empty(M, D, device=dev()), new_t(), [torch.zeros(D, device=dev()) for _ in range(4)]
    ops.mona_pre_bwd(None, x, dy3, nw, nb, gam, gamx, dx0, t0, *G0, dt_w1t=(dtt, w1t))
    t1, lo1, G1 = new_t(), torch.full((M, D), 99, device=dev(), dtype=torch.int8), [torch.zeros(D, device=dev()) for _ in range(4)]
    ops.mona_pre_bwd(None, x, (dyh, dyl), nw, nb, gam, gamx, None, t1, *G1, dt_w1t=(dtt, w1t), dx_lo=lo1)
    torch.cuda.synchronize()
    got = ops.three_byte_to_float(t1, lo1)
    assert float(((got - dx0).abs() / dx0.abs().clamp_min(1e-3)).max()) < 2.0 ** -14
    # the hi plane IS the T copy: the bf16 rounding of the value it stands for, and — up to the last fp32 bit of two differently contracted instantiations — the fp32 form's copy
    rows = lambda t: t.t.permute(1, 0, 2).reshape(t.rows, t.cols)[:M] if kb else t
    assert torch.equal(rows(t1), got.bfloat16())
    assert float((rows(t1) != rows(t0)).float().mean()) < 1e-3
    for a, b in zip(G0, G1):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max() + 1e-12)
    with pytest.raises(Exception):                                            # three-byte in without three-byte out is not a form the kernel has
        ops.mona_pre_bwd(None, x, (dyh, dyl), nw, nb, gam, gamx, dx0, t1, *G1, dt_w1t=(dtt, w1t))


def test_layernorm_bwd_takes_a_k_blocked_residual_gradient_plane():
    """uia_layernorm_bwd3 with dres_kb_rows: the hi plane of the residual gradient is the K-blocked T copy the Mona backward left for the fc2 data-gradient GEMM."""
    from uia_hip import ops
    torch.manual_seed(21)
    M, D = 4100, 768
    x = torch.randn(M, D, device=dev()) * 2 + 0.3
    dy = torch.randn(M, D, device=dev()).bfloat16()
    gamma = torch.randn(D, device=dev())
    rh, rl = ops.float_to_three_byte(torch.randn(M, D, device=dev()))
    out = []
    for hi in (rh, _kb(rh)):
        dxt, dlo = torch.empty(M, D, device=dev(), dtype=torch.bfloat16), torch.empty(M, D, device=dev(), dtype=torch.int8)
        ops.layernorm_bwd(dy, x, gamma, 1e-6, dres=(hi, rl), dx_t=dxt, dx_lo=dlo)
        out.append((dxt, dlo))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_three_byte_residual_gradients_between_the_backward_functions():
    """engine.GRAD_RESID3 / UF.set_grad_resid3: between MonaFn.backward and VitBlockFn.backward the residual gradient travels as (T copy, low bytes) under a NaN token
    instead of fp32 + T copy.  ViT-B/16 + Mona at 12 images (2364 rows: the ring kernels and their K-blocked copies): the adapter gradients against the fp32 hand-off
    (16 stored mantissa bits against 24 on a stream every GEMM reads in bf16 anyway), tokens actually used, nothing non-finite."""
    import contextlib, io
    from uia_hip import functional as UF
    from src.adapters import inject_mona_variant_to_open_clip
    from src.losses import InfoNCELoss
    from src.third_party.biomedclip.model import create_biomedclip
    model = create_biomedclip(seed=3)
    for p in model.parameters():
        p.requires_grad_(False)
    with contextlib.redirect_stdout(io.StringIO()):
        inject_mona_variant_to_open_clip(model, variant="freq_enhanced", bottleneck_dim=64)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "mona" in k:
                p.copy_((1.0 if k.endswith(("norm.weight", "gammax", "freq_filter")) else 0.0) + 0.05 * torch.randn(p.shape, generator=g))
    for k, p in model.named_parameters():
        p.requires_grad_("mona" in k)
    model = model.to(dev()).train()
    images = torch.rand(12, 3, 224, 224, generator=g).to(dev())
    ids = torch.randint(4, 3000, (12, 256), generator=g)
    ids[:, 40:] = 0
    ids = ids.to(dev())
    made = []
    publish = UF.publish_grad3

    def counting(*a, **k):
        made.append(1)
        return publish(*a, **k)

    def run(flag):
        UF.set_compute_dtype(torch.bfloat16)
        UF.set_grad_resid3(flag)
        UF.set_dropout_seed(7)
        for p in model.parameters():
            p.grad = None
        fi, ft = model.encode_image(images), model.encode_text(ids)
        InfoNCELoss(0.07)(fi, ft).backward()
        UF.clear_t_copies()
        return torch.cat([p.grad.flatten() for p in model.parameters() if p.requires_grad]).clone()

    UF.publish_grad3 = counting
    try:
        g0 = run(False)
        n0 = len(made)
        g1 = run(True)
    finally:
        UF.publish_grad3 = publish
        UF.set_grad_resid3(False)
    assert n0 == 0 and len(made) >= 20                                     # 11 adapters + 11 blocks hand a token on (the last adapter gets fp32 from the head, the first block returns nothing)
    assert bool(torch.isfinite(g1).all())
    assert float(torch.nn.functional.cosine_similarity(g1, g0, dim=0)) > 0.9995
    assert float((g1 - g0).norm() / g0.norm()) < 3e-2


@pytest.mark.parametrize("M", [50432, 1000, 37])
def test_mona_pre_fwd_with_project1_inside_equals_the_two_launches(M):
    """uia_mona_pre_fwd_t: u = norm(x)·gamma + x·gammax AND t = project1(u) (reference mona.py:118-127) in one launch — the u tile goes through LDS to the
    matrix cores in the k order of the N = 64 stream kernel: u and t bit-identical to uia_mona_pre_fwd + uia_gemm."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(M + 1)
    D, dt = 768, torch.bfloat16
    x = torch.randn(M, D, generator=g).to(dev()) * 1.5 + 0.3
    nw, nb = (1 + 0.1 * torch.randn(D, generator=g)).to(dev()), (0.1 * torch.randn(D, generator=g)).to(dev())
    gam, gamx = (0.5 * torch.randn(D, generator=g)).to(dev()), (1 + 0.1 * torch.randn(D, generator=g)).to(dev())
    w1 = (torch.randn(64, D, generator=g) * 0.05).to(dev()).to(dt)
    b1 = torch.randn(64, generator=g).to(dev())
    u0, t0 = torch.empty(M, D, device=dev(), dtype=dt), torch.empty(M, 64, device=dev(), dtype=dt)
    ops.mona_pre_fwd(x, nw, nb, gam, gamx, u0)
    ops.gemm(u0, w1, bias=b1, out_t=t0)
    u1, t1 = torch.full((M, D), float("nan"), device=dev(), dtype=dt), torch.full((M, 64), float("nan"), device=dev(), dtype=dt)
    ops.mona_pre_fwd(x, nw, nb, gam, gamx, u1, proj1=(w1, b1, t1))
    torch.cuda.synchronize()
    assert torch.equal(u0, u1) and not torch.isnan(t1.float()).any()
    if M > 2048:
        assert torch.equal(t0, t1)                                           # the stream kernel (tile cfg 16): the same MFMA chain
    else:
        assert rel(t1, t0) < 1e-2                                            # small M runs project1 on another tile config: equal up to the fp32 order
    assert rel(t1, u0.float() @ w1.float().T + b1) < 1e-2


@pytest.mark.parametrize("M,D,n,p", [(32896, 1024, 3, 0.1), (1000, 768, 3, 0.25), (37, 1024, 1, 0.0), (530, 768, 2, 0.5)])
def test_layernorm_with_the_lora_down_projections_inside_equals_the_separate_launches(M, D, n, p):
    """uia_ln_lora_down: h = LayerNorm(x) and t_s = dropout_s(h)·A_sᵀ (reference lora.py:82-87 on q, k, v of a block) in one launch.  h must be bit-identical
    to uia_layernorm_fwd; t_s equal to the N = 64 stream launch with the same seed up to the association of four K partials; columns 16..63 zero."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(M + D + n)
    dt = torch.bfloat16
    x = torch.randn(M, D, generator=g).to(dev()) * 2 + 0.5
    gw, gb = (1 + 0.1 * torch.randn(D, generator=g)).to(dev()), (0.1 * torch.randn(D, generator=g)).to(dev())
    As = []
    for _ in range(n):
        a = torch.zeros(64, D)
        a[:16] = torch.randn(16, D, generator=g) * 0.05
        As.append(a.to(dev()).to(dt))
    seeds = [1000003 + 17 * i for i in range(n)]
    h0 = torch.empty(M, D, device=dev(), dtype=dt)
    ops.layernorm_fwd(x, gw, gb, 1e-5, y_t=h0)
    h1 = torch.full((M, D), float("nan"), device=dev(), dtype=dt)
    t_all = torch.full((n, M, 64), float("nan"), device=dev(), dtype=dt)
    ops.ln_lora_down(x, gw, gb, 1e-5, h1, As, t_all, p, seeds)
    torch.cuda.synchronize()
    assert torch.equal(h0, h1) and not torch.isnan(t_all.float()).any()
    assert float(t_all[:, :, 16:].abs().max()) == 0.0
    for i in range(n):
        if p > 0:
            hd = torch.empty_like(h0)
            ops.dropout(h0, hd, p, seeds[i])
        else:
            hd = h0
        want = hd.float() @ As[i][:16].float().T
        assert rel(t_all[i, :, :16], want) < 6e-3, i
        if M > 2048 and p > 0:                                              # the launch it replaces, same seed
            t_ref = torch.empty(M, 64, device=dev(), dtype=dt)
            ops.gemm(h0, As[i], out_t=t_ref, drop=("a", p, seeds[i], None))
            assert rel(t_all[i], t_ref) < 6e-3


def test_text_tower_boundary_masks_equal_the_run_time_epilogue_bit_for_bit():
    """The first and the last layer of the three-byte text tower use two more compile-time epilogue masks (fp32 LayerNorm rows in / three-byte sum + row sums
    out; three-byte sum in / fp32 rows out).  Same arithmetic as the run-time epilogue (tile cfg 10) on a ragged shape: every output bit-identical."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(77)
    M, N, K, dt = 4353, 768, 128, torch.bfloat16
    a = torch.randn(M, K, generator=g).to(dev()).to(dt)
    w = ops.PackedW((torch.randn(N, K, generator=g) * K ** -0.5).to(dev()).to(dt))
    bias = torch.randn(N, generator=g).to(dev())
    prev = torch.randn(M, N, generator=g).to(dev()) * 2 + 0.3
    stats = torch.stack([prev.mean(1), (prev.var(1, unbiased=False) + 1e-12).rsqrt()], 1).contiguous()
    lw, lb = torch.randn(N, generator=g).to(dev()), torch.randn(N, generator=g).to(dev())
    hi, lo = ops.float_to_three_byte(prev)

    def first(cfg):                                                          # BIAS | RESID | RESID_LN | OUTT | OUT_LO | ROWSUM
        ot, ol = torch.empty(M, N, device=dev(), dtype=dt), torch.full((M, N), 99, device=dev(), dtype=torch.int8)
        rs = torch.zeros(M, 2, device=dev(), dtype=torch.int64)
        ops.gemm(a, w, bias=bias, resid=prev, resid_ln=(stats, lw, lb), out_t=ot, out_lo=ol, rowsum=rs, tile_cfg=cfg)
        return ot, ol, rs

    def last(cfg):                                                           # BIAS | RESID_LO | RESID_LN | OUT32
        o32 = torch.full((M, N), float("nan"), device=dev())
        ops.gemm(a, w, bias=bias, resid3=(hi, lo), resid_ln=(stats, lw, lb), out32=o32, tile_cfg=cfg)
        return (o32,)

    for fn in (first, last):
        ref, got = fn(10), fn(8)
        torch.cuda.synchronize()
        for x, y in zip(ref, got):
            assert torch.equal(x, y), fn.__name__
    want = a.float() @ w.row.float().T + bias + ((prev - stats[:, :1]) * stats[:, 1:]) * lw + lb
    assert rel(ops.three_byte_to_float(*first(8)[:2]), want) < 3e-5


# ------------------------------------------------------------------------------------------------ weight caches across streams
def test_cold_weight_cache_filled_on_one_stream_is_safe_to_read_from_another():
    """the operand forms of a weight (cast, transpose, K-blocked twin) are written by kernels enqueued on whichever stream misses first; a
    second stream that hits the Python cache must not read them before they exist.  Real-size layer, COLD cache, and the filling stream kept busy so that
    the fill is still queued when the other stream launches its GEMM: with the event of uia_hip.ops.Ready the result is right; without it the GEMM would
    read an unwritten buffer."""
    from uia_hip import functional as UF
    from uia_hip import ops
    dt = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(3)
    w = torch.nn.Parameter((torch.randn(3072, 768, generator=g) * 768 ** -0.5).to(dev()), requires_grad=False)
    a = torch.randn(4352, 768, generator=g).to(dev()).to(dt)
    want = (a.float() @ w.detach().to(dt).float().T)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    busy = torch.randn(8192, 8192, device=dev(), dtype=dt)
    torch.cuda.synchronize()
    for transpose in (False, True):
        a_t = a if not transpose else torch.randn(4352, 3072, generator=g).to(dev()).to(dt)
        ref = want if not transpose else a_t.float() @ w.detach().to(dt).float()
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            for _ in range(6):
                busy = (busy @ busy).clamp_(-1, 1)                       # ~10 ms of work in front of the fill
            op = UF.WEIGHTS.get(w, dt, transpose=transpose)               # cold: cast / transpose / pack kernels queued on sa behind it
        out = torch.full((a_t.shape[0], ref.shape[1]), float("nan"), device=dev(), dtype=dt)
        with torch.cuda.stream(sb):
            hit = UF.WEIGHTS.get(w, dt, transpose=transpose)              # Python-side cache hit from another stream
            assert hit is op
            ops.gemm(a_t, hit, out_t=out)
        torch.cuda.synchronize()
        assert not torch.isnan(out.float()).any() and rel(out, ref) < 1e-2


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_contrastive_step_with_the_image_tower_in_two_slices_equals_one_slice(mode):
    """engine.contrastive_step(image_split=k) (engine.IMAGE_SPLIT, the default above 32 images): the image tower runs as two slices on two HIP streams beside the text
    tower's stream, forward and backward; the loss is still one InfoNCE over all pairs — same loss, same gradients (float atomics reorder the weight-gradient sums)."""
    from uia_hip import functional as UF
    from uia_hip.engine import FlatAdapterOptimizer, contrastive_step
    from src.losses import InfoNCELoss
    from tests.test_round2_gpu import _toy_batch, _toy_model
    UF.set_compute_dtype(torch.float32 if mode == "fp32" else torch.bfloat16)
    images, ids = _toy_batch(23, B=8)
    outs = []
    for split, mbs in ((0, 1), (3, 1), (0, 2), (1, 2)):                  # the last two: two micro-batches of four pairs, whole and cut 1 + 3
        model = _toy_model("hybrid", seed=13).to(dev())
        opt = FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-3)
        UF.set_dropout_seed(7)
        opt.snapshot_grads = True                                        # the guarded update zeroes the accumulator: keep a copy of what it consumed
        loss = contrastive_step(model, InfoNCELoss(0.07), opt, images.to(dev()), ids.to(dev()), overlap_text=True, image_split=split, micro_batches=mbs)
        torch.cuda.synchronize()
        outs.append((float(loss), opt.last_g, opt.grad_norm()))
    UF.set_compute_dtype(torch.bfloat16)
    tol = 1e-5 if mode == "fp32" else 2e-2
    for a, b in ((0, 1), (2, 3)):
        assert abs(outs[a][0] - outs[b][0]) < tol * max(1.0, abs(outs[a][0]))
        g1, g2 = outs[a][1], outs[b][1]
        assert float((g1 - g2).norm() / g1.norm()) < (1e-4 if mode == "fp32" else 5e-2), float((g1 - g2).norm() / g1.norm())
        assert abs(outs[a][2] - outs[b][2]) < (1e-4 if mode == "fp32" else 5e-2) * outs[a][2]


def test_contrastive_step_leaves_the_three_byte_gradient_mode_off_and_skips_hooked_models():
    """engine.contrastive_step turns functional.set_grad_resid3 on for its own forward + backward only — a later loop in the same process (tapped blocks feeding a
    segmentation head) must not inherit tokens — and not at all for a model that carries module hooks (something else may read the tensors between the Functions)."""
    from uia_hip import functional as UF
    from uia_hip import engine
    from src.losses import InfoNCELoss
    from tests.test_round2_gpu import _toy_batch, _toy_model
    UF.set_compute_dtype(torch.bfloat16)
    images, ids = _toy_batch(29, B=8)
    seen = []
    orig = UF.set_grad_resid3

    def spy(flag):
        seen.append(bool(flag))
        orig(flag)

    UF.set_grad_resid3 = spy
    try:
        for hooked in (False, True):
            model = _toy_model("freq_enhanced", seed=3).to(dev())
            if hooked:
                model.visual.trunk.blocks[1].register_forward_hook(lambda m, i, o: None)
            opt = engine.FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-3)
            seen.clear()
            opt.snapshot_grads = True
            engine.contrastive_step(model, InfoNCELoss(0.07), opt, images.to(dev()), ids.to(dev()))
            torch.cuda.synchronize()
            assert seen[0] == (engine.GRAD_RESID3 and not hooked) and seen[-1] is False
            assert not UF.grad_resid3_enabled()
            assert bool(torch.isfinite(opt.last_g).all()) and float(opt.last_g.abs().sum()) > 0
    finally:
        UF.set_grad_resid3 = orig


def test_text_tower_ahead_of_the_previous_step_gives_the_same_steps():
    """contrastive_step(inputs_ready=True): with a frozen text tower and resident inputs its stream does not wait for the caller's, so step t+1's text tower may run beside step
    t's backward and optimiser.  Three consecutive steps (the second and third read weights the first two updated — in the IMAGE tower only) give the same losses and the same
    parameters as the serialised order."""
    from uia_hip import functional as UF
    from uia_hip.engine import FlatAdapterOptimizer, contrastive_step
    from src.losses import InfoNCELoss
    from tests.test_round2_gpu import _toy_batch, _toy_model
    UF.set_compute_dtype(torch.float32)
    images, ids = _toy_batch(31, B=8)
    images, ids = images.to(dev()), ids.to(dev())
    torch.cuda.synchronize()
    outs = []
    try:
        for ahead in (False, True):
            model = _toy_model("freq_enhanced", seed=17).to(dev())
            opt = FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-2)
            UF.set_dropout_seed(3)
            losses = [float(contrastive_step(model, InfoNCELoss(0.07), opt, images, ids, overlap_text=True, image_split=4, inputs_ready=ahead)) for _ in range(3)]
            torch.cuda.synchronize()
            outs.append((losses, torch.cat([p.detach().flatten() for p in model.parameters() if p.requires_grad]).clone()))
    finally:
        UF.set_compute_dtype(torch.bfloat16)
    assert max(abs(a - b) for a, b in zip(*[o[0] for o in outs])) < 1e-5 and outs[0][0][2] != outs[0][0][0]
    d = (outs[0][1] - outs[1][1]).abs()                                  # (AdamW turns the last-bit noise of the float-atomic gradient sums into lr-sized steps where a gradient is ~0)
    assert float((d > 1e-3).float().mean()) < 1e-3 and float(d.mean()) < 1e-5


@pytest.mark.parametrize("p_drop", [0.2, 0.0])
def test_grouped_lora_weight_gradients_equal_the_separate_launches(p_drop):
    """uia_wgrad_group (ops.LORA_WGRAD_GROUP): the dB launches of q, k, v as one launch and their dA launches — each regenerating its own dropout mask — as another
    (reference lora.py:82-87 three times).  Same products per problem as uia_wgrad_ex / uia_wgrad_drop: the flat gradient buffer equals the six-launch form up to the
    order of the float atomics."""
    from uia_hip import functional as UF
    from uia_hip import ops
    from uia_hip.engine import FlatAdapterOptimizer
    from src.third_party.openai_clip.model import ResidualAttentionBlock
    from src.adapters.lora import PlainMultiheadAttentionLoRA
    UF.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(6)
    D, H, B, L = 256, 4, 40, 65
    blk = ResidualAttentionBlock(D, H).to(dev())
    for q in blk.parameters():
        q.requires_grad_(False)
    blk.attn = PlainMultiheadAttentionLoRA(blk.attn, r=16, lora_alpha=32, dropout_rate=p_drop).to(dev())
    for k, q in blk.named_parameters():
        q.requires_grad_("lora" in k.lower())
        if "lora_B" in k:
            torch.nn.init.normal_(q, std=0.05)
    blk.train()
    opt = FlatAdapterOptimizer([(k, q) for k, q in blk.named_parameters() if q.requires_grad], lr=1e-3)
    x = torch.randn(L, B, D, device=dev(), requires_grad=True)
    calls, outs = [], []
    real = ops.wgrad_group

    def counting(*a, **k):
        calls.append(1)
        return real(*a, **k)

    ops.wgrad_group = counting
    try:
        for grouped in (True, False):
            ops.LORA_WGRAD_GROUP = grouped
            UF.set_dropout_seed(99)
            opt.zero_grad()
            x.grad = None
            n0 = len(calls)
            (blk(x).float() ** 2).mean().backward()
            torch.cuda.synchronize()
            outs.append((opt.g.clone(), x.grad.clone(), len(calls) - n0))
    finally:
        ops.wgrad_group = real
        ops.LORA_WGRAD_GROUP = True
    (g0, gx0, n_g), (g1, gx1, n_s) = outs
    assert n_g == 2 and n_s == 0
    assert torch.equal(gx0, gx1)
    assert float(g1.abs().max()) > 0 and float((g0 - g1).abs().max()) <= 1e-5 * float(g1.abs().max())
