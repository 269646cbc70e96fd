"""-m gpu: LayerNorm folded into the GEMMs on either side of it (uia_gemm_desc.rowsum_out / lnfold_* / resid_ln_dim; bf16 training step).

Kernel level: the row sums a producing epilogue leaves, the normalised accumulators of a consuming GEMM and the deferred residual that
reads sums instead of (mean, rstd), each against torch on the same operands, on the ring tile configs (compile-time masks and the run-time
epilogue), the half-height tail config and the small-M config.  Model level: the folded ViT block / BERT layer against the same model
with the stand-alone LayerNorm kernels (bf16) and against the oracle (tests/test_parity_gpu.py and test_fullshape_gpu.py run with the
fold on, the default)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture(autouse=True)
def _mode():
    from uia_hip import functional as UF
    yield
    UF.set_compute_dtype(torch.bfloat16)
    UF.set_ln_fold(True)
    UF._STATE.pop("ln_fold_min_rows", None)
    UF.clear_t_copies()


def _ln(x, w, b, eps):
    return torch.nn.functional.layer_norm(x, (x.shape[1],), w, b, eps)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("M,cfg", [(4100, 0), (4100, 10), (2500, 13), (300, 0), (2500, 14)])
def test_gemm_rowsum_of_stored_rows(dt, M, cfg):
    """producer: out32 = a·wᵀ + bias + resid, T copy of the same rows, (Σ, Σ²) per row; M is no multiple of any tile height"""
    from uia_hip import ops
    g = torch.Generator().manual_seed(M + cfg)
    N, K = 768, 256
    a = torch.randn(M, K, generator=g).to(dev()).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev()).to(dt)
    bias = torch.randn(N, generator=g).to(dev())
    resid = (torch.randn(M, N, generator=g) * 2 + 0.5).to(dev())
    out32 = torch.empty(M, N, device=dev())
    out_t = torch.empty(M, N, device=dev(), dtype=dt)
    sums_q = torch.zeros(M, 2, device=dev(), dtype=torch.int64)
    ops.gemm(a, w, bias=bias, resid=resid, out32=out32, out_t=out_t, rowsum=sums_q, tile_cfg=cfg)
    sums = ops.rowsum_to_float(sums_q)
    again = torch.zeros_like(sums_q)                                        # integer atomics: the same bits on every launch
    ops.gemm(a, w, bias=bias, resid=resid, out32=torch.empty_like(out32), out_t=torch.empty_like(out_t), rowsum=again, tile_cfg=cfg)
    assert torch.equal(again, sums_q)
    ref = a.float() @ w.float().T + bias + resid
    assert rel(out32, ref) < 2e-5
    assert rel(out_t, ref.to(dt)) < (1e-2 if dt == torch.bfloat16 else 2e-5)
    s1, s2 = out32.double().sum(1), (out32.double() ** 2).sum(1)            # sums of the rows the launch itself stored
    assert float((sums[:, 0].double() - s1).abs().max() / s1.abs().max()) < 2e-6
    assert float((sums[:, 1].double() - s2).abs().max() / s2.abs().max()) < 2e-6


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("M,cfg,act,aux", [(4100, 0, None, False), (4100, 0, "gelu", False), (4100, 0, "gelu", True), (4100, 10, "gelu", True),
                                           (4100, 12, "gelu", True), (70000, 12, None, False), (2500, 13, None, False), (300, 0, "gelu", True)])
def test_gemm_layernorm_folded_into_consumer(dt, M, cfg, act, aux):
    """consumer: A = raw rows, W' = W·ln_w, epilogue rstd·(acc − mean·colsum) + (b + W·ln_b): against LayerNorm → Linear in torch"""
    from uia_hip import ops
    g = torch.Generator().manual_seed(7 * M + cfg)
    D, N, eps = 256, 768, 1e-6
    x = (torch.randn(M, D, generator=g) * 1.7 + 0.3).to(dev())               # a non-zero row mean: the −mean·colsum term matters
    lw = (1 + 0.3 * torch.randn(D, generator=g)).to(dev())
    lb = (0.2 * torch.randn(D, generator=g)).to(dev())
    W = (torch.randn(N, D, generator=g) * 0.05).to(dev())
    b = torch.randn(N, generator=g).to(dev())
    x_t = x.to(dt)
    wf = (W * lw[None, :]).to(dt).contiguous()
    colsum = wf.float().sum(1).contiguous()
    bias = (b + W @ lb).contiguous()
    sums = ops.rowsum_from_float(torch.stack([x.sum(1), (x * x).sum(1)], 1))
    out = torch.empty(M, N, device=dev(), dtype=dt)
    pre = torch.empty(M, N, device=dev(), dtype=dt) if aux else None
    ops.gemm(x_t, wf, bias=bias, act=act, aux_out=pre, out_t=out, lnfold=(sums, colsum, D, eps), tile_cfg=cfg)
    # same operands, fp64 arithmetic
    mean = x.double().mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(x.double().var(1, unbiased=False, keepdim=True) + eps)
    z = rstd * (x_t.double() @ wf.double().T - mean * colsum.double()[None, :]) + bias.double()
    y = torch.nn.functional.gelu(z) if act else z
    tol = 1.2e-2 if dt == torch.bfloat16 else 3e-5                            # bf16: the output rounding (+ the polynomial GELU of the bf16 epilogue)
    assert rel(out, y) < tol
    if aux:
        assert rel(pre, z) < tol
    # and it IS the LayerNorm followed by the Linear, up to the rounding of the operands
    true = _ln(x, lw, lb, eps) @ W.T + b
    true = torch.nn.functional.gelu(true) if act else true
    assert rel(out, true) < (3e-2 if dt == torch.bfloat16 else 1e-4)


@pytest.mark.parametrize("shift", [0.0, 1.0, 4.0, 16.0])
def test_fold_error_grows_with_row_mean_over_std(shift):
    """What the fold costs when rows are NOT centred: the operand is bf16(x) instead of bf16(LN(x)), so its rounding error is relative to
    |x| ≈ |mean| rather than to the spread, and the LayerNorm output inherits it amplified by |mean| / std.  The test pins that law
    (error ≤ 2^-8·sqrt(1 + (mean/std)²) of the output scale, a few times the plain bf16 path at ratio 0-1) — transformer residual
    rows sit at |mean| / std well below 1; a model whose rows do not should run with set_ln_fold(False)."""
    from uia_hip import ops
    g = torch.Generator().manual_seed(int(shift * 10) + 1)
    M, D, N, eps = 2304, 256, 768, 1e-6
    x = (torch.randn(M, D, generator=g) + shift).to(dev())                  # std 1, mean = shift
    lw = (1 + 0.3 * torch.randn(D, generator=g)).to(dev())
    lb = (0.2 * torch.randn(D, generator=g)).to(dev())
    W = (torch.randn(N, D, generator=g) * 0.05).to(dev())
    b = torch.randn(N, generator=g).to(dev())
    wf = (W * lw[None, :]).bfloat16().contiguous()
    sums = ops.rowsum_from_float(torch.stack([x.sum(1), (x * x).sum(1)], 1))
    out = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    ops.gemm(x.bfloat16(), wf, bias=(b + W @ lb).contiguous(), out_t=out, lnfold=(sums, wf.float().sum(1).contiguous(), D, eps))
    true = _ln(x.double(), lw.double(), lb.double(), eps) @ W.double().T + b.double()
    plain = (_ln(x, lw, lb, eps).bfloat16().float() @ W.bfloat16().float().T + b).bfloat16()
    e_fold, e_plain = rel(out, true), rel(plain, true)
    assert e_fold < 2.0 ** -8 * (1 + shift * shift) ** 0.5 + 4e-3, (shift, e_fold, e_plain)
    if shift <= 1.0:
        assert e_fold < 2.5 * e_plain + 2e-3, (shift, e_fold, e_plain)


@pytest.mark.parametrize("M,cfg", [(4100, 0), (4100, 10), (300, 0)])
def test_gemm_deferred_residual_from_row_sums(M, cfg):
    """post-LN sub-layer sum whose residual is LayerNorm(resid) with the statistics given as (Σ, Σ²): against torch, and against the
    (mean, rstd) form of the same descriptor"""
    from uia_hip import ops
    g = torch.Generator().manual_seed(M)
    N, K, eps = 768, 256, 1e-12
    a = torch.randn(M, K, generator=g).to(dev()).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev()).bfloat16()
    bias = torch.randn(N, generator=g).to(dev())
    raw = (torch.randn(M, N, generator=g) * 1.3 - 0.2).to(dev())
    lw = (1 + 0.3 * torch.randn(N, generator=g)).to(dev())
    lb = (0.2 * torch.randn(N, generator=g)).to(dev())
    sums = ops.rowsum_from_float(torch.stack([raw.sum(1), (raw * raw).sum(1)], 1))
    out = torch.empty(M, N, device=dev())
    out_t = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    s2q = torch.zeros(M, 2, device=dev(), dtype=torch.int64)
    ops.gemm(a, w, bias=bias, resid=raw, resid_ln=(sums, lw, lb, N, eps), out32=out, out_t=out_t, rowsum=s2q, tile_cfg=cfg)
    s2 = ops.rowsum_to_float(s2q)
    ref = a.float() @ w.float().T + bias + _ln(raw, lw, lb, eps)
    assert rel(out, ref) < 2e-5
    assert float((s2[:, 0] - out.sum(1)).abs().max() / out.sum(1).abs().max()) < 1e-5
    stats = torch.stack([raw.mean(1), 1.0 / torch.sqrt(raw.var(1, unbiased=False) + eps)], 1).contiguous()
    out2 = torch.empty_like(out)
    ops.gemm(a, w, bias=bias, resid=raw, resid_ln=(stats, lw, lb), out32=out2, tile_cfg=cfg)
    assert rel(out, out2) < 1e-5


def test_gemm_lnfold_argument_checks():
    from uia_hip import ops
    from uia_hip._lib import UiaError
    a = torch.zeros(256, 128, device=dev(), dtype=torch.bfloat16)
    w = torch.zeros(64, 128, device=dev(), dtype=torch.bfloat16)
    out = torch.empty(256, 64, device=dev(), dtype=torch.bfloat16)
    with pytest.raises(UiaError):
        ops.gemm(a, w, out_t=out, rowsum=torch.zeros(100, 2, device=dev(), dtype=torch.int64))   # too few rows
    with pytest.raises(UiaError):
        ops.gemm(a, w, out_t=out, rowsum=torch.zeros(256, 2, device=dev()))                      # float sums: the kernels exchange int64 fixed point
    q = torch.zeros(256, 2, device=dev(), dtype=torch.int64)
    with pytest.raises(UiaError):
        ops.gemm(a, w, out_t=out, lnfold=(q, torch.zeros(32, device=dev()), 128, 1e-6))          # colsum too short
    with pytest.raises(UiaError):
        ops.gemm(a, w, out_t=out, alpha=2.0, lnfold=(q, torch.zeros(64, device=dev()), 128, 1e-6))


def _to_kb(t):
    """row-major [M, K] -> ops.KBlocked ([K/g, M, g], g = 64 bytes of elements), converted with torch ops"""
    from uia_hip import ops
    g = 64 // t.element_size()
    M, K = t.shape
    return ops.KBlocked(t.view(M, K // g, g).permute(1, 0, 2).contiguous())


def _from_kb(kb):
    Kb, M, g = kb.t.shape
    return kb.t.permute(1, 0, 2).reshape(M, Kb * g)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("M,cfg", [(4100, 0), (4100, 10), (4100, 12), (2500, 13), (50432, 0)])
def test_gemm_kblocked_activations(dt, M, cfg):
    """A read K-blocked and the T result written K-blocked (what a GEMM -> GEMM activation does between two ring launches), alone and
    chained, against the row-major launches of the same operands: identical arithmetic, so the results must be bit-identical."""
    from uia_hip import ops
    if dt == torch.float32 and M > 5000:
        pytest.skip("one large case is enough")
    g = torch.Generator().manual_seed(M + cfg)
    K, N = 256, 768
    a = torch.randn(M, K, generator=g).to(dev()).to(dt)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev()).to(dt)
    w2 = (torch.randn(128, N, generator=g) * 0.05).to(dev()).to(dt)
    bias = torch.randn(N, generator=g).to(dev())
    ref = torch.empty(M, N, device=dev(), dtype=dt)
    ops.gemm(a, w, bias=bias, act="gelu", out_t=ref, tile_cfg=cfg)
    # (1) K-blocked A
    o1 = torch.empty(M, N, device=dev(), dtype=dt)
    ops.gemm(_to_kb(a), w, bias=bias, act="gelu", out_t=o1, tile_cfg=cfg)
    assert torch.equal(o1, ref)
    # (2) K-blocked T output (inside a larger K-blocked tensor: the plane stride is not M)
    big = ops.kb_empty(M + 300, N, dt, dev())
    big.t.zero_()
    o2 = big.row_range(100, 100 + M)
    ops.gemm(a, w, bias=bias, act="gelu", out_t=o2, tile_cfg=cfg)
    assert torch.equal(_from_kb(o2), ref)
    assert float(big.t[:, :100].float().abs().max()) == 0 and float(big.t[:, 100 + M:].float().abs().max()) == 0    # nothing outside its rows
    # (3) chained: the K-blocked result is the next launch's A
    r3, o3 = torch.empty(M, 128, device=dev(), dtype=dt), torch.empty(M, 128, device=dev(), dtype=dt)
    ops.gemm(ref, w2, out_t=r3, tile_cfg=cfg)
    ops.gemm(o2, w2, out_t=o3, tile_cfg=cfg)
    assert torch.equal(o3, r3)


@pytest.mark.parametrize("M,D", [(2500, 768), (197, 128)])
def test_mona_pre_bwd_kblocked_t_copy(M, D):
    """the T copy of dx written K-blocked holds exactly the values of the row-major copy of the same launch"""
    from uia_hip import ops
    g = torch.Generator().manual_seed(M)
    mk = lambda *sh: torch.randn(*sh, generator=g).to(dev())
    du, x, dy = mk(M, D).bfloat16(), mk(M, D), mk(M, D)
    nw, nb, ga, gx = 1 + 0.2 * mk(D), 0.1 * mk(D), 0.3 * mk(D), 1 + 0.2 * mk(D)
    outs = []
    for kb in (False, True):
        dx32 = torch.empty(M, D, device=dev())
        dx_t = ops.kb_empty(M, D, torch.bfloat16, dev()) if kb else torch.empty(M, D, device=dev(), dtype=torch.bfloat16)
        grads = [torch.zeros(D, device=dev()) for _ in range(4)]
        ops.mona_pre_bwd(du, x, dy, nw, nb, ga, gx, dx32, dx_t, *grads)
        outs.append((dx32, _from_kb(dx_t) if kb else dx_t, grads))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0])
    assert rel(outs[1][1], outs[1][0]) < 1e-2


@pytest.mark.parametrize("L,mask", [(197, None), (256, "keypad"), (50, "causal")])
def test_attention_kblocked_tensors(L, mask):
    """forward output written K-blocked, backward reading it K-blocked and writing the fused dq/dk/dv K-blocked: the same values as the
    row-major launches, bit for bit (only addresses change)"""
    from uia_hip import ops
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
    lse2 = torch.empty_like(lse)
    ops.attn_fwd(q, k, v, okb, B, H, L, lse=lse2, mask=mask, keylen=keylen)
    assert torch.equal(_from_kb(okb), out) and torch.equal(lse, lse2)
    do = torch.randn(B * L, D, generator=g).to(dev()).bfloat16()
    dqkv = torch.empty_like(qkv)
    ops.attn_bwd(q, k, v, out, do, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, L, mask=mask, keylen=keylen)
    dkb = ops.kb_empty(B * L, 3 * D, torch.bfloat16, dev())
    ops.attn_bwd(q, k, v, okb, do, lse, dkb, None, None, B, H, L, mask=mask, keylen=keylen)
    assert torch.equal(_from_kb(dkb), dqkv)


def test_gemm_kblocked_needs_a_ring_config():
    from uia_hip import ops
    from uia_hip._lib import UiaError
    a = torch.zeros(256, 128, device=dev(), dtype=torch.bfloat16)
    w = torch.zeros(128, 128, device=dev(), dtype=torch.bfloat16)
    out = torch.empty(256, 128, device=dev(), dtype=torch.bfloat16)
    with pytest.raises(UiaError):
        ops.gemm(_to_kb(a), w, out_t=out)                                   # M <= 2048: the small-M config cannot read it
    with pytest.raises(UiaError):
        ops.gemm(a, w, out_t=ops.kb_empty(256, 128, torch.bfloat16, dev()))


TOY = dict(embed_dim=128, vision_cfg=dict(img_size=32, patch_size=8, embed_dim=128, depth=3, num_heads=2),
           text_cfg=dict(vocab_size=120, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256,
                         max_position_embeddings=40))


def _toy(variant="freq_enhanced"):
    from src.adapters import inject_mona_variant_to_open_clip
    from src.third_party.biomedclip.model import create_biomedclip
    import contextlib, io
    model = create_biomedclip(config=TOY, seed=11)
    for p in model.parameters():
        p.requires_grad_(False)
    with contextlib.redirect_stdout(io.StringIO()):
        inject_mona_variant_to_open_clip(model, variant=variant, bottleneck_dim=64)
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "mona" in k:
                p.copy_((1.0 if k.endswith(("norm.weight", "gammax", "freq_filter")) else 0.0) + 0.08 * torch.randn(p.shape, generator=g))
            elif "norm" in k.lower() and k.endswith("weight"):
                p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=g))           # LayerNorm weights / biases away from (1, 0): the fold must carry them
            elif "norm" in k.lower() and k.endswith("bias"):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
    for k, p in model.named_parameters():
        p.requires_grad_("mona" in k)
    return model.eval().to(dev())


def test_folded_towers_match_the_layernorm_kernels():
    """bf16 step with the fold (default) against the same step with the stand-alone LayerNorm kernels: features, loss and adapter
    gradients agree to bf16 rounding; and the folded run stays within the bf16 bound of the fp32-mode features"""
    from uia_hip import functional as UF
    from src.losses import InfoNCELoss
    model = _toy()
    g = torch.Generator().manual_seed(3)
    images = torch.rand(6, 3, 32, 32, generator=g).to(dev())
    ids = torch.randint(4, 120, (6, 12), generator=g)
    ids[:, 0] = 2
    ids[1, 7:] = 0
    ids[4, 3:] = 0
    ids = ids.to(dev())

    UF._STATE["ln_fold_min_rows"] = 0                      # the product applies the fold above 2048 rows only: force it at this toy batch
                                                           # (small-M tile config, plain epilogue with per-segment atomics)

    def run(fold, dt):
        UF.set_compute_dtype(dt)
        UF.set_ln_fold(fold)
        UF.set_dropout_seed(5)
        for p in model.parameters():
            p.grad = None
        fi, ft = model.encode_image(images), model.encode_text(ids)
        loss = InfoNCELoss(0.07)(fi, ft)
        loss.backward()
        UF.clear_t_copies()
        gr = torch.cat([p.grad.flatten() for p in model.parameters() if p.requires_grad])
        return fi.detach().clone(), ft.detach().clone(), float(loss), gr.clone()

    fi32, ft32, l32, g32 = run(False, torch.float32)
    fi0, ft0, l0, g0 = run(False, torch.bfloat16)
    fi1, ft1, l1, g1 = run(True, torch.bfloat16)
    print("folded-vs-fp32", rel(fi1, fi32), rel(ft1, ft32), "plain-vs-fp32", rel(fi0, fi32), rel(ft0, ft32))
    assert rel(fi1, fi32) < 1e-2 and rel(ft1, ft32) < 1e-2, (rel(fi1, fi32), rel(ft1, ft32))
    # the fold is no less accurate than the LayerNorm kernels' bf16 path (same number of bf16 roundings on the way)
    assert rel(fi1, fi32) < 2.0 * rel(fi0, fi32) + 2e-3 and rel(ft1, ft32) < 2.0 * rel(ft0, ft32) + 2e-3
    cos = float(torch.nn.functional.cosine_similarity(g1, g32, dim=0))
    assert cos > 0.99, cos
    assert abs(l1 - l32) < 2e-2 * max(1.0, abs(l32))


def test_large_batch_step_on_the_ring_kernels_fold_and_kblocked_vs_plain():
    """M > 2048 rows in both towers: the step runs on the ring kernels with the LayerNorms folded and the GEMM -> GEMM activations
    K-blocked (the configuration bench.py times) — against the same bf16 step with both switched off, and the fp32-mode features."""
    from uia_hip import functional as UF, ops
    from src.losses import InfoNCELoss
    model = _toy()
    g = torch.Generator().manual_seed(4)
    B = 136                                                             # 136 x 17 tokens = 2312 rows; 136 x 24 positions = 3264 rows
    images = torch.rand(B, 3, 32, 32, generator=g).to(dev())
    ids = torch.randint(4, 120, (B, 24), generator=g)
    ids[:, 0] = 2
    for b in range(B):
        ids[b, 5 + b % 19:] = 0
    ids = ids.to(dev())
    assert ops.kb_ok(B * 17, 384, 128, torch.bfloat16) and ops.kb_ok(B * 24, 256, 128, torch.bfloat16)

    def run(fold, kb, dt):
        UF.set_compute_dtype(dt)
        UF.set_ln_fold(fold)
        ops.KBLOCK_ACT = kb
        UF.set_dropout_seed(5)
        for p in model.parameters():
            p.grad = None
        fi, ft = model.encode_image(images), model.encode_text(ids)
        loss = InfoNCELoss(0.07)(fi, ft)
        loss.backward()
        UF.clear_t_copies()
        gr = torch.cat([p.grad.flatten() for p in model.parameters() if p.requires_grad])
        return fi.detach().clone(), ft.detach().clone(), float(loss.detach()), gr.clone()

    try:
        fi32, ft32, l32, g32 = run(False, False, torch.float32)
        fi0, ft0, l0, g0 = run(False, False, torch.bfloat16)
        fi1, ft1, l1, g1 = run(True, False, torch.bfloat16)
        fi2, ft2, l2, g2 = run(True, True, torch.bfloat16)
    finally:
        ops.KBLOCK_ACT = True
    # the layout changes nothing: same kernels, same operands, same order of additions per output element (row sums: atomics, so only close)
    print("large-batch: kb-vs-fold", rel(fi2, fi1), rel(ft2, ft1), "fold-vs-fp32", rel(fi2, fi32), rel(ft2, ft32), "plain-vs-fp32", rel(fi0, fi32), rel(ft0, ft32),
          "cos", float(torch.nn.functional.cosine_similarity(g2, g32, dim=0)), "loss", l2, l32)
    assert rel(fi2, fi1) < 2e-3 and rel(ft2, ft1) < 2e-3
    assert rel(fi2, fi32) < 1e-2 and rel(ft2, ft32) < 1e-2, (rel(fi2, fi32), rel(ft2, ft32))
    assert rel(fi2, fi32) < 2.0 * rel(fi0, fi32) + 2e-3 and rel(ft2, ft32) < 2.0 * rel(ft0, ft32) + 2e-3
    assert float(torch.nn.functional.cosine_similarity(g2, g32, dim=0)) > 0.99
    assert abs(l2 - l32) < 2e-2 * max(1.0, abs(l32))


def test_fold_guard_switches_the_fold_off_for_uncentred_rows():
    """device-side guard (uia_gemm_desc.ln_flag): EVERY consumer launch of a folded LayerNorm checks its rows; a row with |mean| > 8 std sets
    bit 0, and the host poll turns the fold off with a warning.  Centred rows leave the word clear."""
    from uia_hip import functional as UF
    from uia_hip import ops
    M, D, N = 4096, 768, 768
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(M, D, generator=g).to(dev())
    w = ops.PackedW((torch.randn(N, D, generator=g) * 0.03).to(dev()).to(torch.bfloat16))
    cs = w.row.float().sum(1).contiguous()
    out = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)

    def consume(rows):
        sums = ops.rowsum_from_float(torch.stack([rows.sum(1), (rows * rows).sum(1)], 1))
        ops.gemm(rows.to(torch.bfloat16), w, out_t=out, lnfold=(sums, cs, D, 1e-5), bias=torch.zeros(N, device=dev()))

    try:
        UF.set_ln_fold(True)
        UF.reset_ln_flag()
        consume(x)
        assert UF.poll_ln_flag(sync=True) == 0 and UF.ln_fold_enabled(torch.bfloat16)
        x[3001] += 40.0                                        # one row deep inside the launch, far from the first tiles
        consume(x)
        with pytest.warns(UserWarning, match="LayerNorm fold switched off"):
            assert UF.poll_ln_flag(sync=True) & 1
        assert not UF.ln_fold_enabled(torch.bfloat16)
    finally:
        UF.reset_ln_flag()
        UF.set_ln_fold(True)


def test_rowsum_out_of_range_is_flagged_not_wrapped():
    """A producer whose rows leave the fixed-point range (or are not finite) clamps its partial sums — no integer wrap, no undefined
    llrintf — and sets bit 1 of the guard word: the host raises instead of training on a finite but wrong LayerNorm statistic."""
    from uia_hip import functional as UF
    from uia_hip import ops
    M, K, N = 4096, 64, 768
    a = torch.randn(M, K, device=dev()).to(torch.bfloat16)
    w = ops.PackedW((torch.randn(N, K, device=dev()) * 0.1).to(torch.bfloat16))
    resid = torch.randn(M, N, device=dev())
    resid[2500, 17] = 3.0e5                                    # Σ² of that wave column = 9e10 > 5e8
    resid[100, 5] = float("inf")
    out32, out_t = torch.empty(M, N, device=dev()), torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    sums = torch.zeros(M, 2, device=dev(), dtype=torch.int64)
    try:
        UF.reset_ln_flag()
        ops.gemm(a, w, bias=torch.zeros(N, device=dev()), resid=resid, out32=out32, out_t=out_t, rowsum=sums)
        with pytest.raises(UF.LnFoldRangeError):
            UF.poll_ln_flag(sync=True)
        s = ops.rowsum_to_float(sums)
        assert torch.isfinite(s).all() and float(s[2500, 1]) <= 12 * 5.0e8 * 1.001 and float(s[2500, 1]) >= 5.0e8 * 0.999
        ok = torch.ones(M, dtype=torch.bool, device=dev())
        ok[2500] = ok[100] = False
        want = torch.stack([out32.sum(1), (out32 * out32).sum(1)], 1)
        assert torch.allclose(s[ok], want[ok], rtol=2e-4, atol=1e-2)
    finally:
        UF.reset_ln_flag()
