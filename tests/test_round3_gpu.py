"""-m gpu, round 3: the fused Mona forward (csrc/mona_fused.hip) against the four unfused launches it replaces and against the oracle."""
import pytest
import torch

from oracle import mona_ref

pytestmark = pytest.mark.gpu
VARIANTS = ["baseline", "noise_aware", "freq_enhanced", "hybrid"]


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture(autouse=True)
def _mode():
    from uia_hip import functional as UF
    from uia_hip import ops
    UF.set_compute_dtype(torch.bfloat16)
    saved = ops.MONA_FUSED
    ops.MONA_FUSED = True
    yield
    ops.MONA_FUSED = saved
    UF.set_ln_fold(True)
    UF.clear_t_copies()


def _module(variant, D, g):
    from src.adapters import mona as M
    mod = M._VARIANTS[variant](D, 64)
    with torch.no_grad():
        for k, p in mod.named_parameters():
            if k.endswith(("norm.weight", "gammax", "freq_filter")):
                p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=g))
            elif k.endswith("gamma"):
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return mod


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("shape", [(768, 14, 6), (768, 14, 20), (128, 4, 5), (512, 14, 3)])
def test_fused_mona_forward_matches_unfused_launches_and_oracle(variant, shape):
    """uia_mona_fused_fwd (one launch, one workgroup per image) vs uia_mona_pre_fwd -> uia_gemm -> uia_mona_spatial_fwd -> uia_gemm on the same
    inputs, parameters and dropout mask: y, and what the backward is handed (u, t, d through dx and every parameter gradient).  The fused
    kernel keeps t in fp32 between project1 and the stencils, so the two differ by bf16 roundings of t — both are held to the oracle too.
    Reference: src/adapters/mona.py:96-151, 198-253, 319-362, 427-487."""
    from uia_hip import functional as UF
    from uia_hip import ops
    D, w, B = shape
    assert ops.mona_fused_ok(torch.bfloat16, D, w, w, 64)
    g = torch.Generator().manual_seed(100 + D + w + B)
    mod = _module(variant, D, g)
    N = 1 + w * w
    x = torch.randn(B, N, D, generator=g) * 1.5
    dy = torch.randn(B, N, D, generator=g)
    keep = (torch.rand(B, N, 64, generator=g) >= 0.1).to(torch.uint8)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in mod.named_parameters()}
    xr = x.clone().requires_grad_(True)
    yr = mona_ref.forward(xr, P, variant, (w, w), keep_mask=keep.bool(), p_drop=0.1)
    yr.backward(dy)
    mod = mod.to(dev()).train()
    outs = {}
    for fused in (False, True):
        ops.MONA_FUSED = fused
        mod.zero_grad(set_to_none=True)
        mod.keep_mask = keep.to(dev())
        xg = x.to(dev()).requires_grad_(True)
        y = mod(xg.permute(1, 0, 2), (w, w)).permute(1, 0, 2)
        y.backward(dy.to(dev()))
        UF.clear_t_copies()
        outs[fused] = (y.detach().clone(), xg.grad.clone(), {k: p.grad.clone() for k, p in mod.named_parameters()})
    mod.keep_mask = None
    for fused in (False, True):
        y, dx, gr = outs[fused]
        assert rel(y, yr) < 1e-2 and rel(dx, xr.grad) < 3e-2, (fused, rel(y, yr), rel(dx, xr.grad))
        for k, e in ((k, rel(gr[k], P[k].grad)) for k in gr):
            assert e < (0.2 if "noise_estimator" in k else 0.12), (fused, k, e)
    # fused vs unfused: same arithmetic up to the bf16 rounding of t and the fp32 summation order of the row statistics
    assert rel(outs[True][0], outs[False][0]) < 6e-3, rel(outs[True][0], outs[False][0])
    assert rel(outs[True][1], outs[False][1]) < 2e-2


@pytest.mark.parametrize("variant", ["freq_enhanced", "hybrid"])
def test_fused_mona_outputs_for_the_folded_layernorm(variant):
    """Past 2048 rows the adapter also leaves the T copy of y (K-blocked) and its row sums (Σ, Σ²) for the LayerNorm folded into the next
    block's QKV GEMM: both must describe the fp32 y the kernel stored (row sums exactly: they are fixed-point integers)."""
    from uia_hip import functional as UF
    from uia_hip import ops
    D, w, B = 768, 14, 12                                      # 2364 rows
    g = torch.Generator().manual_seed(7)
    mod = _module(variant, D, g).to(dev()).eval()
    x = (torch.randn(B, 1 + w * w, D, generator=g) * 1.5).to(dev())
    UF.set_ln_fold(True)
    UF.clear_t_copies()
    with torch.no_grad():
        y = mod(x.permute(1, 0, 2), (w, w)).permute(1, 0, 2).contiguous()
    M = B * (1 + w * w)
    hit = UF.take_rows(y, torch.bfloat16)
    assert hit is not None, "the fused forward did not publish its rows"
    y_t, sums = hit
    y2 = y.view(M, D)
    rows = y_t.t.permute(1, 0, 2).reshape(M, D) if ops.is_kb(y_t) else y_t.view(M, D)
    assert torch.equal(rows, y2.to(torch.bfloat16))
    s = ops.rowsum_to_float(sums)
    want = torch.stack([y2.double().sum(1), (y2.double() ** 2).sum(1)], 1).float()
    assert torch.allclose(s, want, rtol=2e-5, atol=2e-3), float((s - want).abs().max())
    assert UF.poll_ln_flag(sync=True) & 2 == 0


def test_fused_mona_rejects_unsupported_shapes_loudly():
    from uia_hip import ops
    assert not ops.mona_fused_ok(torch.bfloat16, 1024, 16, 16, 64)       # ViT-L width: the unfused launches
    assert not ops.mona_fused_ok(torch.float32, 768, 14, 14, 64)         # fp32 parity mode: the unfused launches
    x = torch.zeros(2, 17, 96, device=dev())
    w1 = torch.zeros(64, 96, device=dev(), dtype=torch.bfloat16)
    w2 = torch.zeros(96, 64, device=dev(), dtype=torch.bfloat16)
    v = torch.zeros(96, device=dev())
    z = torch.zeros(64, device=dev())
    sp = dict(conv1_w=torch.zeros(64, 9, device=dev()), conv1_b=z, conv2_w=torch.zeros(64, 25, device=dev()), conv2_b=z,
              conv3_w=torch.zeros(64, 49, device=dev()), conv3_b=z, proj_w=torch.zeros(64, 64, device=dev()), proj_b=z)
    with pytest.raises(ops.UiaError, match="use the unfused launches"):
        ops.mona_fused_fwd("baseline", 2, 4, 4, x, v, v, v, v, w1, z, w2, v, sp, torch.empty_like(x))


# ------------------------------------------------------------------------------------------------ the pipelined GEMM epilogue
@pytest.mark.parametrize("M,N,K", [(2500, 768, 128), (4353, 776, 64), (2177, 2304, 192)])
def test_pipelined_epilogue_equals_inline_epilogue_bit_for_bit(M, N, K):
    """Round 3: the compile-time epilogue masks that read an operand or leave row sums (aux_in with GELU', fp32 / T residual, deferred-LayerNorm
    residual, row sums) request their rows a chunk ahead and issue the row-sum atomics after the last store.  Same arithmetic in the same order:
    on RAGGED shapes (last row panel and last column tile partial) every output must equal, bit for bit, what tile cfg 10 — the same ring kernel
    with the run-time epilogue and its in-line loads — produces, on the 256-row tiles (8), on both half-height configs (13, 14) and on the four-wave kernel of round 4 (25 / 26: two epilogue calls per wave tile)."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    dt = torch.bfloat16
    a = torch.randn(M, K, generator=g).to(dev()).to(dt)
    w = ops.PackedW((torch.randn(N, K, generator=g) * K ** -0.5).to(dev()).to(dt))
    bias = torch.randn(N, generator=g).to(dev())
    resid = torch.randn(M, N, generator=g).to(dev())
    resid_t = torch.randn(M, N, generator=g).to(dev()).to(dt)
    aux = (torch.randn(M, N, generator=g) * 1.5).to(dev()).to(dt)
    lnw, lnb = (1.0 + 0.2 * torch.randn(N, generator=g)).to(dev()), (0.1 * torch.randn(N, generator=g)).to(dev())
    mean = resid.mean(1)
    stats = torch.stack([mean, (resid.var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
    sums_in = ops.rowsum_from_float(torch.stack([resid.sum(1), (resid * resid).sum(1)], 1))

    csum = w.row.float().sum(1).contiguous()
    aux_o = lambda: torch.full((M, N), float("nan"), device=dev(), dtype=dt)

    def run(cfg, kind):
        if kind in ("plain", "bias", "gelu", "gelu_stash", "lnfold_bias", "lnfold_gelu", "lnfold_gelu_stash", "qgelu", "qgelu_stash"):     # T outputs only
            ot, ax = aux_o(), aux_o()
            kw = {}
            if kind != "plain":
                kw["bias"] = bias
            if "gelu" in kind:
                kw["act"] = "quick_gelu" if kind.startswith("q") else "gelu"
            if "stash" in kind:
                kw["aux_out"] = ax
            if kind.startswith("lnfold"):
                kw["lnfold"] = (sums_in, csum, N, 1e-5)                # row sums of SOME rows: any finite statistics exercise the arithmetic
            ops.gemm(a, w, out_t=ot, tile_cfg=cfg, **kw)
            torch.cuda.synchronize()
            return ot, ax, torch.zeros(1, dtype=torch.int64)
        o32 = torch.full((M, N), float("nan"), device=dev())
        ot = torch.full((M, N), float("nan"), device=dev(), dtype=dt)
        rs = torch.zeros(M, 2, device=dev(), dtype=torch.int64)
        if kind == "dgelu":                                            # mask 136
            ops.gemm(a, w, dact="gelu", aux_in=aux, out_t=ot, tile_cfg=cfg)
        elif kind == "dqgelu":                                         # mask 136 | EPI_QUICK: fc2 data gradient of an OpenAI-CLIP tower
            ops.gemm(a, w, dact="quick_gelu", aux_in=aux, out_t=ot, tile_cfg=cfg)
        elif kind == "resid32":                                        # mask 81
            ops.gemm(a, w, bias=bias, resid=resid, out32=o32, tile_cfg=cfg)
        elif kind == "residT":                                         # mask 97
            ops.gemm(a, w, bias=bias, resid_t=resid_t, out32=o32, tile_cfg=cfg)
        elif kind == "fold_producer":                                  # mask 721
            ops.gemm(a, w, bias=bias, resid=resid, out32=o32, out_t=ot, rowsum=rs, tile_cfg=cfg)
        elif kind == "resid_ln":                                       # mask 337: (mean, rstd) statistics
            ops.gemm(a, w, bias=bias, resid=resid, out32=o32, resid_ln=(stats, lnw, lnb), tile_cfg=cfg)
        elif kind == "resid_ln_sums_fold_producer":                    # mask 977: statistics from row sums, and row sums out
            ops.gemm(a, w, bias=bias, resid=resid, out32=o32, out_t=ot, rowsum=rs, resid_ln=(sums_in, lnw, lnb, N, 1e-5), tile_cfg=cfg)
        torch.cuda.synchronize()
        return o32, ot, rs

    for kind in ("plain", "bias", "gelu", "gelu_stash", "lnfold_bias", "lnfold_gelu", "lnfold_gelu_stash", "qgelu", "qgelu_stash", "dqgelu",
                 "dgelu", "resid32", "residT", "fold_producer", "resid_ln", "resid_ln_sums_fold_producer"):
        ref = run(10, kind)
        for cfg in (8, 13, 14, 25, 26, 27, 28, 29):                      # 25 / 26: the four-wave kernel (round 4), compile-time and run-time epilogue; 27 / 28 / 29: its register-staged forms (round 5)
            got = run(cfg, kind)
            for name, x, y in zip(("out32", "outT", "rowsum"), got, ref):
                same = torch.equal(x, y) if x.dtype == torch.int64 else torch.equal(torch.nan_to_num(x.float(), nan=-7.0), torch.nan_to_num(y.float(), nan=-7.0))
                assert same, (kind, cfg, name)
        # and against torch, so that "equal" is not "equally wrong"
        pre = a.float() @ w.row.float().T
        if kind == "fold_producer":
            want = pre + bias + resid
            assert float((ref[0] - want).abs().max()) <= 3e-5 * float(want.abs().max())
            s = ops.rowsum_to_float(ref[2])
            assert torch.allclose(s[:, 0], ref[0].sum(1), rtol=1e-4, atol=1e-2)


# ------------------------------------------------------------------------------------------------ small fp32 GEMMs on 32 x 64 tiles
@pytest.mark.parametrize("M,N,K", [(256, 640, 768), (256, 512, 640), (100, 776, 96), (1, 72, 32)])
def test_small_fp32_gemm_tile_cfg_21_equals_the_128_tile_bit_for_bit(M, N, K):
    """The fp32 head projections ([B, 512-768] x K = 512-768) moved from 128 x 128 tiles (8-12 workgroups, bound by their CUs' fp32 MFMA rate)
    to 32 x 64 tiles (tile cfg 21, the launcher's choice for fp32 shapes with fewer than 64 of the big tiles).  Every output element sees the
    same MFMA sequence over K, so the results must be bit-identical to tile cfg 3 — ragged M / N, bias, GELU, residual included."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N + K)
    a = torch.randn(M, K, generator=g).to(dev())
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev())
    bias = torch.randn(N, generator=g).to(dev())
    resid = torch.randn(M, N, generator=g).to(dev())
    assert ops.auto_tile_cfg(M, N, K, 4) == 21 and ops.auto_tile_cfg(M, N, K, 2) == 3
    for kw in ({}, {"bias": bias}, {"bias": bias, "act": "gelu"}, {"bias": bias, "resid": resid}):
        outs = []
        for cfg in (3, 21, 0):
            o = torch.full((M, N), float("nan"), device=dev())
            ops.gemm(a, w, out32=o, tile_cfg=cfg, **kw)
            torch.cuda.synchronize()
            outs.append(o)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), sorted(kw)
        want = a.double() @ w.double().T
        if "bias" in kw:
            want = want + bias.double()
        if "act" in kw:
            want = torch.nn.functional.gelu(want)
        if "resid" in kw:
            want = want + resid.double()
        assert rel(outs[1], want.float()) < 2e-5, sorted(kw)


# ------------------------------------------------------------------------------------------------ LoRA: dropout without a pass of its own
@pytest.mark.parametrize("M,K", [(2500, 1024), (300, 128), (16, 64)])
def test_dropout_on_the_stream_kernels_operand_equals_the_dropout_pass_bit_for_bit(M, K):
    """LoRA input dropout (lora.py:82-83) applied to the A operand of the N = 64 stream kernel in flight: t = drop(x)·Aᵀ and the by-product
    drop(x) must equal, bit for bit, uia_dropout followed by the plain launch (same seed, same eight-wide generator)."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(M + K)
    x = torch.randn(M, K, generator=g).to(dev()).to(torch.bfloat16)
    a = (torch.randn(64, K, generator=g) * K ** -0.5).to(dev()).to(torch.bfloat16)
    for p_drop, seed in ((0.1, 12345), (0.5, (1 << 40) + 77)):
        xd_ref = torch.empty_like(x)
        ops.dropout(x, xd_ref, p_drop, seed)
        t_ref = torch.empty(M, 64, device=dev(), dtype=torch.bfloat16)
        ops.gemm(xd_ref, a, out_t=t_ref, tile_cfg=16)
        xd = torch.full_like(x, float("nan"))
        t = torch.empty_like(t_ref)
        ops.gemm(x, a, out_t=t, drop=("a", p_drop, seed, xd))
        t2 = torch.empty_like(t_ref)
        ops.gemm(x, a, out_t=t2, drop=("a", p_drop, seed))                # without the by-product
        torch.cuda.synchronize()
        assert torch.equal(xd, xd_ref) and torch.equal(t, t_ref) and torch.equal(t2, t_ref)
        kept = (xd_ref.float() != 0).float().mean().item()
        assert abs(kept - (1.0 - p_drop)) < 0.02 + 2.0 / (M * K) ** 0.5
    with pytest.raises(ops.UiaError, match="drop_where"):
        ops.gemm(x, torch.randn(128, K, device=dev()).to(torch.bfloat16), out_t=torch.empty(M, 128, device=dev(), dtype=torch.bfloat16), drop=("a", 0.1, 1))


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
@pytest.mark.parametrize("M,N", [(2500, 1024), (321, 136)])
def test_dropout_in_the_gemm_epilogue_is_the_backward_of_the_same_mask(mode, M, N):
    """dx += drop(s·q·A) in ONE launch (run-time epilogue, drop = ("acc", p, seed)) against s·q·A followed by uia_dropout(accumulate): the zero
    pattern must be the mask uia_dropout draws for (seed, m·N + n), the kept values agree to rounding (the fused form does not round s·q·A to T
    before scaling it)."""
    from uia_hip import ops
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    g = torch.Generator(device="cpu").manual_seed(M + N)
    q = torch.randn(M, 64, generator=g).to(dev()).to(dt)
    at = (torch.randn(N, 64, generator=g) * 0.2).to(dev()).to(dt)
    dx0 = torch.randn(M, N, generator=g).to(dev()).to(dt)
    p_drop, seed, s = 0.25, 987654321, 1.75
    ones = torch.ones(M, N, device=dev(), dtype=dt)
    kept = torch.empty_like(ones)
    ops.dropout(ones, kept, p_drop, seed)
    keep = kept.float() > 0
    # the mask alone: zero residual
    zero = torch.zeros(M, N, device=dev(), dtype=dt)
    only = torch.empty_like(zero)
    ops.gemm(q, at, alpha=s, resid_t=zero, out_t=only, drop=("acc", p_drop, seed))
    prod = (q.float() @ at.float().T) * s
    want = torch.where(keep, prod / (1.0 - p_drop), torch.zeros_like(prod))
    assert torch.equal(only.float() != 0, keep & (want.to(dt).float() != 0))
    assert rel(only, want) < (1e-2 if mode == "bf16" else 1e-5)
    # accumulate onto dx, in place, against the two-launch form
    ref = dx0.clone()
    dxd = torch.empty_like(ref)
    ops.gemm(q, at, alpha=s, out_t=dxd)
    ops.dropout(dxd, ref, p_drop, seed, accumulate=True)
    got = dx0.clone()
    ops.gemm(q, at, alpha=s, resid_t=got, out_t=got, drop=("acc", p_drop, seed))
    torch.cuda.synchronize()
    assert rel(got, ref) < (2e-2 if mode == "bf16" else 1e-5)
    assert torch.equal(got[~keep], dx0[~keep])                           # dropped positions keep dx untouched


def test_wgrad_into_the_unpadded_gradient_and_padded_factor_packs():
    """uia_wgrad_ex: 64-padded operands, gradient of the factor's own shape ([out, r] / [r, in], r = 16), accumulating; and the batched
    uia_pack_weights forms of a LoRA factor zero-padded to rank 64 (WEIGHTS.get(..., pad_rows_to / pad_cols_to) on a trainable parameter)."""
    from uia_hip import functional as UF
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K, r = 1500, 320, 256, 16
    dy = torch.randn(M, N, generator=g).to(dev()).to(torch.bfloat16)
    t = torch.zeros(M, 64, device=dev(), dtype=torch.bfloat16)
    t[:, :r] = torch.randn(M, r, generator=g).to(dev()).to(torch.bfloat16)
    full = torch.zeros(N, 64, device=dev())
    ops.wgrad(dy, t, full, alpha=0.5)
    small = torch.full((N, r), 3.0, device=dev())
    ops.wgrad(dy, t, small, alpha=0.5)
    torch.cuda.synchronize()
    assert rel(small - 3.0, full[:, :r]) < 1e-5 and float(full[:, r:].abs().max()) == 0.0
    qv = torch.zeros(M, 64, device=dev(), dtype=torch.bfloat16)
    qv[:, :r] = torch.randn(M, r, generator=g).to(dev()).to(torch.bfloat16)
    xd = torch.randn(M, K, generator=g).to(dev()).to(torch.bfloat16)
    fullA = torch.zeros(64, K, device=dev())
    ops.wgrad(qv, xd, fullA)
    smallA = torch.zeros(r, K, device=dev())
    ops.wgrad(qv, xd, smallA)
    assert rel(smallA, fullA[:r]) < 1e-5
    assert rel(smallA, qv[:, :r].float().T @ xd.float()) < 2e-3
    for dt in (torch.bfloat16, torch.float32):
        A = torch.nn.Parameter(torch.randn(r, K, generator=g).to(dev()))
        Bm = torch.nn.Parameter(torch.randn(N, r, generator=g).to(dev()))
        a_f, a_b = UF.WEIGHTS.get(A, dt, pad_rows_to=64), UF.WEIGHTS.get(A, dt, transpose=True, pad_rows_to=64)
        b_f, b_b = UF.WEIGHTS.get(Bm, dt, pad_cols_to=64), UF.WEIGHTS.get(Bm, dt, transpose=True, pad_cols_to=64)
        Ap = torch.zeros(64, K, device=dev()); Ap[:r] = A.detach()
        Bp = torch.zeros(N, 64, device=dev()); Bp[:, :r] = Bm.detach()
        assert torch.equal(a_f.row, Ap.to(dt)) and torch.equal(a_b.row, Ap.T.contiguous().to(dt))
        assert torch.equal(b_f.row, Bp.to(dt)) and torch.equal(b_b.row, Bp.T.contiguous().to(dt))
        gk = 64 // torch.empty(0, dtype=dt).element_size()
        assert torch.equal(b_f.kblocked(), Bp.to(dt).view(N, 64 // gk, gk).permute(1, 0, 2).contiguous())
        with torch.no_grad():
            A.mul_(2.0)                                                   # the version counter moves: the next get() refreshes in place
        assert torch.equal(UF.WEIGHTS.get(A, dt, pad_rows_to=64).row, (2.0 * Ap).to(dt))


@pytest.mark.parametrize("mode,D,H", [("fp32", 128, 2), ("bf16", 128, 2), ("bf16", 256, 4)])      # D = 256: the K-extension form of the rank update (ops.LORA_KEXT)
@pytest.mark.parametrize("p_drop", [0.0, 0.2])
def test_lora_attention_half_as_one_node_equals_the_composition(mode, D, H, p_drop):
    """PlainMultiheadAttentionLoRA.block_half (LoraAttnHalfFn: one frozen GEMM and one data-gradient GEMM for q, k, v, gradients of h summed
    inside the launches) against LayerNormFn -> 3 x LoraLinearFn -> AttentionFn -> LoraLinearFn under the same dropout seeds: output, input
    gradient and every factor / bias gradient."""
    from uia_hip import functional as UF
    from src.adapters.lora import PlainMultiheadAttentionLoRA
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[mode]
    UF.set_compute_dtype(dt)
    g = torch.Generator().manual_seed(23)
    B, L, r = (3, 50, 8) if D == 128 else (6, 50, 8)                      # D = 256: 300 rows >= 256 for the K extension
    mha = torch.nn.MultiheadAttention(D, H)
    mod = PlainMultiheadAttentionLoRA(mha, enable_lora=["q", "k", "v", "o"], r=r, lora_alpha=16, dropout_rate=p_drop)
    ln = torch.nn.LayerNorm(D)
    with torch.no_grad():
        for k, p in list(mod.named_parameters()) + list(ln.named_parameters()):
            # D = 256: smaller weights — the two forms then differ by bf16 roundings of q, k, v (one rounding of x·Wᵀ + s·t·Bᵀ instead of two), and attention
            # logits of +-50 would turn those into percent-level differences of the softmax
            p.copy_((1.0 if k == "weight" and p.dim() == 1 else 0.0) + (0.15 if D == 128 else 0.05) * torch.randn(p.shape, generator=g))
    mod, ln = mod.to(dev()).train(), ln.to(dev())
    for p in ln.parameters():
        p.requires_grad_(False)
    x = torch.randn(B, L, D, generator=g).to(dev())
    dy = torch.randn(B, L, D, generator=g).to(dev())
    names = [k for k, p in mod.named_parameters() if p.requires_grad]

    def run(fused):
        for p in mod.parameters():
            p.grad = None
        UF.set_dropout_seed(31)
        UF.clear_t_copies()
        xx = x.clone().requires_grad_(True)
        if fused:
            y = mod.block_half(xx, ln, B, L, None)
        else:
            h = UF.LayerNormFn.apply(xx, ln.weight, ln.bias, ln.eps).view(B * L, D)
            y = mod.rows_forward(h, B, L, None, resid32=xx.view(B * L, D)).view(B, L, D)
        y.backward(dy)
        return y.detach(), xx.grad.detach(), {k: dict(mod.named_parameters())[k].grad.detach().clone() for k in names}

    y0, dx0, g0 = run(False)
    y1, dx1, g1 = run(True)
    tol = 1e-5 if mode == "fp32" else 2e-2
    assert rel(y1, y0) < tol and rel(dx1, dx0) < tol
    assert sorted(g1) == sorted(g0) and len(g1) == 12
    gmax = max(float(v.abs().max()) for v in g0.values())
    for k in names:                                                        # k_proj.bias has an exactly zero gradient (softmax rows sum to one): both sides hold
        scale = max(float(g0[k].abs().max()), 0.05 * gmax)                 # rounding noise there, which is compared on the scale of the other gradients
        assert float((g1[k] - g0[k]).abs().max()) < tol * scale, k


@pytest.mark.parametrize("M,N", [(32896, 1024), (2500, 640), (2049, 128)])
def test_k64_stream_kernel_equals_the_tiled_kernel(M, N):
    """Tile cfg 23 (K = 64 read-modify-write stream: the LoRA rank update and its data gradient) against the 128 x 256 tiles of cfg 14 with the
    run-time epilogue: T residual -> T output in place, fp32 residual -> fp32 output in place, alpha, bias, dropout on the accumulator."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(M + N)
    dt = torch.bfloat16
    t = torch.randn(M, 64, generator=g).to(dev()).to(dt)
    w = (torch.randn(N, 64, generator=g) * 0.2).to(dev()).to(dt)
    bias = torch.randn(N, generator=g).to(dev())
    y_t = torch.randn(M, N, generator=g).to(dev()).to(dt)
    y32 = torch.randn(M, N, generator=g).to(dev())
    for kw in ({}, {"bias": bias}, {"drop": ("acc", 0.1, 4242)}):
        outs = []
        for cfg in (14, 23, 0):
            a, b = y_t.clone(), y32.clone()
            ops.gemm(t, w, alpha=1.5, resid_t=a, out_t=a, tile_cfg=cfg, **kw)
            ops.gemm(t, w, alpha=1.5, resid=b, out32=b, tile_cfg=cfg, **kw)
            torch.cuda.synchronize()
            outs.append((a, b))
        for a, b in outs[1:]:
            assert torch.equal(a, outs[0][0]) and torch.equal(b, outs[0][1]), sorted(kw)
    want = y32 + 1.5 * (t.float() @ w.float().T)
    got = y32.clone()
    ops.gemm(t, w, alpha=1.5, resid=got, out32=got)
    assert rel(got, want) < 1e-5


@pytest.mark.parametrize("M,N,K,slices", [(700, 768, 3072, 3), (128, 1024, 4096, 8), (333, 520, 2048, 5)])
def test_split_k_tail_launch_equals_the_whole_k_launch(M, N, K, slices):
    """The M tail of a large GEMM as two launches (K slices into a workspace, then sum + epilogue) against the same tiles with the whole K chain:
    every epilogue the step's tails use — fp32 residual, plain, GELU with stash, folded LayerNorm, GELU', row sums.  One fp32 sum per element
    either way; only its association differs."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    dt = torch.bfloat16
    a = torch.randn(M, K, generator=g).to(dev()).to(dt)
    w = ops.PackedW((torch.randn(N, K, generator=g) * K ** -0.5).to(dev()).to(dt))
    bias = torch.randn(N, generator=g).to(dev())
    resid = torch.randn(M, N, generator=g).to(dev())
    aux = (torch.randn(M, N, generator=g) * 1.5).to(dev()).to(dt)
    sums_in = ops.rowsum_from_float(torch.stack([resid.sum(1), (resid * resid).sum(1)], 1))
    csum = w.row.float().sum(1).contiguous()
    ws = torch.zeros(-(-M // 128) * -(-N // 256) * 128 * 256, device=dev())          # zero before the first use; every pair of launches leaves it zero

    def run(kind, split):
        o32 = torch.full((M, N), float("nan"), device=dev())
        ot = torch.full((M, N), float("nan"), device=dev(), dtype=dt)
        ax = torch.full((M, N), float("nan"), device=dev(), dtype=dt)
        rs = torch.zeros(M, 2, device=dev(), dtype=torch.int64)
        kw = {"resid32": dict(bias=bias, resid=resid, out32=o32), "plain": dict(out_t=ot), "gelu_stash_fold": dict(bias=bias, act="gelu", aux_out=ax, out_t=ot, lnfold=(sums_in, csum, N, 1e-5)),
              "dgelu": dict(dact="gelu", aux_in=aux, out_t=ot), "producer": dict(bias=bias, resid=resid, out32=o32, out_t=ot, rowsum=rs)}[kind]
        if split:
            for phase in (1, 2):
                ops._gemm_one(a, w, tile_cfg=13 | slices << 16 | phase << 22, splitk_ws=ws, **kw)
        else:
            ops._gemm_one(a, w, tile_cfg=13, **kw)
        torch.cuda.synchronize()
        return o32, ot, ax, ops.rowsum_to_float(rs)

    for kind in ("resid32", "plain", "gelu_stash_fold", "dgelu", "producer"):
        ref, got = run(kind, False), run(kind, True)
        for name, x, y in zip(("out32", "outT", "aux_out", "rowsum"), got, ref):
            if torch.isnan(y).all():
                assert torch.isnan(x).all(), (kind, name)
                continue
            assert not torch.isnan(x).any(), (kind, name)
            assert rel(x, y) < (1e-2 if x.dtype == dt else 2e-5), (kind, name, rel(x, y))
    assert float(ws.abs().max()) == 0.0
    with pytest.raises(ops.UiaError, match="split K"):
        ops._gemm_one(a, w, out_t=torch.empty(M, N, device=dev(), dtype=dt), tile_cfg=3 | slices << 16 | 1 << 22, splitk_ws=ws)


@pytest.mark.parametrize("with_bias", [False, True])
def test_lora_gradients_straight_into_the_flat_buffer(with_bias):
    """With the engine's FlatAdapterOptimizer the factors' .grad are views of ONE flat fp32 buffer and the LoRA nodes accumulate into them with
    uia_wgrad_ex (no staging buffer, autograd gets None): the flat buffer after a backward must equal the gradients autograd returns without it,
    through LoraAttnHalfFn and through LoraLinearFn, with the biases frozen (how the fine-tune scripts mark parameters) and trainable (quirk C-4)."""
    from uia_hip import functional as UF
    from uia_hip.engine import FlatAdapterOptimizer
    from src.adapters.lora import LinearLoRA, PlainMultiheadAttentionLoRA
    UF.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(29)
    B, L, D, H, r = 2, 40, 128, 2, 8
    mod = PlainMultiheadAttentionLoRA(torch.nn.MultiheadAttention(D, H), enable_lora=["q", "k", "v", "o"], r=r, lora_alpha=16, dropout_rate=0.1)
    lin = LinearLoRA(torch.nn.Linear(D, 192), r=r, lora_alpha=16, dropout_rate=0.1)
    ln = torch.nn.LayerNorm(D)
    with torch.no_grad():
        for k, p in list(mod.named_parameters()) + list(lin.named_parameters()):
            p.copy_(0.15 * torch.randn(p.shape, generator=g))
    mod, lin, ln = mod.to(dev()).train(), lin.to(dev()).train(), ln.to(dev())
    for p in ln.parameters():
        p.requires_grad_(False)
    named = [(f"a.{k}", p) for k, p in mod.named_parameters()] + [(f"l.{k}", p) for k, p in lin.named_parameters()]
    for k, p in named:
        p.requires_grad_("lora" in k or (with_bias and k.endswith("bias")))
    train = [(k, p) for k, p in named if p.requires_grad]
    x = torch.randn(B, L, D, generator=g).to(dev())
    dy = torch.randn(B, L, D, generator=g).to(dev())
    dz = torch.randn(B * L, 192, generator=g).to(dev()).to(torch.bfloat16)

    def run():
        UF.set_dropout_seed(41)
        UF.clear_t_copies()
        y = mod.block_half(x.clone().requires_grad_(True), ln, B, L, None)
        z = lin.apply_rows(y.detach().view(B * L, D).to(torch.bfloat16).requires_grad_(True))
        torch.autograd.backward([y, z], [dy, dz])

    for _, p in train:
        p.grad = None
    run()
    want = {k: p.grad.detach().clone() for k, p in train}
    assert all(v.abs().max() > 0 for v in want.values())
    opt = FlatAdapterOptimizer(train, lr=1e-3)
    opt.zero_grad()
    assert all(UF._is_flat_grad(p) for _, p in train)
    run()
    assert opt.grad_views_intact()
    for k, p in train:
        assert rel(p.grad, want[k]) < 1e-5, k
    run()                                                                  # accumulates: twice the gradient
    for k, p in train:
        assert rel(p.grad, 2 * want[k]) < 1e-5, k


def test_tail_split_with_split_k_through_the_host_scheduler():
    """ops.gemm at the ViT-L/14 + LoRA row count (32 896 = 128.5 panels of 256 rows): the host cuts a 128-row tail off the 256x256 launch and, at
    K = 4096, runs that tail split over K (two launches through a zeroed workspace the launches keep zero) — against the un-split launch."""
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    M, N, K = 32896, 1024, 4096
    assert ops.tail_split_rows(M, N, 256) == 32768 and ops.tail_k_slices(M - 32768, N, K, 2, 256) == 8
    assert ops.tail_k_slices(6912, 768, 3072, 2, 256) == 0 and ops.tail_k_slices(128, 1024, 1024, 2, 256) == 0      # the headline step's 81-tile tails; short K
    a = torch.randn(M, K, generator=g).to(dev()).to(torch.bfloat16)
    w = ops.PackedW((torch.randn(N, K, generator=g) * K ** -0.5).to(dev()).to(torch.bfloat16))
    bias = torch.randn(N, generator=g).to(dev())
    resid = torch.randn(M, N, generator=g).to(dev())
    o_ref, o = torch.empty(M, N, device=dev()), torch.empty(M, N, device=dev())
    ops.gemm(a, w, bias=bias, resid=resid, out32=o_ref, tile_cfg=8)
    for _ in range(2):                                                     # twice: the second use finds the workspace as the first left it
        o.fill_(float("nan"))
        ops.gemm(a, w, bias=bias, resid=resid, out32=o)
        torch.cuda.synchronize()
        assert torch.equal(o[:32768], o_ref[:32768])
        assert rel(o[32768:], o_ref[32768:]) < 2e-5 and not torch.isnan(o).any()
    assert all(float(t.abs().max()) == 0.0 for t in ops._SPLITK_WS.values())


@pytest.mark.parametrize("M", [300, 2500, 8224])
def test_k_extension_gemm_equals_frozen_gemm_plus_rank_update(M):
    """uia_gemm_desc.A2 / K2: y = [x | t]·[W | s·B]ᵀ in ONE K loop (three column groups, each with its own t — a fused q | k | v projection — and
    the single-group fp32-residual form of the output projection) against x·Wᵀ + s·t·Bᵀ evaluated in fp32 from the same bf16 operands.  The
    [W | s·B] operand comes from WEIGHTS.get_lora_ext, its factor part refreshed by the batched pack launch when a factor changes."""
    from uia_hip import functional as UF
    from uia_hip import ops
    g = torch.Generator(device="cpu").manual_seed(M)
    D, r, s = 256, 16, 8.0
    dt = torch.bfloat16
    ws = [torch.nn.Parameter((torch.randn(D, D, generator=g) * D ** -0.5).to(dev()), requires_grad=False) for _ in range(3)]
    Bs = [torch.nn.Parameter((torch.randn(D, r, generator=g) * 0.1).to(dev())) for _ in range(3)]
    bias = torch.randn(3 * D, generator=g).to(dev())
    x = torch.randn(M, D, generator=g).to(dev()).to(dt)
    t_all = torch.zeros(3, M, 64, device=dev(), dtype=dt)
    t_all[:, :, :r] = torch.randn(3, M, r, generator=g).to(dev()).to(dt)
    resid = torch.randn(M, D, generator=g).to(dev())

    def want_qkv():
        cols = [x.float() @ w.detach().to(dt).float().T + s * (t_all[i, :, :r].float() @ (Bs[i].detach()).to(dt).float().T) for i, w in enumerate(ws)]
        return torch.cat(cols, 1) + bias

    ext = UF.WEIGHTS.get_lora_ext(tuple(ws), tuple(Bs), s, dt)
    assert tuple(ext.kb.shape) == ((D + 64) // 32, 3 * D, 32) and float(ext.kb[D // 32:, :, r:].abs().max()) == 0.0
    y = torch.full((M, 3 * D), float("nan"), device=dev(), dtype=dt)
    ops.gemm(x, ext, bias=bias, out_t=y, a2=(t_all, D))
    torch.cuda.synchronize()
    assert rel(y, want_qkv()) < 1e-2
    with torch.no_grad():
        Bs[1].mul_(-0.5)                                                   # a factor changes: the next get refreshes its columns of [W | s·B]
    ext2 = UF.WEIGHTS.get_lora_ext(tuple(ws), tuple(Bs), s, dt)
    assert ext2 is ext
    ops.gemm(x, ext, bias=bias, out_t=y, a2=(t_all, D))
    torch.cuda.synchronize()
    assert rel(y, want_qkv()) < 1e-2
    # single group, fp32 residual -> fp32 output (the output projection)
    ext_o = UF.WEIGHTS.get_lora_ext((ws[0],), (Bs[0],), s, dt)
    o = torch.full((M, D), float("nan"), device=dev())
    ops.gemm(x, ext_o, bias=bias[:D].contiguous(), resid=resid, out32=o, a2=(t_all[0], 0))
    torch.cuda.synchronize()
    want = resid + bias[:D] + x.float() @ ws[0].detach().to(dt).float().T + s * (t_all[0, :, :r].float() @ Bs[0].detach().to(dt).float().T)
    assert rel(o, want) < 2e-3
    with pytest.raises(ops.UiaError):
        ops.gemm(x, ext, bias=bias, out_t=y)                              # no second operand


@pytest.mark.parametrize("resid", [False, True])
def test_linear_lora_with_the_rank_update_in_the_k_loop_equals_the_two_launch_form(resid):
    """LoraLinearFn with ops.LORA_KEXT (one launch [x | t]·[W | s·B]ᵀ) against the frozen GEMM followed by the rank-update launch, same dropout seed:
    output (T, and fp32 with the fused residual), input gradient and the factor gradients — at a row count with a split M tail (2 x 256·128 + 100)."""
    from uia_hip import functional as UF
    from uia_hip import ops
    from src.adapters.lora import LinearLoRA
    UF.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(61)
    M, I, O, r = 65636, 256, 512, 16
    mod = LinearLoRA(torch.nn.Linear(I, O), r=r, lora_alpha=32, dropout_rate=0.1)
    with torch.no_grad():
        for k, p in mod.named_parameters():
            p.copy_(0.05 * torch.randn(p.shape, generator=g))
    mod = mod.to(dev()).train()
    x = torch.randn(M, I, generator=g).to(dev()).to(torch.bfloat16)
    res = torch.randn(M, O, generator=g).to(dev()) if resid else None
    dy = torch.randn(M, O, generator=g).to(dev())
    dy = dy if resid else dy.to(torch.bfloat16)
    outs = []
    for kext in (False, True):
        ops.LORA_KEXT = kext
        try:
            for p in mod.parameters():
                p.grad = None
            UF.set_dropout_seed(9)
            xx = x.clone().requires_grad_(True)
            y = mod.apply_rows(xx, res)
            y.backward(dy)
            outs.append((y.detach(), xx.grad.detach(), mod.w_lora_A.grad.detach().clone(), mod.w_lora_B.grad.detach().clone()))
        finally:
            ops.LORA_KEXT = True
    for name, a, b in zip(("y", "dx", "dA", "dB"), outs[1], outs[0]):
        assert rel(a, b) < 1e-2, name
    assert torch.equal(outs[1][1], outs[0][1])                             # the backward does not depend on the forward's form


def test_lora_attention_half_without_an_input_gradient():
    """First block of a tower behind a frozen embedding: x does not require a gradient, the node skips the data-gradient GEMM and the LayerNorm
    backward and must still produce the factor gradients of the run in which x does require one."""
    from uia_hip import functional as UF
    from src.adapters.lora import PlainMultiheadAttentionLoRA
    UF.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(71)
    B, L, D, H, r = 3, 33, 128, 2, 8
    mod = PlainMultiheadAttentionLoRA(torch.nn.MultiheadAttention(D, H), enable_lora=["q", "k", "v", "o"], r=r, lora_alpha=16, dropout_rate=0.1)
    ln = torch.nn.LayerNorm(D)
    with torch.no_grad():
        for k, p in mod.named_parameters():
            p.copy_(0.1 * torch.randn(p.shape, generator=g))
    mod, ln = mod.to(dev()).train(), ln.to(dev())
    for p in ln.parameters():
        p.requires_grad_(False)
    x = torch.randn(B, L, D, generator=g).to(dev())
    dy = torch.randn(B, L, D, generator=g).to(dev())
    got = []
    for need in (True, False):
        for p in mod.parameters():
            p.grad = None
        UF.set_dropout_seed(5)
        UF.clear_t_copies()
        xx = x.clone().requires_grad_(need)
        mod.block_half(xx, ln, B, L, None).backward(dy)
        assert (xx.grad is not None) == need
        got.append({k: p.grad.detach().clone() for k, p in mod.named_parameters() if p.grad is not None})
    assert sorted(got[0]) == sorted(got[1]) and len(got[0]) == 12
    for k in got[0]:
        assert torch.equal(got[0][k], got[1][k]), k
