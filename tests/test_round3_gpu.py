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
    with the run-time epilogue and its in-line loads — produces, on the 256-row tiles (8) and on both half-height configs (13, 14)."""
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

    def run(cfg, kind):
        o32 = torch.full((M, N), float("nan"), device=dev())
        ot = torch.full((M, N), float("nan"), device=dev(), dtype=dt)
        rs = torch.zeros(M, 2, device=dev(), dtype=torch.int64)
        if kind == "dgelu":                                            # mask 136
            ops.gemm(a, w, dact="gelu", aux_in=aux, out_t=ot, tile_cfg=cfg)
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

    for kind in ("dgelu", "resid32", "residT", "fold_producer", "resid_ln", "resid_ln_sums_fold_producer"):
        ref = run(10, kind)
        for cfg in (8, 13, 14):
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
