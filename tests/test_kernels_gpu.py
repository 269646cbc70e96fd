"""-m gpu: each HIP kernel (through the C ABI / ctypes) against a plain PyTorch fp32 reference of
the same op, on seeded inputs.  Tolerances: fp32 mode 1e-4 rel (exact-fp32 MFMA, different
summation order); bf16 mode 1e-2 rel of max|ref| unless a test states otherwise."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from uia_hip import ops as o
    return o


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


TOL = {torch.float32: 2e-5, torch.bfloat16: 1.2e-2}


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(ops, dt):
    torch.manual_seed(0)
    M, N, K = 197 * 3, 384, 256
    a = torch.randn(M, K, device=dev()).to(dt)
    w = (torch.randn(N, K, device=dev()) * 0.1).to(dt)
    bias = torch.randn(N, device=dev())
    resid = torch.randn(M, N, device=dev())
    pre_ref = a.float() @ w.float().T + bias
    out_t, aux = torch.empty(M, N, device=dev(), dtype=dt), torch.empty(M, N, device=dev(), dtype=dt)
    ops.gemm(a, w, bias=bias, act="gelu", aux_out=aux, out_t=out_t)
    assert rel(aux, pre_ref) < TOL[dt] and rel(out_t, torch.nn.functional.gelu(pre_ref)) < TOL[dt]
    out32 = torch.empty(M, N, device=dev())
    ops.gemm(a, w, bias=bias, resid=resid, out32=out32)
    assert rel(out32, pre_ref + resid) < 1e-5 if dt == torch.float32 else 1e-5 + 1.0  # operands exact: fp32 out is exact-ish
    assert rel(out32, pre_ref + resid) < 2e-5
    ops.gemm(a, w, dact="quick_gelu", aux_in=aux, out32=out32)
    x = aux.float()
    s = torch.sigmoid(1.702 * x)
    assert rel(out32, (a.float() @ w.float().T) * (s * (1 + 1.702 * x * (1 - s)))) < 2e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13])
def test_gemm_every_tile_config_and_epilogue_mask(ops, dt, cfg):
    """Every tile instantiation (lockstep 1-5, ping-pong 6-7, ring 8-10) against torch on a ragged shape, through
    the six compile-time epilogue masks of the ring kernel (cfg 8) and the run-time epilogue (everything else)."""
    torch.manual_seed(10 + cfg)
    M, N, K = 2 * 256 + 77, 264, 192          # ragged in M and N for every tile size
    a = torch.randn(M, K, device=dev()).to(dt)
    w = (torch.randn(N, K, device=dev()) * 0.1).to(dt)
    bias, resid = torch.randn(N, device=dev()), torch.randn(M, N, device=dev())
    pre = a.float() @ w.float().T
    gelu = torch.nn.functional.gelu
    tol = TOL[dt]
    o_t = lambda: torch.full((M, N), float("nan"), device=dev(), dtype=dt)
    o32 = lambda: torch.full((M, N), float("nan"), device=dev())

    y = o32(); ops.gemm(a, w, bias=bias, resid=resid, out32=y, tile_cfg=cfg)                       # proj / fc2
    assert rel(y, pre + bias + resid) < 2e-5 and bool(torch.isfinite(y).all())
    y = o_t(); ops.gemm(a, w, out_t=y, tile_cfg=cfg)                                               # dgrad
    assert rel(y, pre) < tol
    y = o_t(); ops.gemm(a, w, bias=bias, out_t=y, tile_cfg=cfg)                                    # QKV
    assert rel(y, pre + bias) < tol
    y = o_t(); ops.gemm(a, w, bias=bias, act="gelu", out_t=y, tile_cfg=cfg)                        # fc1 (frozen tower)
    assert rel(y, gelu(pre + bias)) < tol
    y, aux = o_t(), o_t(); ops.gemm(a, w, bias=bias, act="gelu", aux_out=aux, out_t=y, tile_cfg=cfg)   # fc1 + stash
    assert rel(y, gelu(pre + bias)) < tol and rel(aux, pre + bias) < tol
    x = aux.float().requires_grad_(True)
    gelu(x).sum().backward()
    y = o_t(); ops.gemm(a, w, dact="gelu", aux_in=aux, out_t=y, tile_cfg=cfg)                      # fc2 dgrad through GELU'
    assert rel(y, pre * x.grad) < tol
    # combinations outside the specialised masks take the run-time epilogue
    rt = torch.randn(M, N, device=dev()).to(dt)
    y, y2 = o32(), o_t(); ops.gemm(a, w, bias=bias, resid=resid, resid_t=rt, out32=y, out_t=y2, alpha=0.5, tile_cfg=cfg)
    assert rel(y, 0.5 * pre + bias + resid + rt.float()) < 2e-5 and rel(y2, 0.5 * pre + bias + resid + rt.float()) < tol
    y = o_t(); ops.gemm(a, w, bias=bias, act="quick_gelu", out_t=y, tile_cfg=cfg)
    assert rel(y, (pre + bias) * torch.sigmoid(1.702 * (pre + bias))) < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_persistent_kernel_walks_many_tiles(ops, dt):
    """cfg 12 (persistent): more tiles than CUs, ragged edges, so every workgroup crosses the tile seam (next tile's
    LDS-DMA issued under the epilogue) at least once; checked against torch and, bit for bit, against cfg 8."""
    torch.manual_seed(5)
    M, N, K = 20 * 256 - 19, 17 * 256 - 40, 192          # 340 tiles on 256 CUs
    a = torch.randn(M, K, device=dev()).to(dt)
    w = (torch.randn(N, K, device=dev()) * 0.1).to(dt)
    bias, resid = torch.randn(N, device=dev()), torch.randn(M, N, device=dev())
    pre = a.float() @ w.float().T
    y12, y8 = torch.full((M, N), float("nan"), device=dev()), torch.full((M, N), float("nan"), device=dev())
    ops.gemm(a, w, bias=bias, resid=resid, out32=y12, tile_cfg=12)
    ops.gemm(a, w, bias=bias, resid=resid, out32=y8, tile_cfg=8)
    assert rel(y12, pre + bias + resid) < 2e-5 and torch.equal(y12, y8)
    z12, z8 = torch.empty(M, N, device=dev(), dtype=dt), torch.empty(M, N, device=dev(), dtype=dt)
    x12, x8 = torch.empty_like(z12), torch.empty_like(z8)
    ops.gemm(a, w, bias=bias, act="gelu", aux_out=x12, out_t=z12, tile_cfg=12)
    ops.gemm(a, w, bias=bias, act="gelu", aux_out=x8, out_t=z8, tile_cfg=8)
    assert rel(z12, torch.nn.functional.gelu(pre + bias)) < TOL[dt] and torch.equal(z12, z8) and torch.equal(x12, x8)
    # K shorter than the ring (2 sub-tiles): the Mona project2 shape
    a2, w2 = a[:, :64].contiguous() if dt == torch.bfloat16 else a[:, :32].contiguous(), None
    w2 = w[:, :a2.shape[1]].contiguous()
    ops.gemm(a2, w2, bias=bias, resid=resid, out32=y12, tile_cfg=12)
    assert rel(y12, a2.float() @ w2.float().T + bias + resid) < 2e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_patch_embed_epilogue(ops, dt):
    torch.manual_seed(1)
    B, G, K, D = 3, 16, 192, 128          # 3 images, 16 patches each
    a = torch.randn(B * G, K, device=dev()).to(dt)
    w = (torch.randn(D, K, device=dev()) * 0.1).to(dt)
    bias = torch.randn(D, device=dev())
    pos = torch.randn(G + 1, D, device=dev())
    x = torch.zeros(B, G + 1, D, device=dev())
    ops.gemm(a, w, bias=bias, resid=pos, resid_mod=G, resid_row_off=1, out_group=G, out32=x.view(-1, D))
    ref = (a.float() @ w.float().T + bias).view(B, G, D) + pos[1:]
    assert rel(x[:, 1:], ref) < 2e-5 and float(x[:, 0].abs().max()) == 0.0


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("L,mask", [(197, "none"), (77, "causal"), (256, "keypad"), (17, "none"), (50, "keypad"), (257, "none"), (130, "causal")])
def test_attention_fwd_bwd(ops, dt, L, mask):
    torch.manual_seed(2)
    B, H, D = 3, 4, 256
    qkv = (torch.randn(B * L, 3 * D, device=dev()) * 1.5).to(dt)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    keylen = torch.tensor([L, max(1, L // 3), max(2, L - 5)], device=dev(), dtype=torch.int32) if mask == "keypad" else None
    out = torch.empty(B * L, D, device=dev(), dtype=dt)
    lse = torch.empty(B, H, L, device=dev())
    ops.attn_fwd(q, k, v, out, B, H, L, lse=lse, mask=mask, keylen=keylen)

    qf = qkv.float().view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(True)   # [3,B,H,L,64]
    s = qf[0] @ qf[1].transpose(-1, -2) / 8.0
    if mask == "causal":
        s = s + torch.full((L, L), float("-inf"), device=dev()).triu_(1)
    if mask == "keypad":
        ar = torch.arange(L, device=dev())
        s = s.masked_fill(ar[None, None, None, :] >= keylen[:, None, None, None], float("-inf"))
    ref = torch.softmax(s, -1) @ qf[2]                                   # [B,H,L,64]
    ref_o = ref.permute(0, 2, 1, 3).reshape(B * L, D)
    assert rel(out, ref_o) < TOL[dt]
    assert rel(lse, torch.logsumexp(s, -1)) < (1e-5 if dt == torch.float32 else 2e-2)

    dout = torch.randn(B * L, D, device=dev()).to(dt)
    ref.backward(dout.float().view(B, L, H, 64).permute(0, 2, 1, 3))
    dqkv = torch.zeros(B * L, 3 * D, device=dev(), dtype=dt)
    ops.attn_bwd(q, k, v, out, dout, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, L, mask=mask, keylen=keylen)
    g = qf.grad.permute(1, 3, 0, 2, 4).reshape(B * L, 3 * D)              # back to [B*L, (3,H,64)]
    tol = 1e-4 if dt == torch.float32 else 2.5e-2
    for i, name in enumerate("qkv"):
        assert rel(dqkv[:, i * D:(i + 1) * D], g[:, i * D:(i + 1) * D]) < tol, name


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D", [768, 64, 1024])
def test_layernorm_fwd_bwd(ops, dt, D):
    torch.manual_seed(3)
    M = 37
    x = (torch.randn(M, D, device=dev()) * 2 + 0.5).requires_grad_(True)
    g, b = torch.randn(D, device=dev()), torch.randn(D, device=dev())
    y_t, y32 = torch.empty(M, D, device=dev(), dtype=dt), torch.empty(M, D, device=dev())
    ops.layernorm_fwd(x.detach(), g, b, 1e-6, y_t=y_t, y32=y32)
    ref = torch.nn.functional.layer_norm(x, (D,), g, b, 1e-6)
    assert rel(y32, ref) < 1e-5 and rel(y_t, ref) < TOL[dt]
    dy = torch.randn(M, D, device=dev()).to(dt)
    dres = torch.randn(M, D, device=dev())
    ref.backward(dy.float())
    dx32, dx_t = torch.empty(M, D, device=dev()), torch.empty(M, D, device=dev(), dtype=dt)
    ops.layernorm_bwd(dy, x.detach(), g, 1e-6, dres=dres, dx32=dx32, dx_t=dx_t)
    assert rel(dx32, x.grad + dres) < 2e-5 and rel(dx_t, x.grad + dres) < TOL[dt]


def test_layernorm_strided_rows(ops):
    torch.manual_seed(4)
    B, N, D = 5, 7, 128
    x = torch.randn(B, N, D, device=dev())
    g, b = torch.randn(D, device=dev()), torch.randn(D, device=dev())
    y = torch.empty(B, D, device=dev())
    ops.layernorm_fwd(x, g, b, 1e-5, y32=y, rows=B, ldx=N * D)
    assert rel(y, torch.nn.functional.layer_norm(x[:, 0], (D,), g, b, 1e-5)) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,I,J", [(1000, 128, 64), (197 * 5, 64, 192), (513, 64, 64)])
def test_wgrad(ops, dt, M, I, J):
    torch.manual_seed(5)
    a = torch.randn(M, I, device=dev()).to(dt)
    b = torch.randn(M, J, device=dev()).to(dt)
    dw = torch.zeros(I, J, device=dev())
    db = torch.zeros(I, device=dev())
    ops.wgrad(a, b, dw, db, alpha=0.5)
    assert rel(dw, 0.5 * a.float().T @ b.float()) < 2e-5
    assert rel(db, a.float().sum(0)) < 2e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_layout_helpers(ops, dt):
    torch.manual_seed(6)
    w = torch.randn(70, 130, device=dev())
    wt = torch.empty(130, 70, device=dev(), dtype=dt)
    ops.transpose_cast(w, wt)
    assert torch.equal(wt, w.T.to(dt))
    c = torch.empty(70, 132, device=dev(), dtype=dt)
    src = torch.randn(70, 132, device=dev())
    ops.cast(src, c, 0.5)
    assert torch.equal(c, (src * 0.5).to(dt))
    img = torch.rand(2, 3, 32, 32, device=dev())
    cols = torch.empty(2 * 16, 3 * 64, device=dev(), dtype=dt)
    ops.im2col(img, cols, 8)
    ref = torch.nn.functional.unfold(img, 8, stride=8).transpose(1, 2).reshape(32, 192)
    assert torch.equal(cols, ref.to(dt))
    x = torch.zeros(2, 5, 64, device=dev())
    cls, pos0 = torch.randn(64, device=dev()), torch.randn(64, device=dev())
    ops.fill_cls(x, cls, pos0)
    assert torch.equal(x[:, 0], (cls + pos0).expand(2, -1)) and float(x[:, 1:].abs().max()) == 0
    ids = torch.randint(0, 50, (3, 9), device=dev())
    table, pos, ty = torch.randn(50, 64, device=dev()), torch.randn(9, 64, device=dev()), torch.randn(64, device=dev())
    e = torch.empty(27, 64, device=dev())
    ops.embed(ids, table, pos, ty, e)
    assert rel(e.view(3, 9, 64), table[ids] + pos + ty) < 1e-6
    idx = torch.tensor([26, 0, 13], device=dev())
    gth = torch.empty(3, 64, device=dev())
    ops.gather_rows(e, idx, gth)
    assert torch.equal(gth, e[idx])


def test_rccl_communicator_single_rank_allreduce(ops):
    """The library's own RCCL path (unique id -> ncclCommInitRank -> ncclAllReduce on the compute stream -> destroy) with a
    world of one rank: the sum over one rank is the buffer itself.  (world_size 2 semantics are covered on CPU by
    tests/test_dp_gloo.py; multi-GPU boxes are not available to the test run.)"""
    uid = ops.comm_unique_id()
    assert isinstance(uid, bytes) and len(uid) >= 128
    ops.comm_init(0, 1, uid)
    try:
        assert ops.comm_world() == 1
        g = torch.randn(1343232, device=dev())
        ref = g.clone()
        ops.allreduce_sum(g)
        torch.cuda.synchronize()
        assert torch.equal(g, ref)
    finally:
        ops.comm_destroy()


@pytest.mark.parametrize("C", [2, 3])
def test_dicece_fused_loss_and_gradient(ops, C):
    """uia_dicece_fwd_bwd against the oracle's DiceCE restatement (MONAI semantics) and its autograd gradient, incl. an image
    with an empty foreground (the smooth terms carry it)."""
    from oracle import losses_ref
    g = torch.Generator().manual_seed(8)
    B, H, W = 5, 24, 40
    logits = torch.randn(B, C, H, W, generator=g) * 2
    label = torch.randint(0, C, (B, 1, H, W), generator=g).float()
    label[1] = 0
    lr_ = logits.clone().requires_grad_(True)
    ref = losses_ref.dice_ce(lr_, label)
    (3.0 * ref).backward()
    from src.losses.dice import DiceCELoss
    lg = logits.to(dev()).requires_grad_(True)
    loss = DiceCELoss()(lg, label.to(dev()))
    (3.0 * loss).backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    assert rel(lg.grad.cpu(), lr_.grad) < 1e-4


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_patch_embed_patch14(dt):
    """ViT-L/14 stem: 14x14 patches (588 values, not a multiple of the GEMM's K granule) through the zero-padded im2col + GEMM."""
    from uia_hip import functional as UF
    UF.set_compute_dtype(dt)
    g = torch.Generator().manual_seed(14)
    B, D, P, S = 2, 128, 14, 56
    img = torch.rand(B, 3, S, S, generator=g)
    w, cls, pos = torch.randn(D, 3, P, P, generator=g) * 0.05, torch.randn(D, generator=g), torch.randn((S // P) ** 2 + 1, D, generator=g)
    ref = torch.nn.functional.conv2d(img, w, None, stride=P).flatten(2).transpose(1, 2)
    ref = torch.cat([cls.expand(B, 1, D), ref], 1) + pos
    x = UF.PatchEmbedFn.apply(img.to(dev()), w.to(dev()), None, cls.to(dev()), pos.to(dev()), P)
    assert rel(x.cpu(), ref) < (2e-5 if dt == torch.float32 else 1e-2)


def test_error_behaviour_is_loud_and_specific(ops):
    """The C ABI returns a negative code with a message, surfaced as UiaError: misaligned K, empty problems, CPU tensors, a tile
    config that does not exist, an unsupported attention length — none of them falls back to anything."""
    from uia_hip._lib import UiaError
    a = torch.randn(64, 48, device=dev()).bfloat16()          # K = 48 is not a multiple of 64 for bf16
    w = torch.randn(64, 48, device=dev()).bfloat16()
    y = torch.empty(64, 64, device=dev(), dtype=torch.bfloat16)
    with pytest.raises(UiaError, match="K=48"):
        ops.gemm(a, w, out_t=y)
    a, w = torch.randn(64, 64, device=dev()).bfloat16(), torch.randn(64, 64, device=dev()).bfloat16()
    with pytest.raises(UiaError, match="unknown tile config"):
        ops.gemm(a, w, out_t=y, tile_cfg=99)
    with pytest.raises(UiaError, match="no output"):
        ops.gemm(a, w)
    with pytest.raises(UiaError):
        ops.gemm(a.cpu(), w.cpu(), out_t=y.cpu())
    qkv = torch.randn(2 * 300, 3 * 64, device=dev()).bfloat16()
    out = torch.empty(2 * 300, 64, device=dev(), dtype=torch.bfloat16)
    with pytest.raises(UiaError, match="L=300"):
        ops.attn_fwd(qkv[:, :64], qkv[:, 64:128], qkv[:, 128:], out, 2, 1, 300)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_single_row_and_single_image_edges(ops, dt):
    """M = 1 GEMM row, one image with one head and a one-token key-padding length."""
    torch.manual_seed(3)
    a = torch.randn(1, 128, device=dev()).to(dt)
    w = (torch.randn(72, 128, device=dev()) * 0.1).to(dt)
    y = torch.empty(1, 72, device=dev())
    ops.gemm(a, w, out32=y)
    assert rel(y, a.float() @ w.float().T) < (2e-5 if dt == torch.float32 else 2e-5)
    L = 9
    qkv = torch.randn(L, 192, device=dev()).to(dt)
    out = torch.empty(L, 64, device=dev(), dtype=dt)
    keylen = torch.tensor([1], device=dev(), dtype=torch.int32)
    ops.attn_fwd(qkv[:, :64], qkv[:, 64:128], qkv[:, 128:], out, 1, 1, L, mask="keypad", keylen=keylen)
    assert rel(out, qkv[:1, 128:].float().expand(L, 64)) < TOL[dt]      # every query attends to the single valid key
