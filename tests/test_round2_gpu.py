"""-m gpu: the data-parallel engine on its product path (flat buffers + RCCL + fused clip/AdamW) against the oracle, and the
robustness fixes of round 2: optimiser-independent Mona gradients, LoRA ranks above 64, bounded embedding lookups,
leading-dimension checks, the `hw_shapes=None` Mona path, LoRA-branch dropout with a known mask."""
import math
import os

import pytest
import torch

from oracle import lora_ref, losses_ref, mona_ref, text_ref, train_ref, vit_ref

pytestmark = pytest.mark.gpu
TOY = dict(embed_dim=128, vision_cfg=dict(img_size=32, patch_size=8, embed_dim=128, depth=3, num_heads=2),
           text_cfg=dict(vocab_size=120, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                         max_position_embeddings=40))


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
    UF.clear_t_copies()


def _randomize_mona(model, seed, scale=0.06):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "mona" not in k:
                continue
            if k.endswith(("norm.weight", "gammax", "freq_filter")):
                p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=g))
            elif k.endswith("gamma"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(scale * torch.randn(p.shape, generator=g))


def _toy_model(variant="hybrid", seed=5):
    from src.adapters import inject_mona_variant_to_open_clip
    from src.third_party.biomedclip.model import create_biomedclip
    model = create_biomedclip(config=TOY, seed=seed)
    for p in model.parameters():
        p.requires_grad_(False)
    inject_mona_variant_to_open_clip(model, variant=variant, bottleneck_dim=64)
    _randomize_mona(model, seed + 1)
    for k, p in model.named_parameters():
        p.requires_grad_("mona" in k)
    return model.eval()


def _toy_batch(seed, B=6, L=24):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(B, 3, 32, 32, generator=g)
    ids = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(4, L, (1,), generator=g))
        ids[b, :n] = torch.randint(5, 120, (n,), generator=g)
        ids[b, 0], ids[b, n - 1] = 2, 3
    return images, ids


def test_engine_step_through_rccl_vs_oracle():
    """FlatAdapterOptimizer + init_data_parallel(force_comm) + contrastive_step: the product path of one data-parallel update — the
    flat gradient buffer goes through uia_allreduce_sum on a one-rank RCCL communicator (the identity), then the fused clip+AdamW —
    against oracle/train_ref.clip_and_adamw on the oracle's gradients.  Two micro-batches = the reference's accumulation
    (finetune.py:287-302); two consecutive updates check the Adam state and the refreshed operand copies of the adapter weights."""
    from uia_hip import functional as UF
    from uia_hip import ops
    from uia_hip.engine import FlatAdapterOptimizer, contrastive_step, init_data_parallel
    from src.losses import InfoNCELoss
    UF.set_compute_dtype(torch.float32)
    variant = "hybrid"
    model = _toy_model(variant)
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [k for k in P if "mona" in k]
    images, ids = _toy_batch(4, B=8)
    mona = dict(variant=variant, hw=(4, 4))
    loss_fn = lambda Pq, im, tk: train_ref.biomedclip_loss(Pq, im, tk, mona=mona, heads=2, text_heads=2)
    model = model.to(dev())
    opt = FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95), weight_decay=0.01, max_norm=1.0)
    assert opt.names == names and opt.grad_views_intact()
    init_data_parallel(opt, force_comm=True)
    try:
        assert opt.collective and ops.comm_world_initialised() and ops.comm_world() == 1
        m = {k: torch.zeros_like(P[k]) for k in names}
        v = {k: torch.zeros_like(P[k]) for k in names}
        for step in (1, 2):
            gref, lref = train_ref.grads_of(loss_fn, P, names, [(images[:4], ids[:4]), (images[4:], ids[4:])])
            pr = {k: P[k] for k in names}
            norm = train_ref.clip_and_adamw(pr, gref, m, v, step, 1e-3, (0.9, 0.95), 1e-8, 0.01, 1.0)
            loss = contrastive_step(model, InfoNCELoss(0.07), opt, images.to(dev()), ids.to(dev()), micro_batches=2, overlap_text=False)
            assert abs(float(loss) - lref) < 2e-3 * max(1.0, abs(lref))
            assert abs(opt.grad_norm() - norm) < 2e-3 * norm
            flat = opt.unflatten(opt.p)
            # Adam's step is lr * m_hat / (sqrt(v_hat) + 1e-8): an element whose gradient is itself ~1e-8 turns a last-bit difference
            # of the two gradient computations into a visible fraction of lr, so the update is compared in units of lr (the fused
            # kernel's arithmetic on IDENTICAL gradients is test_adamw_clip_step_vs_oracle, to 1e-5)
            for k in names:
                diff = (flat[k].detach().cpu() - pr[k]).abs()
                far = float((diff > 0.1 * 1e-3).float().mean())           # elements whose near-zero gradient changed sign: up to 2·lr apart
                assert float(diff.mean()) < 0.01 * 1e-3 * step and far < 0.01, (step, k, float(diff.max()), float(diff.mean()), far)
                assert flat[k].data_ptr() == dict(model.named_parameters())[k].data_ptr()      # the module reads the flat buffer
    finally:
        ops.comm_destroy()


def test_global_batch_gather_through_rccl_allgather_world1():
    """§8 f4: GatherFeaturesFn routed through uia_allgather on a one-rank RCCL communicator (init_data_parallel(force_comm=True)) — the
    collective the multi-GPU job uses, not the world-1 copy: the gathered rows equal the input bit for bit, the backward returns the
    local slice, and a global-loss step through it equals the local-loss step (one rank: same batch)."""
    from uia_hip import functional as UF
    from uia_hip import ops
    from uia_hip.engine import FlatAdapterOptimizer, GatherFeaturesFn, contrastive_step, init_data_parallel
    from src.losses import InfoNCELoss
    UF.set_compute_dtype(torch.float32)
    calls = []
    real = ops.allgather
    try:
        init_data_parallel(None, force_comm=True)
        assert ops.comm_world_initialised() and ops.comm_world() == 1
        ops.allgather = lambda a, b: (calls.append(a.numel()), real(a, b))[1]
        x = torch.randn(6, 32, device=dev(), requires_grad=True)
        y = GatherFeaturesFn.apply(x, 0, 1)
        assert calls == [6 * 32] and torch.equal(y.detach(), x.detach())
        (y * torch.arange(6, device=dev())[:, None]).sum().backward()
        assert torch.equal(x.grad, torch.arange(6, device=dev(), dtype=torch.float32)[:, None].expand(6, 32))
        outs = []
        for flag in (False, True):
            model = _toy_model("baseline", seed=11).to(dev())
            opt = FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-3)
            init_data_parallel(opt, force_comm=True)
            images, ids = _toy_batch(4)
            n0 = len(calls)
            loss = contrastive_step(model, InfoNCELoss(0.07), opt, images.to(dev()), ids.to(dev()), overlap_text=False, global_loss=flag)
            assert len(calls) - n0 == (2 if flag else 0)               # image and text features each went through the collective
            outs.append(float(loss))
        assert abs(outs[0] - outs[1]) < 1e-6 * abs(outs[0])
    finally:
        ops.allgather = real
        ops.comm_destroy()


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_contrastive_step_on_two_streams_equals_one_stream(mode):
    """engine.contrastive_step(streams=2): the batch's towers run as two slices on two HIP streams (forward and backward), the loss is
    still one InfoNCE over all pairs — same loss, same gradients (float atomics reorder the weight-gradient sums), same update."""
    from uia_hip import functional as UF
    from uia_hip.engine import FlatAdapterOptimizer, contrastive_step
    from src.losses import InfoNCELoss
    UF.set_compute_dtype(torch.float32 if mode == "fp32" else torch.bfloat16)
    images, ids = _toy_batch(21, B=8)
    outs = []
    for streams in (1, 2):
        model = _toy_model("hybrid", seed=13).to(dev())
        opt = FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-3)
        UF.set_dropout_seed(7)
        opt.snapshot_grads = True                                        # the guarded update zeroes the accumulator: keep a copy of what it consumed
        loss = contrastive_step(model, InfoNCELoss(0.07), opt, images.to(dev()), ids.to(dev()), overlap_text=False, streams=streams)
        torch.cuda.synchronize()
        outs.append((float(loss), opt.last_g, opt.grad_norm()))
    tol = 1e-5 if mode == "fp32" else 2e-2
    assert abs(outs[0][0] - outs[1][0]) < tol * max(1.0, abs(outs[0][0]))
    g1, g2 = outs[0][1], outs[1][1]
    assert float((g1 - g2).norm() / g1.norm()) < (1e-4 if mode == "fp32" else 5e-2), float((g1 - g2).norm() / g1.norm())
    assert abs(outs[0][2] - outs[1][2]) < (1e-4 if mode == "fp32" else 5e-2) * outs[0][2]


def test_mona_gradients_do_not_depend_on_the_flat_optimiser_idiom():
    """ADVICE r01 (functional.py:220): `out = model(x); optimizer.zero_grad(set_to_none=True); loss.backward()` used to orphan the
    gradient buffers captured at forward time.  The .grad views are now looked up at BACKWARD time and the direct path is an
    optimiser opt-in, so (a) dropping the views between forward and backward, (b) a plain torch.optim loop without the flat
    optimiser and (c) torch.autograd.grad all see the same gradients as the direct path."""
    from uia_hip import functional as UF
    from uia_hip.engine import FlatAdapterOptimizer
    UF.set_compute_dtype(torch.float32)
    images, _ = _toy_batch(9)
    runs = {}
    for how in ("direct", "dropped_views", "plain_autograd", "autograd_grad"):
        model = _toy_model("freq_enhanced", seed=7).to(dev())
        params = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
        opt = FlatAdapterOptimizer(params, lr=1e-3) if how in ("direct", "dropped_views") else None
        fi = model.encode_image(images.to(dev()))
        loss = fi.square().sum()
        if how == "dropped_views":
            for _, p in params:
                p.grad = None                                            # optimizer.zero_grad(set_to_none=True) between forward and backward
        if how == "autograd_grad":
            gs = torch.autograd.grad(loss, [p for _, p in params])
            runs[how] = torch.cat([g.flatten() for g in gs]).cpu()
            continue
        loss.backward()
        assert all(p.grad is not None for _, p in params), how
        runs[how] = torch.cat([p.grad.detach().flatten() for _, p in params]).cpu()
        if how == "dropped_views":
            before = opt.p.clone()
            opt.all_reduce()
            opt.step()                                                   # folds the foreign .grad tensors back into the flat buffer first
            assert opt.grad_views_intact() and not torch.equal(before, opt.p)
            assert torch.equal(torch.cat([v.flatten() for v in opt.unflatten(opt.g).values()]).cpu(), runs[how])
    scale = float(runs["direct"].abs().max())
    for how in ("dropped_views", "plain_autograd", "autograd_grad"):
        # float atomics: two runs differ in the last bits (relative to the sums that cancel, not to each element)
        assert float((runs[how] - runs["direct"]).abs().max()) < 1e-4 * scale, how


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_lora_rank_above_64_vs_oracle(mode):
    """--lora_r 96 (the CLI accepts any rank): the rank-form buffers are padded to 128 columns; it used to write N = r columns
    into 64-wide rows.  Forward + every trainable gradient of LinearLoRA against the oracle's dense-BA form (lora.py:78-90)."""
    from uia_hip import functional as UF
    from src.adapters.lora import LinearLoRA
    UF.set_compute_dtype({"fp32": torch.float32, "bf16": torch.bfloat16}[mode])
    g = torch.Generator().manual_seed(13)
    r, alpha, M, I, O = 96, 32, 333, 128, 192
    lin = torch.nn.Linear(I, O)
    mod = LinearLoRA(lin, r=r, lora_alpha=alpha, dropout_rate=0.0)
    with torch.no_grad():
        mod.weight.copy_(torch.randn(O, I, generator=g) * 0.1)
        mod.bias.copy_(torch.randn(O, generator=g) * 0.1)
        mod.w_lora_A.copy_(torch.randn(r, I, generator=g) * 0.05)
        mod.w_lora_B.copy_(torch.randn(O, r, generator=g) * 0.05)
    x = torch.randn(M, I, generator=g)
    dy = torch.randn(M, O, generator=g)
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in mod.named_parameters()}
    xr = x.clone().requires_grad_(True)
    yr = lora_ref.linear_lora(xr, leaves["weight"], leaves["bias"], leaves["w_lora_A"], leaves["w_lora_B"], r, alpha)
    yr.backward(dy)
    mod = mod.to(dev()).eval()
    xg = x.to(dev()).requires_grad_(True)
    y = mod(xg)
    y.backward(dy.to(dev()))
    tol, gtol = (1e-3, 1e-3) if mode == "fp32" else (1e-2, 3e-2)
    assert rel(y, yr) < tol and rel(xg.grad, xr.grad) < gtol
    for k in ("w_lora_A", "w_lora_B", "bias"):
        assert rel(getattr(mod, k).grad, leaves[k].grad) < gtol, k


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_lora_branch_dropout_with_known_mask_vs_oracle(mode):
    """LinearLoRA in training mode (lora.py:82-83: dropout on the LoRA branch's input only).  The kernel's mask is a counter hash of
    (seed, element); the test reads that mask back with the same seed (uia_dropout on a tensor of ones) and hands it to the oracle."""
    from uia_hip import functional as UF
    from uia_hip import ops
    from src.adapters.lora import LinearLoRA
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16}[mode]
    UF.set_compute_dtype(dt)
    g = torch.Generator().manual_seed(17)
    r, alpha, p_drop, M, I, O = 8, 16, 0.25, 300, 128, 192
    mod = LinearLoRA(torch.nn.Linear(I, O), r=r, lora_alpha=alpha, dropout_rate=p_drop)
    with torch.no_grad():
        mod.weight.copy_(torch.randn(O, I, generator=g) * 0.1)
        mod.bias.copy_(torch.randn(O, generator=g) * 0.1)
        mod.w_lora_A.copy_(torch.randn(r, I, generator=g) * 0.2)
        mod.w_lora_B.copy_(torch.randn(O, r, generator=g) * 0.2)
    x = torch.randn(M, I, generator=g)
    dy = torch.randn(M, O, generator=g)
    # the mask the module is about to draw: first dropout call after set_dropout_seed(77)
    UF.set_dropout_seed(77)
    seed = UF._next_seed()
    ones = torch.ones(M, I, device=dev(), dtype=dt)
    kept = torch.empty_like(ones)
    ops.dropout(ones, kept, p_drop, seed)
    keep = (kept.float().cpu() > 0).float()
    assert 0.6 < float(keep.mean()) < 0.9 and abs(float(kept.float().max()) - 1.0 / (1.0 - p_drop)) < 1e-2
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in mod.named_parameters()}
    xr = x.clone().requires_grad_(True)
    yr = lora_ref.linear_lora(xr, leaves["weight"], leaves["bias"], leaves["w_lora_A"], leaves["w_lora_B"], r, alpha, keep_mask=keep, p_drop=p_drop)
    yr.backward(dy)
    mod = mod.to(dev()).train()
    UF.set_dropout_seed(77)
    xg = x.to(dev()).requires_grad_(True)
    y = mod(xg)
    y.backward(dy.to(dev()))
    tol, gtol = (1e-3, 1e-3) if mode == "fp32" else (1e-2, 3e-2)
    assert rel(y, yr) < tol and rel(xg.grad, xr.grad) < gtol
    for k in ("w_lora_A", "w_lora_B", "bias"):
        assert rel(getattr(mod, k).grad, leaves[k].grad) < gtol, k
    mod.eval()                                                           # eval: no dropout (lora.py:82 `if self.training`)
    y_eval = mod(x.to(dev()))
    assert rel(y_eval, lora_ref.linear_lora(x, leaves["weight"], leaves["bias"], leaves["w_lora_A"], leaves["w_lora_B"], r, alpha)) < tol


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("variant", ["baseline", "noise_aware", "freq_enhanced", "hybrid"])
def test_mona_without_hw_shapes_vs_oracle(mode, variant):
    """reference mona.py:140-144 (and :242-246, :351-355, :476-480): forward(x) with hw_shapes=None has no CLS token — all n tokens
    form an int(sqrt(n))-square grid and go through the spatial operator.  Injected models never call it; a direct caller may."""
    from uia_hip import functional as UF
    from src.adapters import mona as M
    UF.set_compute_dtype({"fp32": torch.float32, "bf16": torch.bfloat16}[mode])
    g = torch.Generator().manual_seed(19)
    D, N, B = 128, 16, 3                                                 # 16 tokens = a 4x4 grid, no CLS
    mod = M._VARIANTS[variant](D, 64)
    with torch.no_grad():
        for k, p in mod.named_parameters():
            p.copy_((1.0 if k.endswith(("norm.weight", "gammax", "freq_filter")) else 0.0) + 0.2 * torch.randn(p.shape, generator=g))
    P = {k: v.detach().clone().requires_grad_(True) for k, v in mod.named_parameters()}
    x = torch.randn(N, B, D, generator=g)
    dy = torch.randn(N, B, D, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = mona_ref.forward(xr.permute(1, 0, 2), P, variant, None).permute(1, 0, 2)
    yr.backward(dy)
    mod = mod.to(dev()).eval()
    xg = x.to(dev()).requires_grad_(True)
    y = mod(xg)                                                          # hw_shapes omitted
    y.backward(dy.to(dev()))
    tol, gtol = (1e-3, 1e-3) if mode == "fp32" else (1e-2, 4e-2)
    assert rel(y, yr) < tol and rel(xg.grad, xr.grad) < gtol
    for k, p in mod.named_parameters():
        assert rel(p.grad, P[k].grad) < gtol * (2.0 if (mode == "bf16" and "noise_estimator" in k) else 1.0), k
    with pytest.raises(RuntimeError, match="square"):
        mod(torch.randn(11, B, D, device=dev()))                         # 11 tokens: the reference's reshape fails too


def test_embedding_lookups_are_bounded():
    """nn.Embedding raises for an id outside the table; the kernels used to read past it.  Now: NaN row (caught by the loops'
    non-finite-loss check), L beyond the position table is rejected, and the backward skips the bad id."""
    from uia_hip import ops
    from uia_hip._lib import UiaError
    V, Pn, D = 50, 16, 64
    table, pos, typ = torch.randn(V, D, device=dev()), torch.randn(Pn, D, device=dev()), torch.randn(D, device=dev())
    ids = torch.tensor([[1, 2, 49, 50], [3, -1, 0, 7]], device=dev())
    out = torch.empty(8, D, device=dev())
    ops.embed(ids, table, pos, typ, out)
    bad = torch.isnan(out).all(dim=1).cpu().tolist()
    assert bad == [False, False, False, True, False, True, False, False]
    good = out[[0, 1, 2, 4, 6, 7]]
    want = (table[ids.flatten()[[0, 1, 2, 4, 6, 7]]] + pos[[0, 1, 2, 0, 2, 3]] + typ)
    assert torch.allclose(good, want)
    with pytest.raises(UiaError, match="position table"):
        ops.embed(torch.zeros(1, Pn + 1, dtype=torch.long, device=dev()), table, pos, typ, torch.empty(Pn + 1, D, device=dev()))
    dtab = torch.zeros(V, D, device=dev())
    ops.embed_bwd(ids, torch.ones(8, D, device=dev()), dtab, pad_id=0)
    assert float(dtab.sum()) == 5 * D                                    # 8 rows minus the pad id, the negative id and the id == V


def test_gemm_rejects_rows_narrower_than_n():
    from uia_hip import ops
    from uia_hip._lib import UiaError
    a = torch.randn(256, 64, device=dev()).bfloat16()
    w = torch.randn(128, 64, device=dev()).bfloat16()
    with pytest.raises(UiaError):
        ops.gemm(a, w, out_t=torch.empty(256, 64, device=dev(), dtype=torch.bfloat16))      # 64-wide rows for a 128-wide result
    with pytest.raises(UiaError):
        ops.gemm(a, w, out32=torch.empty(128, 128, device=dev()))                             # too few rows
    big = torch.empty(256, 256, device=dev(), dtype=torch.bfloat16)
    ops.gemm(a, w, out_t=big[:, :128])                                                        # a wider leading dimension is fine
    assert rel(big[:, :128], a.float() @ w.float().T) < 1e-2


def test_finetune_entry_point_shards_data_and_decides_collectively(tmp_path, monkeypatch):
    """The entry point's data-parallel plumbing on one GPU: DataModule(rank, world) hands disjoint shards to the ranks, and the
    single-process run (world 1) still follows the reference loop (updates per epoch, checkpoint)."""
    from src.datasets.finetune import DataModule
    from src.models.biomedclip import finetune
    monkeypatch.chdir(tmp_path)
    cfg = ("dict(embed_dim=128, vision_cfg=dict(img_size=32, patch_size=8, embed_dim=128, depth=2, num_heads=2), "
           "text_cfg=dict(vocab_size=30000, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, max_position_embeddings=64))")
    argv = ["--method", "mona", "--synthetic", "--synthetic_train", "64", "--synthetic_val", "16", "--img_size", "32", "--batch_size", "8",
            "--accumulation_steps", "2", "--epochs", "2", "--lr", "1e-3", "--dtype", "bf16", "--exp", "dp", "--model_config", cfg]
    args = finetune.get_args(argv)
    seen = []
    for rank in range(2):
        dm = DataModule(args, rank=rank, world=2)
        loader = dm.train_dataloader()
        dm.set_epoch(0)
        assert len(loader) == 64 // 2 // 8
        seen.append([t for _, texts in loader for t in texts])
    assert len(seen[0]) == len(seen[1]) == 32 and not (set(seen[0]) & set(seen[1]))            # disjoint shards, equal sizes
    out = finetune.main(argv)
    assert out["world"] == 1 and out["updates"] == 2 * 4 and math.isfinite(out["best_val"])
    assert (tmp_path / "runs" / "dp" / "best_model.pth").exists()


@pytest.mark.parametrize("case", ["A_empty_gt", "B_empty_pred", "C_three_class", "D_all_foreground"])
def test_dicece_kernel_vs_independent_float64_vectors(golden, case):
    """uia_dicece_fwd_bwd (loss + d loss / d logits fused) and src.losses.dice.dice_per_image against the float64 known-answer
    vectors of oracle/gen_dice_golden.py (MONAI DiceCELoss / compute_dice semantics, reference clipseg/segmentation.py:84,
    tools.py:185-206): empty ground truth, empty prediction, three classes, H != W."""
    from src.losses.dice import DiceCELoss, dice_per_image
    z = golden("dicece_cases")
    logits = z[case + "_logits"].to(dev()).requires_grad_(True)
    label = z[case + "_label"][:, None].float().to(dev())
    loss = DiceCELoss()(logits, label)
    loss.backward()
    assert abs(float(loss) - float(z[case + "_loss"])) < 1e-5 * abs(float(z[case + "_loss"]))
    g = z[case + "_grad"].float()
    assert float((logits.grad.cpu() - g).abs().max()) < 1e-4 * float(g.abs().max()) + 1e-7
    if case + "_dice" in z:
        d, want = dice_per_image(logits.detach(), label).cpu(), z[case + "_dice"]
        assert torch.equal(torch.isnan(d), torch.isnan(want)) and torch.allclose(torch.nan_to_num(d), torch.nan_to_num(want), atol=1e-12)


def test_pack_weights_all_forms_and_cache_refresh():
    """uia_pack_weights (one launch for every operand form of the small trainable matrices) against torch, and the WeightCache
    contract around it: forms are rewritten in place by bump() (the optimiser step), a parameter changed by other means is
    repacked on its next use, frozen and padded weights keep the per-tensor path."""
    from uia_hip import functional as UF, ops
    torch.manual_seed(3)
    dt = torch.bfloat16
    g = 32
    ps = [torch.nn.Parameter(torch.randn(64, 768, device=dev())), torch.nn.Parameter(torch.randn(768, 64, device=dev())),
          torch.nn.Parameter(torch.randn(96, 40, device=dev()))]                       # the last one: no K-blocked forms (40 % 32, 96 % 32 == 0 only)
    cache = UF.WeightCache()

    def check(p):
        f, b = cache.get(p, dt), cache.get(p, dt, transpose=True)
        ref = p.detach().to(dt)
        assert torch.equal(f.row, ref) and torch.equal(b.row, ref.t().contiguous())
        R, Cc = p.shape
        if Cc % g == 0:
            assert torch.equal(f.kblocked(), ref.view(R, Cc // g, g).permute(1, 0, 2).contiguous())
        if R % g == 0:
            assert torch.equal(b.kblocked(), ref.t().contiguous().view(Cc, R // g, g).permute(1, 0, 2).contiguous())
        return f, b

    first = [check(p) for p in ps]
    with torch.no_grad():                                       # "fused optimiser": data changes, version counters do not
        for p in ps:
            p.data.add_(0.25)
    cache.bump()
    for p, (f0, b0) in zip(ps, first):
        f, b = check(p)
        assert f.row.data_ptr() == f0.row.data_ptr() and b.row.data_ptr() == b0.row.data_ptr()      # refreshed in place
    cache.bump()                                                # second step: the resident descriptor table is re-used
    [check(p) for p in ps]
    with torch.no_grad():
        ps[0].mul_(2.0)                                         # in-place op: version counter moves, no bump
    check(ps[0])
    frozen = torch.randn(64, 768, device=dev())
    assert cache.get(frozen, dt) is cache.get(frozen, dt) and id(frozen) not in cache._packed
    padded = cache.get(ps[2], dt, pad_rows_to=128)
    assert tuple(padded.shape) == (128, 40) and torch.equal(padded.row[:96], ps[2].detach().to(dt))


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
@pytest.mark.parametrize("M,N,K,cfg", [(700, 768, 768, 0), (4096, 768, 3072, 8), (300, 256, 128, 3)])
def test_gemm_deferred_layernorm_residual_is_bit_identical(mode, M, N, K, cfg):
    """uia_gemm with resid_ln_* (the residual is LayerNorm(resid rows), statistics from uia_layernorm_fwd_stats) must give exactly
    what the two-step form gives: LayerNorm writes its fp32 output, the GEMM adds it (HF BertSelfOutput / BertOutput)."""
    from uia_hip import ops
    torch.manual_seed(11)
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    a = torch.randn(M, K, device=dev()).to(dt)
    w = (torch.randn(N, K, device=dev()) * K ** -0.5).to(dt)
    bias = torch.randn(N, device=dev())
    raw = torch.randn(M, N, device=dev()) * 3 + 0.5
    lw, lb = torch.randn(N, device=dev()), torch.randn(N, device=dev())
    y32 = torch.empty_like(raw)
    y_t = torch.empty(M, N, device=dev(), dtype=dt)
    stats = torch.empty(M, 2, device=dev())
    ops.layernorm_fwd(raw, lw, lb, 1e-12, y_t=y_t, y32=y32, stats=stats)
    mu = raw.double().mean(1)
    assert float((stats[:, 0].double() - mu).abs().max()) < 1e-5
    assert float((stats[:, 1].double() * (raw.double().var(1, unbiased=False) + 1e-12).sqrt() - 1).abs().max()) < 1e-5
    two_step, fused = torch.empty(M, N, device=dev()), torch.empty(M, N, device=dev())
    ops.gemm(a, w, bias=bias, resid=y32, out32=two_step, tile_cfg=cfg)
    ops.gemm(a, w, bias=bias, resid=raw, resid_ln=(stats, lw, lb), out32=fused, tile_cfg=cfg)
    assert torch.equal(two_step, fused)
    with pytest.raises(Exception):
        ops.gemm(a, w, bias=bias, resid_ln=(stats, lw, lb), out32=fused)                    # statistics without a residual


@pytest.mark.parametrize("M,K,with_bias", [(50432, 768, True), (3000, 768, False), (2049, 64, True), (4100, 1024, True)])
def test_gemm_n64_stream_kernel(M, K, with_bias):
    """Tile cfg 16 (N = 64: W resident in LDS, A streamed from HBM into MFMA fragments) against the 256x64 tile config and torch;
    ragged M, the automatic choice, and the rejection of epilogues it does not carry."""
    from uia_hip import ops
    torch.manual_seed(21)
    a = torch.randn(M, K, device=dev()).bfloat16()
    w = (torch.randn(64, K, device=dev()) * K ** -0.5).bfloat16()
    bias = torch.randn(64, device=dev()) if with_bias else None
    y16, y4, yauto = (torch.empty(M, 64, device=dev(), dtype=torch.bfloat16) for _ in range(3))
    ops.gemm(a, w, bias=bias, out_t=y16, tile_cfg=16)
    ops.gemm(a, w, bias=bias, out_t=y4, tile_cfg=4)
    ops.gemm(a, w, bias=bias, out_t=yauto)
    ref = a.float() @ w.float().t() + (bias if with_bias else 0)
    scale = float(ref.abs().max())
    assert float((y16.float() - ref).abs().max()) < 1e-2 * scale
    assert float((y16.float() - y4.float()).abs().max()) < 1e-2 * scale            # same products, different summation order inside fp32
    assert torch.equal(yauto, y16)                                                  # M > 2048, bias + T output: the automatic choice
    assert ops.auto_tile_cfg(M, 64, K, 2, ops.EPI_BIAS | ops.EPI_OUTT) == 16
    with pytest.raises(Exception):
        ops.gemm(a, w, bias=bias, act="gelu", out_t=y16, tile_cfg=16)
