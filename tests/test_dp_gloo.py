"""CPU, world_size = 2 over gloo: the data-parallel claim of SURVEY §8(e) / DESIGN.md —

    R ranks, each back-propagating its LOCAL mean InfoNCE loss, gradients summed by ONE all-reduce and scaled by 1/R,
    then identical clip+AdamW on every rank
        ==  the reference's single-process gradient accumulation over the same R micro-batches
            (/root/reference/src/models/biomedclip/finetune.py:287-302).

The arithmetic here is the oracle's (CPU); what is under test is the exchange protocol the HIP engine implements with
RCCL (uia_hip/engine.py: FlatAdapterOptimizer.all_reduce + step(grad_scale=1/world)), including the flat-buffer layout."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mona_ref, train_ref, losses_ref


def _problem(seed=0):
    g = torch.Generator().manual_seed(seed)
    D, bott, hw, B = 32, 8, (3, 3), 4
    P = mona_ref.reference_init("hybrid", D, bott, g)
    P["gamma"] = 0.3 * torch.randn(D, generator=g)
    head = torch.randn(16, D, generator=g) * 0.2
    batches = [(torch.randn(B, 1 + hw[0] * hw[1], D, generator=g), torch.randn(B, 16, generator=g)) for _ in range(2)]

    def loss_fn(Pq, x, txt):
        y = mona_ref.forward(x, Pq, "hybrid", hw)
        return losses_ref.info_nce(y[:, 0] @ head.T, txt, 0.07)
    return P, loss_fn, batches


def _flatten(d, names):
    return torch.cat([d[k].reshape(-1) for k in names])


def _worker(rank, world, port, out):
    """One rank of the exchange, on the ENGINE's flat-buffer code (uia_hip.engine.FlatLayout: offsets, 16-byte padding, .grad views,
    dp_grad_scale) with gloo standing in for the RCCL all-reduce and the oracle's clip+AdamW for the fused HIP update."""
    from uia_hip.engine import FlatLayout, dp_grad_scale, all_ranks_agree, sum_over_ranks
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, loss_fn, batches = _problem()
    names = list(P)
    params = [(k, torch.nn.Parameter(P[k].clone())) for k in names]
    lay = FlatLayout(params)                                                       # parameters / gradients now live in the flat buffers
    assert lay.grad_views_intact() and lay.numel % 4 == 0 and all(o % 4 == 0 for o in lay.offsets)
    if rank == 1:                                                                  # replicas must start identical: rank 0's weights win
        with torch.no_grad():
            lay.p.add_(1.0)
    dist.broadcast(lay.p, src=0)
    Pq = {k: p for k, p in params}
    loss = loss_fn(Pq, *batches[rank])                                             # local micro-batch only
    ok = all_ranks_agree(bool(torch.isfinite(loss)))                               # the entry points' collective skip decision
    loss.backward()                                                                # autograd accumulates INTO the flat views
    assert lay.grad_views_intact()
    dist.all_reduce(lay.g, op=dist.ReduceOp.SUM)                                   # ONE collective per step over the flat buffer
    g = {k: v.clone() * dp_grad_scale(world) for k, v in lay.unflatten(lay.g).items()}
    pd = {k: v.detach().clone() for k, v in lay.unflatten(lay.p).items()}
    m = {k: torch.zeros_like(v) for k, v in pd.items()}
    v = {k: torch.zeros_like(v) for k, v in pd.items()}
    norm = train_ref.clip_and_adamw(pd, g, m, v, 1, 1e-3, (0.9, 0.95), 1e-8, 0.01, 1.0)
    tot = sum_over_ranks(float(loss.detach()), 1.0)
    # a rank-local "non-finite" must become a global skip, identically on both ranks
    skip = not all_ranks_agree(rank != 1)
    torch.save({"params": pd, "norm": norm, "ok": ok, "skip": skip, "tot": tot, "numel": lay.numel, "offsets": lay.offsets},
               os.path.join(out, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp2_equals_accumulation(tmp_path):
    world, port = 2, 29611 + os.getpid() % 200
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    P, loss_fn, batches = _problem()
    names = list(P)
    grads, mean_loss = train_ref.grads_of(loss_fn, P, names, batches)              # single process, 2 accumulation steps
    params = {k: v.clone() for k, v in P.items()}
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v = {k: torch.zeros_like(v) for k, v in P.items()}
    norm = train_ref.clip_and_adamw(params, grads, m, v, 1, 1e-3, (0.9, 0.95), 1e-8, 0.01, 1.0)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(2))
    assert abs(r0["norm"] - norm) < 1e-5 * norm
    assert r0["ok"] and r1["ok"] and r0["skip"] and r1["skip"]                      # agreements are global
    assert r0["tot"] == r1["tot"] and abs(r0["tot"][0] / r0["tot"][1] - mean_loss) < 1e-5
    assert r0["numel"] >= sum(v.numel() for v in P.values()) and r0["offsets"] == r1["offsets"]
    for k in names:
        assert torch.equal(r0["params"][k], r1["params"][k])                        # replicas stay bit-identical
        assert torch.allclose(r0["params"][k], params[k], rtol=1e-5, atol=1e-7), k  # == accumulation (fp32 summation order aside)


def test_rank_shard_sampler_partitions_the_epoch():
    """DistributedSampler semantics of src/datasets/finetune.RankShardSampler: equal shard sizes (collectives stay aligned),
    disjoint shards that together cover the epoch's permutation, a new permutation per epoch, identical across ranks."""
    from src.datasets.finetune import RankShardSampler
    for n, world in ((10, 2), (11, 2), (37, 8), (8, 8)):
        shards = [RankShardSampler(n, r, world, shuffle=True, seed=5) for r in range(world)]
        idx = [s.indices() for s in shards]
        assert len({len(i) for i in idx}) == 1 and len(idx[0]) == len(shards[0]) == -(-n // world)
        flat = sorted(j for i in idx for j in i)
        assert set(flat) == set(range(n)) and len(flat) - n < world                 # everything seen; only the wrap-around padding repeats
        for s in shards:
            s.set_epoch(1)
        idx1 = [s.indices() for s in shards]
        assert idx1 != idx and sorted(set(j for i in idx1 for j in i)) == list(range(n))
    a, b = RankShardSampler(16, 0, 2, shuffle=False), RankShardSampler(16, 1, 2, shuffle=False)
    assert a.indices() == list(range(0, 16, 2)) and b.indices() == list(range(1, 16, 2))


def test_flat_layout_survives_set_to_none(tmp_path):
    """zero_grad(set_to_none=True)-style callers orphan the .grad views; the layout notices and re-binds."""
    from uia_hip.engine import FlatLayout
    ps = [("a", torch.nn.Parameter(torch.randn(5))), ("b", torch.nn.Parameter(torch.randn(3, 3)))]
    lay = FlatLayout(ps)
    assert lay.offsets == [0, 8] and lay.numel == 20 and lay.grad_views_intact()
    ps[0][1].grad = None
    assert not lay.grad_views_intact()
    lay.rebind_grads()
    assert lay.grad_views_intact()
    (ps[0][1].sum() * 2 + ps[1][1].sum()).backward()
    assert torch.equal(lay.g[:5], torch.full((5,), 2.0)) and torch.equal(lay.g[8:17], torch.ones(9)) and float(lay.g[5:8].abs().sum()) == 0


# ---- opt-in global-batch loss: all-gather the features, every rank evaluates the same global InfoNCE and
#      back-propagates only the rows it owns; the all-reduced parameter gradient (NOT divided by world) is the gradient of that loss.
class _Gather(torch.autograd.Function):            # the protocol of uia_hip.engine.GatherFeaturesFn, with gloo in place of RCCL
    @staticmethod
    def forward(ctx, x, rank, world):
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x.contiguous())
        ctx.meta = (rank, x.shape[0])
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, g):
        rank, B = ctx.meta
        return g[rank * B:(rank + 1) * B].contiguous(), None, None


def _features(Pq, x, head, hw=(3, 3)):
    return mona_ref.forward(x, Pq, "hybrid", hw)[:, 0] @ head.T


def _global_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, _, batches = _problem()
    head = torch.randn(16, 32, generator=torch.Generator().manual_seed(99)) * 0.2
    Pq = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    x, txt = batches[rank]
    fi = _Gather.apply(_features(Pq, x, head), rank, world)
    ft = _Gather.apply(txt, rank, world)
    loss = losses_ref.info_nce(fi, ft, 0.07)
    loss.backward()
    flat = _flatten({k: v.grad for k, v in Pq.items()}, list(P))
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)                                    # no 1/world: the ranks' contributions add up
    torch.save({"flat": flat, "loss": loss.detach()}, os.path.join(out, f"g{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp2_global_batch_loss_protocol(tmp_path):
    world, port = 2, 29811 + os.getpid() % 200
    mp.spawn(_global_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    P, _, batches = _problem()
    head = torch.randn(16, 32, generator=torch.Generator().manual_seed(99)) * 0.2
    Pq = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    fi = torch.cat([_features(Pq, x, head) for x, _ in batches], 0)                 # one process, the concatenated batch
    ft = torch.cat([t for _, t in batches], 0)
    loss = losses_ref.info_nce(fi, ft, 0.07)
    loss.backward()
    want = _flatten({k: v.grad for k, v in Pq.items()}, list(P))
    r0, r1 = (torch.load(os.path.join(tmp_path, f"g{r}.pt")) for r in range(2))
    assert torch.equal(r0["flat"], r1["flat"]) and abs(float(r0["loss"]) - float(loss)) < 1e-6
    assert torch.allclose(r0["flat"], want, rtol=1e-5, atol=1e-7)


def test_dist_env_parsing(monkeypatch):
    from uia_hip.engine import dist_env
    monkeypatch.setenv("RANK", "3"); monkeypatch.setenv("LOCAL_RANK", "1"); monkeypatch.setenv("WORLD_SIZE", "8")
    assert dist_env() == (3, 1, 8)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k)
    assert dist_env() == (0, 0, 1)


# ---- round 5: the device-guarded loop's exchange (engine.FlatAdapterOptimizer.accumulate / update): the non-finite flag of the boundary micro-batch travels in
#      acc[n] through the SAME all-reduce as the gradients; a skipped update leaves sum/world on every rank so that the next all-reduce restores the sum.
def _accum_guarded(lay, loss):
    """torch restatement of uia_grad_accum_guarded (csrc/optim.hip) on the engine's own FlatLayout buffers."""
    ok = bool(torch.isfinite(loss))
    n = lay.numel
    if ok:
        lay.acc[:n] += lay.g
    lay.g.zero_()
    lay.acc[n] = 0.0 if ok else 1.0
    return ok


def _guarded_worker(rank, world, port, out):
    from uia_hip.engine import FlatLayout, dp_grad_scale
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    w = torch.nn.Parameter(torch.randn(37, generator=g))
    lay = FlatLayout([("w", w)])
    n = lay.numel
    assert lay.acc.numel() == n + FlatLayout.ALIGN
    grads = [[torch.randn(37, generator=g) for _ in range(2)] for _ in range(2)]           # [cycle][rank]
    losses = [[1.0, float("nan")], [0.5, 0.7]]                                              # cycle 0: rank 1's boundary micro-batch is non-finite
    decisions, sums = [], []
    for cycle in range(2):
        w.grad.copy_(grads[cycle][rank])                                                   # this rank's backward into the staging views
        _accum_guarded(lay, torch.tensor(losses[cycle][rank]))
        dist.all_reduce(lay.acc, op=dist.ReduceOp.SUM)                                     # ONE collective: gradients and flag together
        run = float(lay.acc[n]) == 0.0
        decisions.append(run)
        if run:
            sums.append(lay.acc[:n].clone() * dp_grad_scale(world))
            lay.acc.zero_()
        else:
            lay.acc[:n] *= 1.0 / world                                                     # uia_adamw_clip_step_guarded's skip path
    torch.save({"decisions": decisions, "sums": sums, "want": (grads[0][0] + grads[1][0] + grads[1][1]) * dp_grad_scale(world), "pad": n - 37}, os.path.join(out, f"g{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp2_guarded_skip_is_collective_and_keeps_the_accumulated_sum(tmp_path):
    world, port = 2, 29811 + os.getpid() % 200
    mp.spawn(_guarded_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"g{k}.pt")) for k in range(world)]
    for k in range(world):
        assert r[k]["decisions"] == [False, True]                          # both ranks skip the first update although only rank 1 saw the NaN; both run the second
        got, want = r[k]["sums"][0], r[k]["want"]
        assert torch.allclose(got[:37], want, rtol=1e-6, atol=1e-7)        # rank 0's finite micro-batch of the skipped cycle is still in the sum, rank 1's NaN one is not
    assert torch.equal(r[0]["sums"][0], r[1]["sums"][0])
