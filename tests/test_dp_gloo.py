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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, loss_fn, batches = _problem()
    names = list(P)
    grads, _ = train_ref.grads_of(loss_fn, P, names, [batches[rank]])             # local micro-batch only
    flat = _flatten(grads, names)                                                  # the flat adapter-gradient buffer
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)                                    # ONE collective per step
    flat /= world
    off, g = 0, {}
    for k in names:
        n = P[k].numel()
        g[k] = flat[off:off + n].view(P[k].shape).clone()
        off += n
    params = {k: v.clone() for k, v in P.items()}
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v = {k: torch.zeros_like(v) for k, v in P.items()}
    norm = train_ref.clip_and_adamw(params, g, m, v, 1, 1e-3, (0.9, 0.95), 1e-8, 0.01, 1.0)
    torch.save({"params": params, "norm": norm}, os.path.join(out, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp2_equals_accumulation(tmp_path):
    world, port = 2, 29611 + os.getpid() % 200
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    P, loss_fn, batches = _problem()
    names = list(P)
    grads, _ = train_ref.grads_of(loss_fn, P, names, batches)                      # single process, 2 accumulation steps
    params = {k: v.clone() for k, v in P.items()}
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v = {k: torch.zeros_like(v) for k, v in P.items()}
    norm = train_ref.clip_and_adamw(params, grads, m, v, 1, 1e-3, (0.9, 0.95), 1e-8, 0.01, 1.0)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(2))
    assert abs(r0["norm"] - norm) < 1e-5 * norm
    for k in names:
        assert torch.equal(r0["params"][k], r1["params"][k])                        # replicas stay bit-identical
        assert torch.allclose(r0["params"][k], params[k], rtol=1e-5, atol=1e-7), k  # == accumulation (fp32 summation order aside)


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
