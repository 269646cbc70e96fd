"""Child program of tests/test_round6_gpu.py::test_two_ranks_on_two_gpus (not collected by pytest): run under

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tests/dp2_gpu_driver.py OUT.json

one rank per GPU.  Every rank runs two optimiser updates of the contrastive step on ITS batch through the library's own RCCL all-reduce
(engine.init_data_parallel -> uia_comm_init(rank, N) -> uia_allreduce_sum); rank 0 then repeats the two updates ALONE with accumulation over the N ranks'
batches (the reference's --accumulation_steps N, finetune.py:287-302) and the job reports

    world as RCCL saw it, max |p_rank - p_rank0| over ranks (must be 0), ||p_dp - p_accumulation|| / ||p - p_0|| (rounding of a different summation order).

N = 1 rehearses the whole path on a one-GPU box (DP(1) = accumulation(1))."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

TOY = dict(embed_dim=128, vision_cfg=dict(img_size=32, patch_size=8, embed_dim=128, depth=2, num_heads=2),
           text_cfg=dict(vocab_size=30000, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, max_position_embeddings=64))


def build(dev):
    import torch
    from src.adapters import inject_mona_variant_to_open_clip
    from src.third_party.biomedclip.model import create_biomedclip
    torch.manual_seed(17)                                    # the injector initialises the adapters from the global generator
    model = create_biomedclip(config=TOY, seed=5)
    for p in model.parameters():
        p.requires_grad_(False)
    inject_mona_variant_to_open_clip(model, variant="hybrid", bottleneck_dim=64)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "mona" in k and p.dim() >= 2:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    for k, p in model.named_parameters():
        p.requires_grad_("mona" in k)
    return model.to(dev).eval()


def batch(rank, step, B=16):
    import torch
    g = torch.Generator().manual_seed(1000 * step + rank)
    images = torch.rand(B, 3, 32, 32, generator=g)
    ids = torch.zeros(B, 64, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(6, 40, (1,), generator=g))
        ids[b, 1:n - 1] = torch.randint(1000, 30000, (n - 2,), generator=g)
        ids[b, 0], ids[b, n - 1] = 2, 3
    return images, ids


def main():
    import torch
    import torch.distributed as dist
    from uia_hip import functional as UF
    from uia_hip import ops
    from uia_hip.engine import FlatAdapterOptimizer, bind_device, contrastive_step, init_data_parallel
    from src.losses import InfoNCELoss
    rank, local, world = bind_device()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    UF.set_compute_dtype(torch.float32)                      # fp32 operands: DP and accumulation differ by summation order only
    crit = InfoNCELoss(0.07)
    model = build(dev)
    opt = FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-2, max_norm=1.0)
    p0 = opt.p.clone()
    init_data_parallel(opt, force_comm=True)
    rccl_world = ops.comm_world()
    for step in range(2):
        im, ids = batch(rank, step)
        contrastive_step(model, crit, opt, im.to(dev), ids.to(dev), lr=1e-2)
    guard = opt.read_guard()
    mine = opt.p.detach().clone()
    if world > 1:
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
    else:
        gathered = [mine]
    out = None
    if rank == 0:
        rank_spread = max(float((g - gathered[0]).abs().max()) for g in gathered)
        # the same two updates in ONE process: accumulation over the ranks' batches (micro_batches = world), no collective
        ref_model = build(dev)
        ref = FlatAdapterOptimizer([(k, p) for k, p in ref_model.named_parameters() if p.requires_grad], lr=1e-2, max_norm=1.0)
        for step in range(2):
            parts = [batch(r, step) for r in range(world)]
            im = torch.cat([p[0] for p in parts]).to(dev)
            ids = torch.cat([p[1] for p in parts]).to(dev)
            contrastive_step(ref_model, crit, ref, im, ids, micro_batches=world, lr=1e-2)
        moved = float((ref.p - p0).norm())
        # L2 over the whole adapter: AdamW divides by sqrt(v), so an element whose gradient is at the rounding level of the float atomics that sum the weight
        # gradients can move by a whole lr in either direction — a max-norm over 1e5 elements would measure that, not the collective
        out = {"env_world": world, "rccl_world": int(rccl_world), "updates": guard["updates"], "skipped": guard["skipped"], "rank_spread": rank_spread,
               "dp_vs_accumulation": float((mine - ref.p).norm()) / max(moved, 1e-30), "moved": moved}
    if world > 1:
        dist.barrier()
    ops.comm_destroy()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f)


if __name__ == "__main__":
    main()
