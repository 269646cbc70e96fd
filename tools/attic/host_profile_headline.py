"""Host-side cost of the headline step (bench.py defaults): enqueue time against total time, and where the Python time goes (cProfile, top 30 by own time).
GPU box: python tools/host_profile_headline.py [batch]"""
import contextlib, cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import functional as UF
from uia_hip.engine import FlatAdapterOptimizer, contrastive_step, init_data_parallel
from src.adapters import inject_mona_variant_to_open_clip
from src.losses import InfoNCELoss
from src.third_party.biomedclip.model import create_biomedclip
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
UF.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
model = create_biomedclip(seed=0)
for p in model.parameters():
    p.requires_grad_(False)
with contextlib.redirect_stdout(sys.stderr):
    inject_mona_variant_to_open_clip(model, variant="freq_enhanced", bottleneck_dim=64)
for k, p in model.named_parameters():
    p.requires_grad_("mona" in k.lower())
model = model.to(dev).train()
opt = FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.95), weight_decay=0.01, max_norm=1.0)
init_data_parallel(opt)
crit = InfoNCELoss(0.07)
images, ids = bench.synthetic_batch(B, 0, dev)
step = lambda: contrastive_step(model, crit, opt, images, ids, inputs_ready=True)
for _ in range(3): step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(5):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(round((t1 - t0) * 1e3, 2)); tot.append(round((t2 - t0) * 1e3, 2))
print("batch", B, "enqueue ms", enq, "total ms", tot)
pr = cProfile.Profile()
pr.enable()
for _ in range(5): step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
