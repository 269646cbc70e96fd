"""How long the host takes to ENQUEUE one training step (no synchronisation) against the step's GPU time: is the step launch-bound?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import functional as UF
from uia_hip.engine import FlatAdapterOptimizer, contrastive_step
from src.adapters import inject_mona_variant_to_open_clip
from src.losses import InfoNCELoss
from src.third_party.biomedclip.model import create_biomedclip
import bench
UF.set_compute_dtype(torch.bfloat16)
model = create_biomedclip(seed=0)
for p in model.parameters(): p.requires_grad_(False)
inject_mona_variant_to_open_clip(model, variant="freq_enhanced", bottleneck_dim=64)
for k, p in model.named_parameters(): p.requires_grad_("mona" in k.lower())
model = model.cuda().train()
opt = FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.95), weight_decay=0.01, max_norm=1.0)
images, ids = bench.synthetic_batch(256, 0, torch.device("cuda", 0))
crit = InfoNCELoss(0.07)
OV = "--serial" not in sys.argv      # default: the entry points' configuration (text tower on its own stream, the image tower in two slices)
for _ in range(3): contrastive_step(model, crit, opt, images, ids, overlap_text=OV)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(5):
    t0 = time.perf_counter(); contrastive_step(model, crit, opt, images, ids, overlap_text=OV); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print("enqueue ms", [round(x * 1e3, 1) for x in enq], "total ms", [round(x * 1e3, 1) for x in tot])
