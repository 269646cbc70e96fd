"""Where do the device-to-device copies / fills of one training step come from?  (torch.profiler, python stacks)"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from uia_hip import functional as UF
from uia_hip.engine import FlatAdapterOptimizer, contrastive_step
from src.adapters import inject_mona_variant_to_open_clip
from src.losses import InfoNCELoss
from src.third_party.biomedclip.model import create_biomedclip

dev = torch.device("cuda", 0)
UF.set_compute_dtype(torch.bfloat16)
model = create_biomedclip(seed=0)
for p in model.parameters():
    p.requires_grad_(False)
inject_mona_variant_to_open_clip(model, variant="freq_enhanced", bottleneck_dim=64)
for k, p in model.named_parameters():
    p.requires_grad_("mona" in k.lower())
model = model.to(dev).train()
opt = FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-4)
crit = InfoNCELoss(0.07)
images, ids = bench.synthetic_batch(256, 0, dev)
for _ in range(2):
    contrastive_step(model, crit, opt, images, ids, overlap_text=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    contrastive_step(model, crit, opt, images, ids, overlap_text=False)
    torch.cuda.synchronize()
agg = collections.Counter()
tim = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::contiguous", "aten::clone", "aten::to", "aten::_to_copy", "aten::zeros", "aten::zeros_like", "aten::mul", "aten::add", "aten::add_", "aten::div"):
        st = [s for s in (e.stack or []) if "nextgen-uia_amd" in s or "bench.py" in s or "engine.py" in s]
        key = (e.name, st[0] if st else "?", str(e.input_shapes)[:60])
        agg[key] += 1
        tim[key] += e.device_time_total
for k, n in sorted(agg.items(), key=lambda kv: -tim[kv[0]])[:40]:
    print(f"{n:4d}x  {tim[k]:8.0f} us  {k[0]:18s} {k[2]:60s} {k[1]}")
