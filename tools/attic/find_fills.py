"""Where do the small ATen fill launches of a ViT-L/14 + LoRA step come from?  torch profiler with stacks, grouped by Python call site.  Run on the GPU box."""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import functional as UF
from uia_hip.engine import FlatAdapterOptimizer, contrastive_step, init_data_parallel
from src.adapters import inject_lora_to_clip
from src.losses import InfoNCELoss
from src.third_party.openai_clip.model import CLIP

UF.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = CLIP(768, 224, 4, 1024, 14, 77, 49408, 768, 12, 2)
for p in model.parameters():
    p.requires_grad_(False)
model, n = inject_lora_to_clip(model, lora_r=16, lora_alpha=32, lora_dropout=0.1)
for k, p in model.named_parameters():
    p.requires_grad_("lora" in k.lower())
model = model.to(dev).train()
opt = FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-4)
init_data_parallel(opt)
B = 32
images = torch.rand(B, 3, 224, 224).to(dev)
ids = torch.zeros(B, 77, dtype=torch.long)
ids[:, 0], ids[:, 1:9], ids[:, 9] = 49406, 1000, 49407
ids = ids.to(dev)
crit = InfoNCELoss(0.07)
for _ in range(2):
    contrastive_step(model, crit, opt, images, ids, overlap_text=False)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    contrastive_step(model, crit, opt, images, ids, overlap_text=False)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_stack_n=8)
for r in sorted(rows, key=lambda r: -r.count):
    if r.key in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::copy_", "aten::cat", "aten::add_", "aten::add", "aten::mul", "aten::contiguous", "aten::to", "aten::_to_copy") and r.count >= 4:
        site = [s.split("/repo/")[-1] for s in r.stack if "/repo/" in s][:3]
        print(f"{r.count:5d}  {r.key:14s} {' <- '.join(site)[:230]}")
