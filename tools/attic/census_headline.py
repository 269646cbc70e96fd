"""Which launches of the headline step are NOT this library's kernels, and which Python lines issue them?  torch profiler with stacks over one step of the
bench.py default configuration (BiomedCLIP ViT-B/16 + Mona, 256 pairs), grouped by kernel name and by call site.  GPU box: python tools/census_headline.py [batch]"""
import collections
import contextlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch  # noqa: E402
from uia_hip import functional as UF  # noqa: E402
from uia_hip.engine import FlatAdapterOptimizer, contrastive_step, init_data_parallel  # noqa: E402
from src.adapters import inject_mona_variant_to_open_clip  # noqa: E402
from src.losses import InfoNCELoss  # noqa: E402
from src.third_party.biomedclip.model import create_biomedclip  # noqa: E402

import bench  # noqa: E402

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
UF.set_dropout_seed(1234)
for _ in range(3):
    contrastive_step(model, crit, opt, images, ids, overlap_text=True)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    contrastive_step(model, crit, opt, images, ids, overlap_text=True)
    torch.cuda.synchronize()
kern = collections.Counter()
ktime = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern[e.name[:90]] += 1
        ktime[e.name[:90]] += e.device_time if hasattr(e, "device_time") else e.cuda_time
print("device launches of one step that are not uia kernels:")
for k, n in kern.most_common():
    if "anonymous namespace" in k or "_GLOBAL__N_" in k or "uia" in k.lower():
        continue
    print(f"  {n:5d} x  {ktime[k] / max(n, 1):7.1f} us   {k}")
print("call sites (aten ops that launch them):")
rows = prof.key_averages(group_by_stack_n=10)
for r in sorted(rows, key=lambda r: -r.count):
    if r.key in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::cat", "aten::add_", "aten::add", "aten::mul", "aten::mul_", "aten::div", "aten::sum", "aten::_to_copy",
                 "aten::index", "aten::index_select", "aten::clone", "aten::sub", "aten::where", "aten::eq", "aten::ne", "aten::cumsum", "aten::arange", "aten::gather") and r.count >= 2:
        st = [s for s in r.stack if "site-packages/torch" not in s and "<built-in" not in s][:4]
        print(f"  {r.count:4d}  {r.key:16s} " + "  <-  ".join(s.replace(ROOT + "/", "")[:90] for s in st))
