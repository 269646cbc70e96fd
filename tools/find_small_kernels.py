"""Which Python call sites launch the small copy / cast / fill kernels of one training step?  Runs the bench model for a few steps under
torch.profiler (with stacks) and prints, per GPU kernel name matching --match, the op and the first frames of this repository that
launched it.  GPU box: python tools/find_small_kernels.py [--match copy,cast,fill,cat]"""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from torch.profiler import profile, ProfilerActivity

ap = argparse.ArgumentParser()
ap.add_argument("--match", default="copy,Copy,cast,fill,Fill,cat,Cat,elementwise")
ap.add_argument("--batch", type=int, default=64)
args = ap.parse_args()
import bench
from uia_hip import functional as UF
from uia_hip.engine import FlatAdapterOptimizer, contrastive_step, init_data_parallel
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
opt = FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.95), weight_decay=0.01, max_norm=1.0)
init_data_parallel(opt)
crit = InfoNCELoss(0.07)
images, ids = bench.synthetic_batch(args.batch, 0, dev)
UF.set_dropout_seed(1)
for _ in range(2):
    contrastive_step(model, crit, opt, images, ids)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    contrastive_step(model, crit, opt, images, ids)
    torch.cuda.synchronize()
pats = args.match.split(",")
ev = prof.events()
by = collections.Counter()
dur = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        for k in e.kernels:
            if any(p in k.name for p in pats):
                frames = [f for f in (e.stack or []) if "/nextgen-uia_amd/" in f or "bench.py" in f][:3]
                key = (k.name[:60], e.name, " <- ".join(f.split("/nextgen-uia_amd/")[-1] for f in frames))
                by[key] += 1
                dur[key] += k.duration
# device-to-device memcpys are not kernels in the profiler's view: count them by the CPU op that is running when they are issued
cpu_ops = sorted([e for e in ev if e.device_type == torch.autograd.DeviceType.CPU], key=lambda e: e.time_range.start)
mem = collections.Counter()
for e in ev:
    if "Memcpy" in e.name or "memcpy" in e.name:
        t = e.time_range.start
        # innermost-to-outermost CPU ops that contain the launch time of the runtime call with the same correlation id
        mem[e.name] += 1
print("memcpy events:", dict(mem))
rt = [e for e in ev if e.name in ("hipMemcpyAsync", "hipMemcpyWithStream", "hipMemcpyDtoDAsync")]
parents = collections.Counter()
for r in rt:
    t = r.time_range.start
    inside = [e.name for e in cpu_ops if e.time_range.start <= t <= e.time_range.end and e.name != r.name]
    parents[" > ".join(inside[:4])] += 1
for k, n in parents.most_common(20):
    print(f"{n:4d} x  {k}")
for key, n in sorted(by.items(), key=lambda kv: -dur[kv[0]]):
    print(f"{n:4d} x {dur[key] / max(n, 1):7.1f} us  {key[0]:60s} {key[1]:24s} {key[2]}")
names = collections.Counter(e.name[:70] for e in ev if any(s in e.name for s in ("opy", "emcpy", "emset", "hip")))
for k, n in names.most_common(25):
    print(f"{n:5d}  {k}")
