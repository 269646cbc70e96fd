"""Where do the entry point's extra milliseconds per update come from?  The bench's model and batch, the entry point's loop, different feeders."""
import os, sys, time, contextlib, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import functional as UF, ops, engine
from src.adapters import inject_mona_variant_to_open_clip
from src.losses import InfoNCELoss
from src.third_party.biomedclip.model import create_biomedclip
sys.argv = ["bench.py"]
import bench
torch.set_num_threads(4)
dev = torch.device("cuda:0")
model = create_biomedclip(seed=0)
for p in model.parameters(): p.requires_grad_(False)
with contextlib.redirect_stdout(sys.stderr):
    inject_mona_variant_to_open_clip(model, variant="freq_enhanced", bottleneck_dim=64)
for k, p in model.named_parameters(): p.requires_grad_("mona" in k.lower())
model = model.to(dev).train()
opt = engine.FlatAdapterOptimizer([(k, p) for k, p in model.named_parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.95), weight_decay=0.01, max_norm=1.0)
crit = InfoNCELoss(0.07)
images, ids = bench.synthetic_batch(256, 0, dev)
UF.set_dropout_seed(1)
N = 30

def timed(name, feeder):
    loop = engine.ContrastiveLoop(model, crit, opt, accumulation_steps=1, lr=1e-4, lr_min=1e-8, total_updates=1000)
    for rep in range(2):
        loop.begin_epoch(N)
        it = iter(feeder())
        torch.cuda.synchronize()
        t0 = time.perf_counter(); enq = 0.0
        for i, (im, tk, ready, kw) in enumerate(it):
            t1 = time.perf_counter()
            loop.micro(im, tk, i, ready=ready, **kw)
            enq += time.perf_counter() - t1
        loop.end_epoch()
        dt = time.perf_counter() - t0
    print(f"{name:44s} {dt / N * 1e3:7.2f} ms/update   enqueue {enq / N * 1e3:6.2f}", flush=True)

def resident():
    for _ in range(N): yield images, ids, None, dict(inputs_ready=True)
def resident_wait():
    for _ in range(N): yield images, ids, None, dict(inputs_ready=False)
def resident_event():
    for _ in range(N):
        ev = torch.cuda.Event(); ev.record()
        yield images, ids, ev, {}
timed("resident batch, inputs_ready (= bench)", resident)
timed("resident batch, text waits for caller", resident_wait)
timed("resident batch + a ready event per step", resident_event)

h_im, h_id = images.cpu(), ids.cpu()
class ListLoader:
    def __len__(self): return N
    def __iter__(self):
        for _ in range(N): yield h_im, ["x"] * 256, h_id
def pf(depth=2):
    p = engine.DevicePrefetcher(ListLoader(), None, dev, depth=depth)
    return lambda: ((a, b, c, {}) for a, b, c in p)
timed("prefetcher: pin copy + H2D (host batch)", pf())
# same, but the H2D copies replaced by nothing (device slots stay as they are): thread + pin copy only
orig = torch.Tensor.copy_
class NoH2D(engine.DevicePrefetcher):
    def _make_slots(self, images, ids):
        super()._make_slots(images, ids)
        for sl in self._slots:
            sl["d_im"].copy_(images); sl["d_id"].copy_(ids)
            sl["d_im"] = _Frozen(sl["d_im"]); sl["d_id"] = _Frozen(sl["d_id"])
class _Frozen:
    def __init__(self, t): self.t = t
    def copy_(self, *a, **k): return self
def pf_noh2d():
    p = NoH2D(ListLoader(), None, dev)
    return lambda: ((a.t, b.t, c, {}) for a, b, c in p)
timed("prefetcher: pin copy only (no H2D)", pf_noh2d())
# a busy python thread beside the main thread (GIL pressure), resident batch
stop = False
def spin():
    x = 0
    while not stop:
        for _ in range(1000): x += 1
        time.sleep(0.0005)
th = threading.Thread(target=spin, daemon=True); th.start()
timed("resident batch + a python thread spinning", resident)
stop = True
