"""--method full shape on one GPU: BiomedCLIP image tower fully trainable (85.8 M parameters), frozen text tower, bs from argv (default 256)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import functional as UF
from uia_hip.engine import FlatAdapterOptimizer, contrastive_step
from src.losses import InfoNCELoss
from src.third_party.biomedclip.model import create_biomedclip
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
UF.set_compute_dtype(torch.bfloat16)
model = create_biomedclip(seed=0)
for k, p in model.named_parameters(): p.requires_grad_(k.startswith("visual."))
model = model.cuda().train()
tr = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
print(f"trainable {sum(p.numel() for _, p in tr):,}")
opt = FlatAdapterOptimizer(tr, lr=1e-6, betas=(0.9, 0.95), weight_decay=0.01, max_norm=1.0)
images, ids = bench.synthetic_batch(B, 0, torch.device("cuda", 0))
crit = InfoNCELoss(0.07)
for _ in range(2): l = contrastive_step(model, crit, opt, images, ids)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): l = contrastive_step(model, crit, opt, images, ids)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
print(f"full fine-tune (image tower), bs={B} bf16: {dt*1e3:.1f} ms/step, {B/dt:.1f} pairs/s, loss {float(l):.4f}, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
