"""Config 4 timing: CLIPSeg ViT-B/16 + decoder, bs=128, synthetic 224x224 (not the headline bench; reported in DESIGN.md)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from src.models.clipseg import segmentation as S
from src.losses.dice import DiceCELoss
from uia_hip import functional as UF
from uia_hip.engine import FlatAdapterOptimizer
args = S.get_args(["--synthetic", "--batch_size", "128"])
UF.set_compute_dtype(torch.bfloat16)
model = S.prepare_model(args)
opt = FlatAdapterOptimizer([(n, p) for n, p in model.named_parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.999), max_norm=0.0)
crit = DiceCELoss()
images, labels = S.synthetic_batch(128, 224, 1, "cuda:0")
prompt = S.busi_prompt.cuda().repeat(128, 1)
def step():
    opt.zero_grad(); loss = crit(model(images, input_ids=prompt), labels); loss.backward(); opt.step(); UF.clear_t_copies(); return loss
for _ in range(3): l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"CLIPSeg bs=128 bf16: {dt*1e3:.2f} ms/step, {128/dt:.1f} images/s, loss {float(l):.4f}, decoder params {sum(p.numel() for p in model.decoder.parameters())}")
