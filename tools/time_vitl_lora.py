"""Config 5 shape on one GPU: OpenAI-layout ViT-L/14 (24 blocks, width 1024, 257 tokens) + LoRA r=16 on q,k,v,o, contrastive step at
128 pairs per GPU, bf16, synthetic data, random init (SURVEY §8 row counts: 3 145 728 LoRA elements + 98 304 biases made trainable)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import functional as UF
from uia_hip.engine import FlatAdapterOptimizer, contrastive_step
from src.adapters import inject_lora_to_clip
from src.losses import InfoNCELoss
from src.third_party.openai_clip.model import CLIP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
UF.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = CLIP(768, 224, 24, 1024, 14, 77, 49408, 768, 12, 12)
for p in model.parameters(): p.requires_grad_(False)
model, n = inject_lora_to_clip(model, lora_r=16, lora_alpha=32, lora_dropout=0.1)
for k, p in model.named_parameters(): p.requires_grad_("lora" in k.lower())
model = model.cuda().train()
trainable = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
print(f"LoRA layers {n}; trainable elements {sum(p.numel() for _, p in trainable):,}")
opt = FlatAdapterOptimizer(trainable, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.01, max_norm=1.0)
g = torch.Generator().manual_seed(1)
images = torch.rand(B, 3, 224, 224, generator=g).cuda()
ids = torch.zeros(B, 77, dtype=torch.long)
for b in range(B):
    L = int(torch.randint(8, 60, (1,), generator=g)); ids[b, 0] = 49406; ids[b, 1:L] = torch.randint(1000, 40000, (L - 1,), generator=g); ids[b, L] = 49407
ids = ids.cuda()
crit = InfoNCELoss(0.07)
UF.set_dropout_seed(3)
for _ in range(2): l = contrastive_step(model, crit, opt, images, ids)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 5
for _ in range(N): l = contrastive_step(model, crit, opt, images, ids)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
print(f"ViT-L/14 + LoRA r=16, bs={B} bf16: {dt*1e3:.2f} ms/step, {B/dt:.1f} pairs/s, loss {float(l):.4f}, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")

# host side: how long the enqueue of one step takes against the step, and the top Python frames
import cProfile, pstats
enq, tot = [], []
for _ in range(4):
    t0 = time.perf_counter(); contrastive_step(model, crit, opt, images, ids); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(round((t1 - t0) * 1e3, 1)); tot.append(round((t2 - t0) * 1e3, 1))
print("enqueue ms", enq, "total ms", tot)
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3): contrastive_step(model, crit, opt, images, ids)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
