"""Census of the GEMM launches in one serial config-2 training step: shape, epilogue feature mask, launches, time.
Run on the GPU box:  python tools/gemm_census.py   (used to pick the epilogue specialisations in csrc/gemm.hip)"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import functional as UF, ops
from uia_hip.engine import FlatAdapterOptimizer, contrastive_step
from src.adapters import inject_mona_variant_to_open_clip
from src.losses import InfoNCELoss
from src.third_party.biomedclip.model import create_biomedclip
import bench

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
crit = InfoNCELoss(0.07)
images, ids = bench.synthetic_batch(256, 0, dev)
UF.set_dropout_seed(1)
for _ in range(2):
    contrastive_step(model, crit, opt, images, ids, overlap_text=False)
log = []
real = ops.gemm
def spy(a, w, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); real(a, w, **kw); e1.record()
    feats = [k for k in ("bias", "act", "dact", "aux_in", "aux_out", "resid", "resid_t", "out_t", "out32") if kw.get(k) is not None]
    if kw.get("resid_mod") or kw.get("out_group"): feats.append("rowmap")
    if kw.get("alpha", 1.0) != 1.0: feats.append("alpha")
    for k in ("resid3", "out_lo", "rowsum", "lnfold", "resid_ln"):
        if kw.get(k) is not None: feats.append(k)
    M, K = (a.rows, a.cols) if ops.is_kb(a) else a.shape
    N = getattr(w, "N", None) or w.shape[0]
    log.append(((M, N, K), "+".join(feats), e0, e1))
ops.gemm = spy
UF.ops.gemm = spy
contrastive_step(model, crit, opt, images, ids, overlap_text=False)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for shape, feats, e0, e1 in log:
    d = agg.setdefault((shape, feats), [0, 0.0]); d[0] += 1; d[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
for (shape, feats), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K = shape
    print(f"{ms:8.3f} ms  x{n:3d}  {ms/n*1e3:7.1f} us  {2.0*M*N*K*n/ms*1e-9:7.1f} TF  M={M} N={N} K={K}  {feats}")
print(f"total {tot:.2f} ms in {len(log)} launches")
