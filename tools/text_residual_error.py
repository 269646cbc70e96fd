"""Full-depth check of the text tower's bf16-mode accuracy: fp32 mode as the reference, bf16 with the fp32 residual, bf16 with the
T residual (post_ln_layer).  Run on the GPU box: python tools/text_residual_error.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import functional as UF
from src.third_party.biomedclip.model import create_biomedclip
import bench
model = create_biomedclip(seed=0).cuda().eval()
for p in model.parameters(): p.requires_grad_(False)
_, ids = bench.synthetic_batch(64, 0, torch.device("cuda", 0))
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
UF.set_compute_dtype(torch.float32); ref = model.encode_text(ids).clone()
UF.set_compute_dtype(torch.bfloat16)
for flag in (False, True):
    UF._STATE["text_resid_t"] = flag
    out = model.encode_text(ids)
    cos = torch.nn.functional.cosine_similarity(out, ref, dim=1).min()
    print(f"T residual={flag}: max rel err of the 512-d text features vs fp32 mode {rel(out, ref):.4e}, min cosine {float(cos):.6f}")
