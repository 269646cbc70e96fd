import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/nextgen-uia_amd'); sys.path.insert(0, '/root/repo/tests')
import test_parity_gpu as T
from uia_hip import functional as UF
from src.adapters import inject_mona_variant_to_open_clip
from src.losses import InfoNCELoss
from src.third_party.biomedclip.model import create_biomedclip
from oracle import train_ref, vit_ref
for mode in ("fp32", "bf16"):
    variant = "hybrid"
    UF.set_compute_dtype(T.DT[mode])
    g = torch.Generator().manual_seed(3)
    model = create_biomedclip(config=T.TOY, seed=1)
    T.randomize(model, g, 0.08)
    for p in model.parameters(): p.requires_grad_(False)
    inject_mona_variant_to_open_clip(model, variant=variant, bottleneck_dim=64)
    T.randomize(torch.nn.ModuleList([b.mona for b in model.visual.trunk.blocks]), g, 0.15)
    for k, p in model.named_parameters(): p.requires_grad_("mona" in k)
    model.eval()
    images, ids = T.toy_batch(g)
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    trainable = [k for k in P if "mona" in k]
    mona = dict(variant=variant, hw=(4, 4))
    gref, lref = train_ref.grads_of(lambda Pq, im, tk: train_ref.biomedclip_loss(Pq, im, tk, mona=mona, heads=2, text_heads=2), P, trainable, [(images, ids)])
    model = model.cuda()
    fi = model.encode_image(images.cuda()); ft = model.encode_text(ids.cuda())
    loss = InfoNCELoss(0.07)(fi, ft); loss.backward()
    gmax = max(float(v.abs().max()) for v in gref.values())
    print(mode, "loss", float(loss), lref, "global max grad", gmax)
    for k, p in model.named_parameters():
        if "mona" in k:
            r = T.rel(p.grad, gref[k])
            if r > 0.02: print(f"  {k:70s} rel {r:.3e}  max|ref| {float(gref[k].abs().max()):.3e}  abs err {float((p.grad.cpu()-gref[k]).abs().max()):.3e}")
