import sys, torch, numpy as np
sys.path[:0]=["/root/repo","/root/repo/nextgen-uia_amd","/root/repo/tests"]
import test_parity_gpu as T
from oracle import fpn_ref
from uia_hip import functional as UF
g = {k: torch.from_numpy(v) for k, v in np.load("/root/repo/tests/golden/fpn_adapter.npz").items()}
mode, task = sys.argv[1], sys.argv[2]
UF.set_compute_dtype(T.DT[mode])
ad, P = T._fpn_model(task, False)
A = {k[2:]: v.clone() for k, v in g.items() if k.startswith("A.")}
sd = ad.state_dict(); sd.update(A); ad.load_state_dict(sd); ad.eval(); ad.freeze_clip_backbone()
images, dy = g["images"], g[f"{task}.dy"]
Pq = {k: v.detach().clone() for k, v in ad.clip_model.state_dict().items()}
Aq = {k: v.clone().requires_grad_(True) for k, v in A.items()}
ref = fpn_ref.adapter_forward(images, Pq, Aq, task=task); (ref*dy).sum().backward()
ad = ad.to("cuda"); out = ad(images.cuda()); (out*dy.cuda()).sum().backward()
print("out rel", T.rel(out, ref), "ref absmax", float(ref.abs().max()))
for k,p in ad.named_parameters():
    if k in Aq and p.grad is not None:
        a, b = p.grad.float().cpu().flatten(), Aq[k].grad.flatten()
        print(f"{k:28s} cos {float(torch.dot(a,b)/(a.norm()*b.norm()+1e-30)):.5f} relL2 {float((a-b).norm()/(b.norm()+1e-30)):.4f} |ref| {float(b.norm()):.3e}")
