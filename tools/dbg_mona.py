import sys, torch
sys.path[:0]=["/root/repo","/root/repo/nextgen-uia_amd","/root/repo/tests"]
import test_parity_gpu as T
from oracle import mona_ref
from uia_hip import functional as UF
from src.adapters import mona as M
UF.set_compute_dtype(torch.bfloat16)
order = sys.argv[1].split(",")
for variant in order:
    g = torch.Generator().manual_seed(11)
    B, D, hw = 3, 128, (14,14)
    N = 1 + hw[0]*hw[1]
    mod = M._VARIANTS[variant](D, 64)
    T.randomize(mod, g)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in mod.named_parameters()}
    x = torch.randn(B, N, D, generator=g) * 1.5
    dy = torch.randn(B, N, D, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = mona_ref.forward(xr, P, variant, hw, keep_mask=None, p_drop=0.1)
    yr.backward(dy)
    mod = mod.to("cuda"); mod.train(False); mod.keep_mask=None
    xg = x.to("cuda").requires_grad_(True)
    y = mod(xg.permute(1,0,2), hw).permute(1,0,2)
    y.backward(dy.to("cuda"))
    worst = max((T.rel(p.grad, P[k].grad), k) for k,p in mod.named_parameters())
    print(variant, "y", round(T.rel(y, yr),5), "dx", round(T.rel(xg.grad, xr.grad),5), "worst param grad", worst)
