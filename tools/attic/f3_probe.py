import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import functional as UF, ops
from src.adapters import mona as M
from src.third_party.biomedclip.model import Block
UF.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
B, N, D = 12, 197, 768
mod = M.BatchFirstMonaWrapper(M._VARIANTS["hybrid"](D, 64)).cuda().eval()
with torch.no_grad():
    for k, p in mod.named_parameters():
        if not k.endswith(("norm.weight", "gammax")): p.copy_(0.05 * torch.randn_like(p))
blk = Block(D, 12).cuda().eval()
for p in list(mod.parameters()) + list(blk.parameters()): p.requires_grad_(False)
x = torch.randn(B, N, D, device="cuda")
def run(flag):
    UF.set_fwd_resid3(flag); UF.clear_t_copies()
    with UF.linear_chain() as ch:
        ch.next_is_plain_block(True)
        y = mod(x, (14, 14))
        tok = UF._F3.get(y.data_ptr()) if flag else None
        dec = ops.three_byte_to_float(tok[0], tok[1]).float().view(B, N, D).clone() if tok else None
        hi = (tok[0] if tok else UF._ROWS[ops.raw_stream()][2])
        hi_rm = (ops.three_byte_to_float(hi, torch.zeros_like(tok[1])) if tok else None)
        sums = (tok[2] if tok else UF._ROWS[ops.raw_stream()][3]).clone()
        ch.next_is_plain_block(False)
        z = blk(y)
    torch.cuda.synchronize()
    return (y.clone() if not flag else dec), sums, z.clone()
y0, s0, z0 = run(False)
y1, s1, z1 = run(True)
print("y: decode(hi,lo) vs fp32", float((y1 - y0).abs().max() / y0.abs().max()))
print("rowsums", float((ops.rowsum_to_float(s1) - ops.rowsum_to_float(s0)).abs().max() / ops.rowsum_to_float(s0).abs().max()) if hasattr(ops, "rowsum_to_float") else "n/a")
print("block out z", float((z1 - z0).abs().max() / z0.abs().max()), "rms", float((z1 - z0).pow(2).mean().sqrt() / z0.abs().max()))
