"""Host-side cost of one CLIPSeg training step (BASELINE configs[3]): enqueue time against GPU time, launches per step, and where the Python time goes (cProfile, top 25 by own time).
GPU box: python tools/host_profile_clipseg.py"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from src.models.clipseg import segmentation as S
from src.losses.dice import DiceCELoss
from uia_hip import functional as UF
from uia_hip import _lib
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
for _ in range(3): step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(5):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(round((t1 - t0) * 1e3, 2)); tot.append(round((t2 - t0) * 1e3, 2))
print("enqueue ms", enq, "total ms", tot)
pr = cProfile.Profile()
pr.enable()
for _ in range(5): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(25)
st.print_callers("method 'to' of")
st.sort_stats("cumtime").print_stats(30)
