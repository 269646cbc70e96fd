#!/bin/bash
# Diagnostic builds of the fused Mona pre-norm backward (uia_mona_pre_bwd_du / _du3) timed at the ViT-B/16 shape; gradients WRONG in the MPB_NO_ACC builds.
# GPU box: bash tools/mpb_variants.sh
cd $GRAFT_REPO_ROOT/nextgen-uia_amd/csrc
mkdir -p /tmp/mpb
OBJS=$(ls *.o | grep -v "^mona.o$" | tr "\n" " ")
for v in ${MPB_VARIANTS:-BASE MPB_NO_ACC}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D$v -c mona.hip -o /tmp/mpb/mona.o 2>/dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/mpb/lib.so /tmp/mpb/mona.o $OBJS -L/opt/rocm/lib -lrccl || exit 1
  UIA_HIP_LIB=/tmp/mpb/lib.so python3 - <<PY
import sys, torch
sys.path[:0] = ["$GRAFT_REPO_ROOT", "$GRAFT_REPO_ROOT/nextgen-uia_amd"]
from uia_hip import ops
M, D, dt = 256 * 197, 768, torch.bfloat16
dev = torch.device("cuda", 0)
x, dy = torch.randn(M, D, device=dev), torch.randn(M, D, device=dev)
dyh, dyl = ops.float_to_three_byte(dy)
dtt = torch.randn(M, 64, device=dev).to(dt)
w1t = (torch.randn(D, 64, device=dev) * 0.05).to(dt)
nw, nb, g, gx = (torch.randn(D, device=dev) for _ in range(4))
G = [torch.zeros(D, device=dev) for _ in range(4)]
dx, dxt, lo = torch.empty_like(x), ops.kb_empty(M, D, dt, dev), torch.empty(M, D, device=dev, dtype=torch.int8)
def timeit(f):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / 20 * 1e3, 1)
a = timeit(lambda: ops.mona_pre_bwd(None, x, dy, nw, nb, g, gx, dx, dxt, *G, dt_w1t=(dtt, w1t)))
b = timeit(lambda: ops.mona_pre_bwd(None, x, (dyh, dyl), nw, nb, g, gx, None, dxt, *G, dt_w1t=(dtt, w1t), dx_lo=lo))
print("$v: fp32 gradients in/out", a, "us   three-byte in/out", b, "us (both + the 6 us reduction launch)", flush=True)
PY
done
