#!/bin/bash
# Diagnostic builds of the skinny weight-gradient kernel (results WRONG in some) timed at the two Mona shapes.  GPU box: bash tools/wgrad_variants.sh
cd $GRAFT_REPO_ROOT/nextgen-uia_amd/csrc
mkdir -p /tmp/wgv
OBJS=$(ls *.o | grep -v "^wgrad.o$" | tr "
" " ")     # every object of the library but the one rebuilt here
for v in ${WG_VARIANTS:-BASE WG_NO_ATOMIC WG_NO_BIAS WG_CHUNK_SLABS=8 WG_CHUNK_SLABS=16 WG_CHUNK_SLABS=2}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D$v -c wgrad.hip -o /tmp/wgv/wgrad.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/wgv/lib.so /tmp/wgv/wgrad.o $OBJS -L/opt/rocm/lib -lrccl
  UIA_HIP_LIB=/tmp/wgv/lib.so python3 - <<PY
import sys, torch
sys.path[:0] = ["$GRAFT_REPO_ROOT/nextgen-uia_amd"]
from uia_hip import ops
M = 256 * 197
dev = "cuda"
wide = torch.randn(M, 768, device=dev).bfloat16(); thin = torch.randn(M, 64, device=dev).bfloat16()
g2, b2 = torch.zeros(768, 64, device=dev), torch.zeros(768, device=dev)
g1, b1 = torch.zeros(64, 768, device=dev), torch.zeros(64, device=dev)
def timeit(f):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / 20 * 1e3, 1)
print("$v", "dW2 (I=768,J=64):", timeit(lambda: ops.wgrad(wide, thin, g2, b2)), "us   dW1 (I=64,J=768):", timeit(lambda: ops.wgrad(thin, wide, g1, b1)), "us")
PY
done
