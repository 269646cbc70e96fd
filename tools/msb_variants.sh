#!/bin/bash
# Where does the bf16 Mona spatial backward (one workgroup per image) spend its time?  Builds diagnostic variants of libuia_hip.so that
# return after phase k (results are WRONG in them) and times the kernel at the ViT-B/16 shape.  Run on the GPU box: bash tools/msb_variants.sh
cd $GRAFT_REPO_ROOT/nextgen-uia_amd/csrc
mkdir -p /tmp/msb
OBJS=$(ls *.o | grep -v "^mona.o$" | tr "
" " ")     # every object of the library but the one rebuilt here
for v in ${MSB_VARIANTS:-0 1 2 3 4 5 6 7 8 9 99}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMSB_STOP=$v -c mona.hip -o /tmp/msb/mona_$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/msb/lib_$v.so /tmp/msb/mona_$v.o $OBJS -L/opt/rocm/lib -lrccl
  UIA_HIP_LIB=/tmp/msb/lib_$v.so python3 - <<PY
import sys, torch
sys.path[:0] = ["$GRAFT_REPO_ROOT/nextgen-uia_amd"]
from uia_hip import ops
B, h, w = 256, 14, 14
dev = "cuda"
t = (torch.randn(B * (h * w + 1), 64, device=dev)).bfloat16(); dd = torch.randn_like(t); dt = torch.empty_like(t)
shapes = dict(conv1_w=(64, 9), conv1_b=(64,), conv2_w=(64, 25), conv2_b=(64,), conv3_w=(64, 49), conv3_b=(64,), proj_w=(64, 64), proj_b=(64,), freq=(64,))
P = {k: torch.randn(*s, device=dev) * 0.1 for k, s in shapes.items()}
G = {k: torch.zeros_like(v) for k, v in P.items()}
f = lambda: ops.mona_spatial_bwd("freq_enhanced", B, h, w, t, P, dd, dt, G, p_drop=0.1, seed=5)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print("stop after phase $v:", round(e0.elapsed_time(e1) / 20 * 1e3, 1), "us (kernel + workspace reduction)")
PY
done
