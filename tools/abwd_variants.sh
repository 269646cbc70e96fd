#!/bin/bash
# Where does the bf16 attention backward spend its time?  Builds diagnostic variants of libuia_hip.so (results are WRONG in them)
# and times the kernel at the ViT-B shape.  Run on the GPU box: bash tools/abwd_variants.sh
cd $GRAFT_REPO_ROOT/nextgen-uia_amd/csrc
mkdir -p /tmp/abwd
for v in BASE ABWD_NO_DQ ABWD_NO_VALU ABWD_NO_DVDK ABWD_NO_DELTA ABWD_PROLOGUE_ONLY; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D$v -c attention_bwd.hip -o /tmp/abwd/attention_bwd_$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/abwd/lib_$v.so /tmp/abwd/attention_bwd_$v.o $(ls *.o | grep -v '^attention_bwd.o$' | tr '\n' ' ') -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  UIA_HIP_LIB=/tmp/abwd/lib_$v.so python3 - <<PY
import sys, torch
sys.path[:0] = ["$GRAFT_REPO_ROOT/nextgen-uia_amd"]
from uia_hip import ops
B, H, L, D = 256, 12, 197, 768
qkv = (torch.randn(B * L, 3 * D, device="cuda") * 0.5).bfloat16()
out = torch.empty(B * L, D, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, L, device="cuda")
ops.attn_fwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], out, B, H, L, lse=lse)
do = torch.randn_like(out); dqkv = torch.empty_like(qkv)
f = lambda: ops.attn_bwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], out, do, lse, dqkv[:, :D], dqkv[:, D:2*D], dqkv[:, 2*D:], B, H, L)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print("$v", round(e0.elapsed_time(e1) / 10 * 1e3, 1), "us")
PY
done
