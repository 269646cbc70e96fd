#!/bin/bash
# What bounds the sweeps of the barrier-free attention backward (unit kernel, cfg 2, and its persistent form, cfg 5)?  Diagnostic builds of
# libuia_hip.so with one ingredient of the block loop removed at a time (results are WRONG in them), timed at the ViT-B shape.
#   ABWU_NO_TR   three quarters of the transposed LDS reads      ABWU_NO_ROWS  the row-fragment LDS reads of the next block
#   ABWU_NO_EXP  the exponentials                                 ABWU_NO_DVDK  the dV / dK / dQ products       ABWU_NO_SDP  the S / dP products
cd $GRAFT_REPO_ROOT/nextgen-uia_amd/csrc
mkdir -p /tmp/abwu
for v in BASE ABWU_NO_TR ABWU_NO_ROWS ABWU_NO_EXP ABWU_NO_DVDK ABWU_NO_SDP "ABWU_NO_TR -DABWU_NO_ROWS" "ABWU_NO_DVDK -DABWU_NO_SDP" "ABWU_NO_TR -DABWU_NO_ROWS -DABWU_NO_DVDK -DABWU_NO_SDP -DABWU_NO_EXP"; do
  tag=$(echo $v | tr -d ' ' | tr -d '-')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -D$v -c attention_bwd.hip -o /tmp/abwu/a_$tag.o 2>/dev/null || { echo "build failed: $v"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/abwu/lib_$tag.so /tmp/abwu/a_$tag.o $(ls *.o | grep -v '^attention_bwd.o$' | tr '\n' ' ') -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  UIA_HIP_LIB=/tmp/abwu/lib_$tag.so python3 - <<PY
import sys, torch
sys.path[:0] = ["$GRAFT_REPO_ROOT/nextgen-uia_amd"]
from uia_hip import ops
B, H, L, D = 256, 12, 197, 768
qkv = (torch.randn(B * L, 3 * D, device="cuda") * 0.5).bfloat16()
out = torch.empty(B * L, D, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, L, device="cuda")
ops.attn_fwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], out, B, H, L, lse=lse)
do = torch.randn_like(out); dqkv = torch.empty_like(qkv)
res = []
for cfg in (2, 5):
    f = lambda: ops.attn_bwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], out, do, lse, dqkv[:, :D], dqkv[:, D:2*D], dqkv[:, 2*D:], B, H, L, cfg=cfg)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    res.append(round(e0.elapsed_time(e1) / 10 * 1e3, 1))
print(f"{'$v':80s} cfg 2: {res[0]:6.1f} us   cfg 5: {res[1]:6.1f} us")
PY
done
