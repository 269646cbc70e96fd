#!/bin/bash
# In-kernel cycle stamps of the bf16 attention forward (diagnostic build -DAFWD_STAMPS, workgroup 1500 of the ViT-B launch).
cd $GRAFT_REPO_ROOT/nextgen-uia_amd/csrc
mkdir -p /tmp/afwd
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DAFWD_STAMPS -c attention_fwd.hip -o /tmp/afwd/attention_fwd_stamps.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/afwd/lib_stamps.so /tmp/afwd/attention_fwd_stamps.o $(ls *.o | grep -v '^attention_fwd.o$' | tr '\n' ' ') -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
UIA_HIP_LIB=/tmp/afwd/lib_stamps.so python3 - <<PY
import sys, ctypes, torch
sys.path[:0] = ["$GRAFT_REPO_ROOT/nextgen-uia_amd"]
from uia_hip import ops
B, H, L, D = 256, 12, 197, 768
qkv = (torch.randn(B * L, 3 * D, device="cuda") * 0.5).bfloat16()
out = torch.empty(B * L, D, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, L, device="cuda")
for _ in range(5): ops.attn_fwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], out, B, H, L, lse=lse)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
h = ctypes.CDLL("/tmp/afwd/lib_stamps.so")
assert h.uia_afwd_read_stamps(buf) == 0
print("wave  issue  wait+barrier  S+max  exp+sum   PV  store | kernel   (cycles; S..store summed over the wave's query tiles)")
for w in range(7):
    r = [buf[w * 8 + i] for i in range(7)]
    print(f"{w:4d} {r[0]:6d} {r[1]:10d} {r[2]:9d} {r[3]:8d} {r[4]:6d} {r[5]:6d} | {r[6]:7d}")
PY
