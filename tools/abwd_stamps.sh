#!/bin/bash
# In-kernel cycle stamps of the bf16 attention backward (diagnostic build -DABWD_STAMPS, workgroup 1500 of the ViT-B launch): where do
# the waves of one head spend their cycles?  GPU box: bash tools/abwd_stamps.sh   (UIA_ABWD_CFGS="1,2,3" picks the kernel configurations)
cd $GRAFT_REPO_ROOT/nextgen-uia_amd/csrc
mkdir -p /tmp/abwd
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DABWD_STAMPS $ABWD_EXTRA -c attention_bwd.hip -o /tmp/abwd/attention_bwd_stamps.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/abwd/lib_stamps.so /tmp/abwd/attention_bwd_stamps.o $(ls *.o | grep -v '^attention_bwd.o$' | tr '\n' ' ') -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
UIA_HIP_LIB=/tmp/abwd/lib_stamps.so python3 - <<PY
import os, sys, ctypes, torch
sys.path[:0] = ["$GRAFT_REPO_ROOT/nextgen-uia_amd"]
from uia_hip import ops, _lib
B, H, L, D = 256, 12, int(os.environ.get("UIA_ABWD_L", "197")), 768
qkv = (torch.randn(B * L, 3 * D, device="cuda") * 0.5).bfloat16()
out = torch.empty(B * L, D, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, L, device="cuda")
ops.attn_fwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], out, B, H, L, lse=lse)
do = torch.randn_like(out); dqkv = torch.empty_like(qkv)
h = ctypes.CDLL("/tmp/abwd/lib_stamps.so")
for cfg in [int(c) for c in os.environ.get("UIA_ABWD_CFGS", "1,2,3").split(",")]:
    f = lambda: ops.attn_bwd(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], out, do, lse, dqkv[:, :D], dqkv[:, D:2*D], dqkv[:, 2*D:], B, H, L, cfg=cfg)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    assert h.uia_abwd_read_stamps(buf) == 0
    print(f"cfg {cfg}: {e0.elapsed_time(e1) * 100:.1f} us per launch (stamped build)")
    if cfg == 1:
        print("wave  top-wait   issue+delta   key tiles   dQ   | prologue  kernel   (cycles, summed over the 8 iterations)")
    else:
        print("wave  staging   KEY units   QRY units   n KEY   n QRY  | kernel   (cycles)")
    for w in range(8):
        r = [buf[w * 8 + i] for i in range(8)]
        if cfg == 1:
            print(f"{w:4d} {r[0]:9d} {r[1]:12d} {r[2]:11d} {r[3]:6d} | {r[4]:8d} {r[5]:8d}")
        else:
            print(f"{w:4d} {r[0]:9d} {r[1]:10d} {r[2]:10d} {r[3]:6d} {r[4]:6d} | {r[5]:8d}   KEY mask-free loop: {r[6]} cycles / {r[7]} blocks")
PY
