#!/bin/bash
# Four-wave workgroups, two per CU (tile cfgs 17 / 18, experiment builds) against cfg 8 and cfg 14 on the short-K, wide-N shapes.  GPU box: bash tools/cfg1718_variants.sh
cd $GRAFT_REPO_ROOT/nextgen-uia_amd/csrc
mkdir -p /tmp/c1718
OBJS=$(ls *.o | grep -v "^gemm.o$" | tr "\n" " ")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DUIA_GEMM_CFG1718 -c gemm.hip -o /tmp/c1718/gemm.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/c1718/lib.so /tmp/c1718/gemm.o $OBJS -L/opt/rocm/lib -lrccl || exit 1
UIA_HIP_LIB=/tmp/c1718/lib.so python3 - <<PY
import sys, statistics, torch
sys.path[:0] = ["$GRAFT_REPO_ROOT", "$GRAFT_REPO_ROOT/nextgen-uia_amd"]
from uia_hip import ops
ops.RING_CFGS = ops.RING_CFGS + (17, 18)
dev = torch.device("cuda", 0); dt = torch.bfloat16
def timeit(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for (M, N, K) in ((65536, 3072, 768), (50432, 3072, 768), (65536, 2304, 768), (65536, 768, 3072)):
    a = ops.KBlocked(torch.randn(K // 32, M, 32, device=dev).to(dt))
    w = ops.PackedW((torch.randn(N, K, device=dev) * K ** -0.5).to(dt))
    bias = torch.randn(N, device=dev)
    out = ops.kb_empty(M, N, dt, dev)
    res, ref = {}, None
    for cfg in (8, 14, 17, 18):
        f = lambda: ops.gemm(a, w, bias=bias, act="gelu", out_t=out, tile_cfg=cfg)
        try:
            res[cfg] = round(statistics.median(timeit(f) for _ in range(3)), 1)
            if ref is None: ref = out.t.clone()
            else: assert torch.equal(ref, out.t), cfg
        except Exception as e:
            res[cfg] = str(e)[:80]
    print(M, N, K, "bias+gelu store (us):", res, flush=True)
PY
