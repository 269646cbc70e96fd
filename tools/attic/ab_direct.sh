#!/bin/bash
# the store-only epilogues with and without their LDS bounce (libuia_hip_direct.so = -DUIA_EPI_DIRECT build): yardstick, then the step
mkdir -p gpurun_out
D=$GRAFT_REPO_ROOT/nextgen-uia_amd/uia_hip/libuia_hip_direct.so
for lib in "" $D; do
  echo "== lib ${lib:-default}" >> gpurun_out/ab_direct.txt
  UIA_HIP_LIB=$lib YARD_KB=1 YARD_CFGS=27,29 YARD_SHAPES=8192x8192x8192,65536x2304x768,65536x3072x768,65536x768x768 timeout -k 10 200 python tools/gemm_square_yardstick.py 2>&1 | grep "^M" >> gpurun_out/ab_direct.txt || exit 1
done
for r in 1 2; do
  for lib in "" $D; do
    out=$(UIA_HIP_LIB=$lib bash tools/bench_ms.sh --no-entry-point) || exit 1
    echo "[$r] step, lib ${lib:-default} -> $out" | tee -a gpurun_out/ab_direct.txt
  done
done
