#!/bin/bash
# one round of tiles on 72 CUs vs on 252 CUs: is the epilogue paced by each CU's own store path or by all CUs bursting together?  (cfg 27, full and without epilogue)
mkdir -p gpurun_out
# needs the diagnostic library: bash tools/attic/quadv_build_ablate.sh (before gpurun)
export UIA_HIP_LIB=$GRAFT_REPO_ROOT/nextgen-uia_amd/uia_hip/libuia_hip_ablate.so
for a in 0 32; do
  echo "== ablate $a" >> gpurun_out/quadv_ablate3.txt
  UIA_QUADV_ABLATE=$a YARD_KB=1 YARD_CFGS=27 YARD_SHAPES=8192x8192x8192,65536x2304x768,2048x2304x768,7168x2304x768,14336x2304x768 timeout -k 10 120 python tools/gemm_square_yardstick.py 2>&1 | grep "^M" >> gpurun_out/quadv_ablate3.txt || exit 1
done
