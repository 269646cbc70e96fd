#!/bin/bash
# K-loop ablations of tile cfg 27 (results wrong by construction; timing only): 1 no barrier, 2 loads from sub-tiles 0/1 only, 4 no ds_write, 8 no fragment reads,
# 16 no global loads, 20 = 4+16, 29 = MFMAs alone
mkdir -p gpurun_out
# needs the diagnostic library: bash tools/attic/quadv_build_ablate.sh (before gpurun)
export UIA_HIP_LIB=$GRAFT_REPO_ROOT/nextgen-uia_amd/uia_hip/libuia_hip_ablate.so
for a in 0 1 2 4 8 16 20 29; do
  echo "== ablate $a" >> gpurun_out/quadv_ablate.txt
  UIA_QUADV_ABLATE=$a YARD_CFGS=27 YARD_SHAPES=8192x8192x8192,65536x2304x768 timeout -k 10 120 python tools/gemm_square_yardstick.py 2>&1 | grep "^M" >> gpurun_out/quadv_ablate.txt || exit 1
done
