#!/bin/bash
# tile-seam ablations of cfg 27: 32 = no epilogue, 61 = no epilogue and MFMAs alone in the K loop
mkdir -p gpurun_out
# needs the diagnostic library: bash tools/attic/quadv_build_ablate.sh (before gpurun)
export UIA_HIP_LIB=$GRAFT_REPO_ROOT/nextgen-uia_amd/uia_hip/libuia_hip_ablate.so
for a in 0 32 61; do
  echo "== ablate $a" >> gpurun_out/quadv_ablate2.txt
  UIA_QUADV_ABLATE=$a YARD_KB=1 YARD_CFGS=27 YARD_SHAPES=8192x8192x8192,65536x2304x768,65536x768x768 timeout -k 10 120 python tools/gemm_square_yardstick.py 2>&1 | grep "^M" >> gpurun_out/quadv_ablate2.txt || exit 1
done
