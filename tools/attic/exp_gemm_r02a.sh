#!/bin/bash
# round-2 experiment A: tile-order group size and K-blocked (diagnostic) operand addressing on the ring GEMM
# the diagnostic binaries are not tracked: build them here (hipcc is on the GPU box too)
make -C $GRAFT_REPO_ROOT/nextgen-uia_amd/csrc tests/test_gemm_exp tests/test_gemm_dma tests/test_gemm_stamps > /dev/null || exit 1
cd $GRAFT_REPO_ROOT/nextgen-uia_amd/csrc/tests
OUT=$GRAFT_REPO_ROOT/gpurun_out/exp_r02a
mkdir -p $OUT
timeout 300 ./test_gemm_exp exp > $OUT/exp.log 2>&1
for kb in 0 1 2 3; do
  timeout 120 ./test_gemm_dma $((8 + kb * 65536)) 50432 2304 768 0 0 >> $OUT/dma.log 2>&1
  timeout 120 ./test_gemm_dma $((8 + kb * 65536)) 50432 768 768 0 0 >> $OUT/dma.log 2>&1
done
for kb in 0 3; do
  timeout 120 ./test_gemm_stamps $((8 + kb * 65536)) 50432 2304 768 0 0 >> $OUT/stamps.log 2>&1
  timeout 120 ./test_gemm_stamps $((8 + kb * 65536 + 8 * 256)) 50432 3072 768 1 0 >> $OUT/stamps.log 2>&1
done
cat $OUT/exp.log $OUT/dma.log $OUT/stamps.log
