#!/bin/bash
# Kernel-trace durations of tools/time_quad.py --diag (cfg 25 K-loop ablation): per-kernel averages from rocprofv3, not host events.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/quad_diag
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o q -- python3 tools/time_quad.py --diag --rounds 2 --iters 5 > $OUT/log.txt 2>&1
tail -4 $OUT/log.txt
python3 tools/show_kernel_stats.py $OUT/q_kernel_stats.csv 14
