#!/bin/bash
# rocprofv3 kernel trace of two short bench runs on ONE box: default (LayerNorm folded) and --no-ln-fold; summaries -> gpurun_out/prof_fold, prof_nofold
cd /tmp && export TMPDIR=/tmp
for V in fold nofold; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$V
  mkdir -p $OUT
  cd $GRAFT_REPO_ROOT
  FLAG=""; [ $V = nofold ] && FLAG="--no-ln-fold"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o step -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $FLAG > $OUT/bench.log 2>&1 || exit 1
  tail -1 $OUT/bench.log | cut -c1-200
done
