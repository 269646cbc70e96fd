#!/bin/bash
# rocprofv3 kernel trace of two short bench runs on ONE box: the default, and the default with the flags in $AB_FLAGS (e.g. --no-ln-fold,
# --no-kblock-act); summaries -> gpurun_out/prof_a (default), gpurun_out/prof_b (with the flags).  Compare: tools/cmp_kernel_stats.py
cd /tmp && export TMPDIR=/tmp
AB_FLAGS=${AB_FLAGS:---no-ln-fold}
for V in a b; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$V
  mkdir -p $OUT
  cd $GRAFT_REPO_ROOT
  FLAG=""; [ $V = b ] && FLAG="$AB_FLAGS"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o step -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap-text --no-secondary --no-entry-point $FLAG > $OUT/bench.log 2>&1 || exit 1
  tail -1 $OUT/bench.log | cut -c1-160
done
