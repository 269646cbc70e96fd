#!/bin/bash
# rocprofv3 kernel trace of a short bench run with --overlap-text (text tower on a second stream; per-kernel durations are inflated by sharing)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_overlap
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o step -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --also-streams 0 --no-secondary --no-entry-point > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-200
