#!/bin/bash
# rocprofv3 kernel trace of a short SERIAL bench run (both towers on one stream: per-kernel durations are standalone)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_serial
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o step -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap-text > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-200
