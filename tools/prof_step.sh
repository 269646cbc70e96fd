#!/bin/bash
# rocprofv3 kernel trace of a short bench run; summary -> gpurun_out/prof/  (copy into profiles/ to commit)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o step -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --also-streams 0 --no-overlap-text --no-secondary --no-entry-point > $OUT/bench.log 2>&1
tail -2 $OUT/bench.log
ls -R $OUT | head -20
