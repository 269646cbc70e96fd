#!/bin/bash
# rocprofv3 kernel trace of a secondary configuration: bash tools/prof_config.sh vitl_lora|clipseg  -> gpurun_out/prof_<config>/
cd /tmp && export TMPDIR=/tmp
CFG=${1:-vitl_lora}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$CFG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o step -- python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-200
