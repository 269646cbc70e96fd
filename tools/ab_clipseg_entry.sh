#!/bin/bash
# A/B of the CLIPSeg entry point's loop (round 6): the child process bench.py times, under measurement knobs.  Prints ms per iteration of epochs 2-3.
#   UIA_POLL_LAG  steps between a guard word's copy and its examination on the host (uia_hip.functional.POLL_LAG: how far the host may run ahead of the GPU)
#   UIA_SEG_AB    nogc | switch | resident (the same device batch every iteration, the prefetcher idle)
#   bash tools/ab_clipseg_entry.sh [out_dir]
out=${1:-gpurun_out/ab_clipseg_entry}; mkdir -p $out
run() {  # $1 = label, env from the caller
  d=$(mktemp -d); ( cd $d && timeout -k 10 120 python $GRAFT_REPO_ROOT/nextgen-uia_amd/src/models/clipseg/segmentation.py --dataset BUSI --synthetic --synthetic_train 7680 --synthetic_val 128 --synthetic_test 128 \
      --batch_size 128 --epochs 3 --dtype bf16 --exp ab --stats_json $d/s.json > $d/log 2>&1 )
  python - "$d/s.json" "$1" "$d/log" <<'PY' | tee -a $out/ab.txt
import json, sys
try:
    o = json.load(open(sys.argv[1])); e = o["epochs"][1:]
    print(sys.argv[2], round(sum(x["ms"] for x in e) / sum(x["updates"] for x in e), 3), "ms/iter; enqueue", [round(x["enqueue_ms"] / x["updates"], 3) for x in e], "drain", [round(x["drain_ms"], 1) for x in e], "loader wait", [round(x["loader_wait_ms"], 1) for x in e])
except Exception as ex:
    print(sys.argv[2], "failed", ex, open(sys.argv[3]).read()[-400:])
PY
  rm -rf $d
}
for lag in 1 2 3 1 2 3; do UIA_POLL_LAG=$lag run "lag$lag"; done
UIA_POLL_LAG=1 UIA_SEG_AB=resident run "lag1+resident"
UIA_POLL_LAG=3 UIA_SEG_AB=resident run "lag3+resident"
