#!/bin/bash
# A/B of the CLIPSeg entry point's host-side loop (round 6): the child process bench.py times, under measurement knobs (UIA_SEG_AB).  Prints ms per iteration of epochs 2-3.
#   bash tools/ab_clipseg_entry.sh [out_dir]
out=${1:-gpurun_out/ab_clipseg_entry}; mkdir -p $out
for ab in base nogc switch nogc+switch resident base; do
  d=$(mktemp -d); ( cd $d && UIA_SEG_AB=$ab timeout -k 10 120 python $OLDPWD/nextgen-uia_amd/src/models/clipseg/segmentation.py --dataset BUSI --synthetic --synthetic_train 7680 --synthetic_val 128 --synthetic_test 128 \
      --batch_size 128 --epochs 3 --dtype bf16 --exp ab --stats_json $d/s.json > $d/log 2>&1 )
  python - "$d/s.json" "$ab" <<'PY' | tee -a $out/ab.txt
import json, sys
try:
    o = json.load(open(sys.argv[1])); e = o["epochs"][1:]
    print(sys.argv[2], round(sum(x["ms"] for x in e) / sum(x["updates"] for x in e), 3), "ms/iter; enqueue", [round(x["enqueue_ms"] / x["updates"], 3) for x in e], "loader wait", [round(x["loader_wait_ms"], 1) for x in e])
except Exception as ex:
    print(sys.argv[2], "failed", ex)
PY
  rm -rf $d
done
