#!/bin/bash
# The CLIPSeg CLI (child process, loader + prefetcher) against bench.py's resident-batch line on the SAME box, alternating, with the board's power / shader clock sampled in both (round 6).
cd $GRAFT_REPO_ROOT
run() { d=$(mktemp -d); ( cd $d && UIA_SEG_AB=power timeout -k 10 120 python $GRAFT_REPO_ROOT/nextgen-uia_amd/src/models/clipseg/segmentation.py --dataset BUSI --synthetic --synthetic_train 7680 --synthetic_val 128 --synthetic_test 128 --batch_size 128 --epochs 3 --dtype bf16 --exp ab --stats_json $d/s.json > $d/log 2>&1 ); python -c "
import json,sys
try:
    o=json.load(open('$d/s.json')); e=o['epochs'][1:]; print('cli  ', round(sum(x['ms'] for x in e)/sum(x['updates'] for x in e),3), o.get('power'))
except Exception as ex: print('cli failed', ex, open('$d/log').read()[-600:])"; rm -rf $d; }
bench() { python bench.py --config clipseg --no-cpu-baseline --no-entry-point --steps 120 --warmup 30 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', o['ms_per_step'], o.get('power'))"; }
run; bench; run; bench
