#!/bin/bash
# Is it the CLI's LOOP or the CLI's PROCESS that runs the CLIPSeg step 4 % slower than bench.py?  UIA_SEG_AB=post times bench.py's loop (resident batch, 120 steps) inside the CLI's process after
# its loaders are shut down; alternated with the CLI's own epochs and with bench.py on the same box.
cd $GRAFT_REPO_ROOT
run() { d=$(mktemp -d); ( cd $d && UIA_SEG_AB=post timeout -k 10 150 python $GRAFT_REPO_ROOT/nextgen-uia_amd/src/models/clipseg/segmentation.py --dataset BUSI --synthetic --synthetic_train 7680 --synthetic_val 128 --synthetic_test 128 --batch_size 128 --epochs 3 --dtype bf16 --exp ab --stats_json $d/s.json $1 > $d/log 2>&1 ); python -c "
import json
try:
    o=json.load(open('$d/s.json')); e=o['epochs'][1:]; print('cli epochs', round(sum(x['ms'] for x in e)/sum(x['updates'] for x in e),3), ' same process, resident batch: [loaders alive, loaders gone]', [round(v,3) for v in o['post_resident_ms']], '$1')
except Exception as ex: print('cli failed', ex, open('$d/log').read()[-600:])"; rm -rf $d; }
bench() { python bench.py --config clipseg --no-cpu-baseline --no-entry-point --steps 120 --warmup 30 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', o['ms_per_step'])"; }
run; bench; run "--num_workers 0"; bench; run
