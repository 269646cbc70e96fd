cd $GRAFT_REPO_ROOT
run() { d=$(mktemp -d); ( cd $d && timeout -k 10 120 python $GRAFT_REPO_ROOT/nextgen-uia_amd/src/models/clipseg/segmentation.py --dataset BUSI --synthetic --synthetic_train 7680 --synthetic_val 128 --synthetic_test 128 --batch_size 128 --epochs 3 --dtype bf16 --exp ab --stats_json $d/s.json $2 > $d/log 2>&1 ); python -c "
import json,sys
try:
    o=json.load(open('$d/s.json')); e=o['epochs'][1:]; print('$1', round(sum(x['ms'] for x in e)/sum(x['updates'] for x in e),3), [round(x['loader_wait_ms'],1) for x in e])
except Exception as ex: print('$1 failed', ex, open('$d/log').read()[-600:])"; rm -rf $d; }
bench() { python bench.py --config clipseg --no-cpu-baseline --no-entry-point --steps 120 --warmup 30 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', o['ms_per_step'])"; }
run cli-pinned-ring; UIA_RING_NO_PIN=1 run cli-staging; bench; run cli-pinned-ring; UIA_RING_NO_PIN=1 run cli-staging; run cli-2workers "--num_workers 2"
