#!/bin/bash
# Which ingredient of the CLIPSeg CLI's epoch loop costs GPU time against bench.py's resident-batch loop?  UIA_SEG_AB=attrib runs the resident loop in the CLI's process with one ingredient
# added at a time (host-to-device copies on a side stream, the wait on their event, the step reading rotating buffers, an event record per step).
cd $GRAFT_REPO_ROOT
d=$(mktemp -d); ( cd $d && UIA_SEG_AB=post,attrib timeout -k 10 200 python $GRAFT_REPO_ROOT/nextgen-uia_amd/src/models/clipseg/segmentation.py --dataset BUSI --synthetic --synthetic_train 7680 --synthetic_val 128 --synthetic_test 128 --batch_size 128 --epochs 3 --dtype bf16 --exp ab --stats_json $d/s.json > $d/log 2>&1 ); python -c "
import json
try:
    o=json.load(open('$d/s.json')); e=o['epochs'][1:]; print('cli epochs', round(sum(x['ms'] for x in e)/sum(x['updates'] for x in e),3), 'resident', o['post_resident_ms'])
    for k,v in o['attrib_ms'].items(): print(f'  {v:7.3f}  {k}')
except Exception as ex: print('failed', ex, open('$d/log').read()[-900:])"; rm -rf $d
