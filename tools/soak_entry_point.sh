#!/bin/bash
# Soak run of the fine-tune CLI: 8 epochs x 150 updates at bs 256 (loader workers, shared batch ring, prefetcher, validation between epochs); prints per-epoch ms per update,
# loader wait, and the process's host RSS / device memory before and after — a leak or a stall in the loader path shows as drift.
cd $GRAFT_REPO_ROOT/nextgen-uia_amd
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/soak
cd $GRAFT_REPO_ROOT/gpurun_out/soak
( while sleep 20; do echo "$(date +%s) shm_free_MB $(df -m /dev/shm | tail -1 | awk '{print $4}') rss_MB $(ps -o rss= -C python | awk '{s+=$1} END {print int(s/1024)}')"; done ) > mem.log 2>&1 &
MON=$!
timeout -k 10 900 python $GRAFT_REPO_ROOT/nextgen-uia_amd/src/models/biomedclip/finetune.py --method mona --synthetic --synthetic_train $((256*150)) --synthetic_val 512 --batch_size 256 \
  --accumulation_steps 1 --epochs 8 --patience 99 --dtype bf16 --exp soak --stats_json stats.json > run.log 2>&1
rc=$?
kill $MON
echo "exit $rc"
python3 - <<'PY'
import json
d = json.load(open("stats.json"))
for i, e in enumerate(d["epochs"]):
    print(f"epoch {i+1}: {e['ms']/e['updates']:.3f} ms/update, loader wait {e['loader_wait_ms']:.1f} ms, enqueue {e['enqueue_ms']:.0f} ms, updates {e['updates']}")
print({k: d[k] for k in d if k != "epochs"})
PY
head -2 mem.log; tail -2 mem.log
