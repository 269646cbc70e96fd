#!/bin/bash
# Soak run of the CLIPSeg CLI (BASELINE configs[3]'s entry point): 41 epochs x 60 iterations at bs 128 — validation + test passes at epochs 10, 20, 30, 40 (the last), best-Dice checkpoints,
# then test() — with the loader workers, the shared ring and the prefetcher running throughout; per-epoch ms per iteration, loader wait, host RSS / shm before and after.
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/soak_clipseg
cd $GRAFT_REPO_ROOT/gpurun_out/soak_clipseg
rm -rf runs
( while sleep 10; do echo "$(date +%s) shm_free_MB $(df -m /dev/shm | tail -1 | awk '{print $4}') rss_MB $(ps -o rss= -C python | awk '{s+=$1} END {print int(s/1024)}')"; done ) > mem.log 2>&1 &
MON=$!
timeout -k 10 600 python $GRAFT_REPO_ROOT/nextgen-uia_amd/src/models/clipseg/segmentation.py --dataset BUSI --synthetic --synthetic_train $((128*60)) --synthetic_val 256 --synthetic_test 256 \
  --batch_size 128 --epochs 41 --lr 3e-4 --dtype bf16 --exp soak --stats_json stats.json > run.log 2>&1
rc=$?
kill $MON
echo "exit $rc"
python3 - <<'PY'
import json
d = json.load(open("stats.json"))
ms = [e["ms"] / e["updates"] for e in d["epochs"]]
print("ms per iteration, epochs 1 / 2 / 11 / 21 / 31 / 41:", [round(ms[i], 3) for i in (0, 1, 10, 20, 30, 40)], " min / median / max over epochs 2-41:", round(min(ms[1:]), 3), round(sorted(ms[1:])[20], 3), round(max(ms[1:]), 3))
print("loader wait ms per epoch (max over epochs 2-41):", round(max(e["loader_wait_ms"] for e in d["epochs"][1:]), 1), " iters", d["iters"], " best val dice", round(d["best_val_dice"], 4))
PY
grep -c "iter: " runs/soak/BUSI/train/log.log; grep "iter: " runs/soak/BUSI/train/log.log | tail -4; ls runs/soak/BUSI/test/*/; cat runs/soak/BUSI/test/*/results.csv
head -1 mem.log; tail -1 mem.log
