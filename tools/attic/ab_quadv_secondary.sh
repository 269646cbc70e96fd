#!/bin/bash
# the two secondary configurations with the 256x256 bf16 launches on tile cfg 27 / 29, alternating with the default, one box
mkdir -p gpurun_out
for c in vitl_lora clipseg; do
for r in 1 2; do
  for v in "" "--quadv 27" "--quadv 29"; do
    out=$(bash tools/bench_ms.sh --config $c --no-entry-point --no-secondary $v) || exit 1
    echo "[$r] $c $v -> $out" | tee -a gpurun_out/ab_quadv_secondary.txt
  done
done
done
