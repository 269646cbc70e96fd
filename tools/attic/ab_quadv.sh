#!/bin/bash
# in-step A/B: the default tree against --quadv 27 / 29 (256x256 bf16 launches on tile cfg 27 / its persistent grid 29), alternating, one box
mkdir -p gpurun_out
for r in 1 2; do
  for v in "" "--quadv 27" "--quadv 29"; do
    out=$(bash tools/bench_ms.sh --no-entry-point $v) || exit 1
    echo "[$r] default $v -> $out" | tee -a gpurun_out/ab_quadv.txt
  done
done
