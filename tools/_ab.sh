#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo "default      $(bash tools/bench_ms.sh --no-secondary --also-streams 0)"
  echo "split 0.5    $(bash tools/bench_ms.sh --no-secondary --also-streams 0 --image-split 0.5)"
  echo "split 0.86   $(bash tools/bench_ms.sh --no-secondary --also-streams 0 --image-split 0.86)"
  echo "split 0.33   $(bash tools/bench_ms.sh --no-secondary --also-streams 0 --image-split 0.33)"
done
for i in 1 2; do
  echo "vitl default   $(bash tools/bench_ms.sh --config vitl_lora --steps 8)"
  echo "vitl split 0.5 $(bash tools/bench_ms.sh --config vitl_lora --steps 8 --image-split 0.5)"
done
