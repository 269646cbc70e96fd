#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  echo "default (chains)     $(bash tools/bench_ms.sh --no-secondary --also-streams 0)"
  echo "chains no-hh-short-k $(bash tools/bench_ms.sh --no-secondary --also-streams 0 --no-half-height-short-k)"
  echo "chains short-k-n 0   $(bash tools/bench_ms.sh --no-secondary --also-streams 0 --short-k-half-n 0)"
  echo "chains both off      $(bash tools/bench_ms.sh --no-secondary --also-streams 0 --short-k-half-n 0 --no-half-height-short-k)"
  echo "no-tail-split        $(bash tools/bench_ms.sh --no-secondary --also-streams 0 --no-tail-split)"
done
