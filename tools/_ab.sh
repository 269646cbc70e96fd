#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo "default      $(bash tools/bench_ms.sh --no-secondary --also-streams 0)"
  echo "no-grad3     $(bash tools/bench_ms.sh --no-secondary --also-streams 0 --no-grad-resid3)"
done
