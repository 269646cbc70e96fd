#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "default        $(bash tools/bench_ms.sh --no-secondary --also-streams 0)"
for g in "768=4" "768=16" "768=32" "2304=4" "2304=16" "3072=4" "3072=16" "768=16,2304=16,3072=16"; do
  echo "group $g   $(bash tools/bench_ms.sh --no-secondary --also-streams 0 --tile-group $g)"
done
echo "default        $(bash tools/bench_ms.sh --no-secondary --also-streams 0)"
