#!/bin/bash
# bash tools/bench_ms.sh <bench.py args...>  -> "ms_per_step value" of one bench.py run (A/B loops inside one gpurun call)
python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
