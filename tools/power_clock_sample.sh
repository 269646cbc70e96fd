#!/bin/bash
# Samples the GPU's power draw, its cap and the shader / memory clocks (rocm-smi: sysfs, no HIP context) every 0.5 s while a short bench.py run is in flight:
# is the step running into the board's power limit (the in-kernel clock is 1.5-1.65 GHz against the 2.4 GHz the MFMA peak is quoted at)?
#   bash tools/power_clock_sample.sh [out_dir]
out=${1:-gpurun_out/power}; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower --showpower --showclocks --showperflevel > $out/idle.txt 2>&1
( while true; do date +%s.%N; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk"; sleep 0.5; done ) > $out/samples.txt 2>&1 &
S=$!
python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-secondary --no-entry-point --also-streams 0 > $out/bench.json 2> $out/bench.err
kill $S
python - "$out" <<'PY'
import re, sys, json
out = sys.argv[1]
txt = open(out + "/samples.txt").read()
pw = [float(x) for x in re.findall(r"Power \(W\):\s*([0-9.]+)", txt)]
sc = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)]
print(open(out + "/idle.txt").read()[-900:])
print("samples", len(pw), "power W: max", max(pw) if pw else None, "median of top half", sorted(pw)[len(pw) * 3 // 4] if pw else None)
print("sclk MHz seen:", sorted(set(sc)))
try:
    o = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1]); print("bench ms/step", o["ms_per_step"], o["roofline"].get("load_clock"))
except Exception as e:
    print("bench line unreadable", e)
PY
