#!/bin/bash
# HBM traffic of the dominant kernel (the 256x256 ring GEMM family) per launch, for bench.py's roofline.traffic.
# Two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short serial bench run, as MI355X_MICROARCH.md §HBM prescribes;
# each pass under its own timeout.  Output: gpurun_out/traffic/summary.json  (copy to profiles/ to commit).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --pmc $c --output-format csv -d $OUT -o $c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-overlap-text > $OUT/$c.log 2>&1
  tail -1 $OUT/$c.log | cut -c1-160
done
python3 - <<PY
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$OUT/*%s*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            k = "ring_gemm" if "gemm_tn_ring_kernel" in r["Kernel_Name"] else "other"
            a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += 1
    out[c] = {k: {"sum": v[0], "launches": v[1]} for k, v in acc.items()}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out))
PY
