#!/bin/bash
# HBM traffic of the dominant kernel (the 256x256 ring GEMM family) per launch, for bench.py's roofline.traffic.
# Two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short serial bench run, as MI355X_MICROARCH.md §HBM prescribes;
# each pass under its own timeout.  Output: gpurun_out/traffic/summary.json  (copy to profiles/ to commit).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --pmc $c --output-format csv -d $OUT -o $c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --also-streams 0 --no-overlap-text --no-secondary --no-entry-point > $OUT/$c.log 2>&1
  tail -1 $OUT/$c.log | cut -c1-160
done
python3 - <<PY
import csv, glob, json, collections, re
def short(name):
    # rocprofv3 leaves these names mangled: "_ZN12_GLOBAL__N_119gemm_tn_ring_kernelIDF16bLi256E...Li81EEEv13uia_gemm_desc"
    # -> "gemm_tn_ring_kernelIDF16bLi256E...Li81EE" (the fragment bench.py reports as roofline.kernel_in_rocprof_csv)
    m = re.search(r"(gemm_tn_[a-z_]*kernelI(?:DF16b|f)(?:Lin?[0-9]+E|Lb[01]E)+E)", name)
    return m.group(1) if m else None
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$OUT/*%s*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            k = short(r["Kernel_Name"])
            if k is None: continue
            a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += 1
    out[c] = {k: {"sum": v[0], "launches": v[1]} for k, v in acc.items()}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k in sorted(out["FETCH_SIZE"]):
    f, w = out["FETCH_SIZE"][k], out["WRITE_SIZE"].get(k, {"sum": 0, "launches": 1})
    print("%-70s launches %4d  fetch %8.1f MB x2  write %8.1f MB" % (k, f["launches"], f["sum"] / f["launches"] / 1024, w["sum"] / w["launches"] / 1024))
PY
