#!/bin/bash
# PMC counters for EVERY kernel of one training step (bench.py --steps 1 --warmup 1), separate rocprofv3 --pmc passes as
# MI355X_MICROARCH.md prescribes (no tracing domains beside --pmc; FETCH_SIZE and WRITE_SIZE do not fit one pass):
#   pass A  SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
#   pass B  FETCH_SIZE          pass C  WRITE_SIZE          pass D  TCC_HIT_sum TCC_MISS_sum
# Output: gpurun_out/pmc_step/summary.json (+ a printed table of the top kernels by GRBM_GUI_ACTIVE); copy into profiles/ to commit.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_step${UIA_PMC_TAG:+_$UIA_PMC_TAG}          # UIA_PMC_TAG=vitl_lora UIA_PMC_ARGS="--config vitl_lora": another configuration
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 420 rocprofv3 --pmc $set --output-format csv -d $OUT -o pass$i -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --also-streams 0 --no-overlap-text --no-secondary --no-entry-point $UIA_PMC_ARGS > $OUT/pass$i.log 2>&1
  tail -1 $OUT/pass$i.log | cut -c1-120
done
python3 - <<PY
import csv, glob, json, collections, re
def short(name):
    m = re.search(r"(gemm_tn_[a-z_]*kernelI(?:DF16b|f)(?:L(?:in?[0-9]+|b[01])E)+E)", name)
    if m: return m.group(1)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(_ZN12_GLOBAL__N_1\d+)?([A-Za-z_0-9]+(<[^>(]*>)?)", name)
    return (m.group(2) if m else name)[:90]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {}
for k, cs in acc.items():
    d = {c: {"per_launch": v[0] / v[1], "launches": v[1]} for c, v in cs.items()}
    g = cs.get("GRBM_GUI_ACTIVE"); m = cs.get("SQ_VALU_MFMA_BUSY_CYCLES"); fsz = cs.get("FETCH_SIZE"); wsz = cs.get("WRITE_SIZE")
    if g and g[0] > 0:
        cyc = g[0] / 8.0                                  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        d["derived"] = {"kernel_cycles_per_launch": cyc / g[1]}
        if m: d["derived"]["mfma_busy_frac"] = m[0] / (cyc * 256 * 4)          # busy cycles summed over 256 CUs x 4 SIMDs
        if fsz and wsz: d["derived"]["hbm_bytes_per_launch"] = (2.0 * fsz[0] / fsz[1] + wsz[0] / wsz[1]) * 1024   # FETCH_SIZE doubled (gfx950 note)
        h, ms = cs.get("TCC_HIT_sum"), cs.get("TCC_MISS_sum")
        if h and ms and h[0] + ms[0] > 0: d["derived"]["l2_hit_rate"] = h[0] / (h[0] + ms[0])
    out[k] = d
json.dump(out, open("$OUT/summary.json", "w"), indent=1, sort_keys=True)
top = sorted(((v["GRBM_GUI_ACTIVE"]["per_launch"] * v["GRBM_GUI_ACTIVE"]["launches"], k) for k, v in out.items() if "GRBM_GUI_ACTIVE" in v), reverse=True)[:18]
for _, k in top:
    dv = out[k].get("derived", {})
    print("%-78s n=%4d  cyc/launch %9.0f  mfma_busy %.3f  hbm MB/launch %8.1f  L2 hit %.2f" % (k, out[k]["GRBM_GUI_ACTIVE"]["launches"], dv.get("kernel_cycles_per_launch", 0),
          dv.get("mfma_busy_frac", 0), dv.get("hbm_bytes_per_launch", 0) / 1e6, dv.get("l2_hit_rate", 0)))
PY
