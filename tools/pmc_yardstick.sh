#!/bin/bash
# MFMA-busy of the yardstick's GEMM launches (tools/gemm_square_yardstick.py; one rocprofv3 --pmc pass, no tracing domains beside it):
#   YARD_CFGS=27,29 YARD_SHAPES=8192x8192x8192,65536x2304x768 bash tools/pmc_yardstick.sh   -> gpurun_out/pmc_yardstick/summary.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_yardstick
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export YARD_KB=${YARD_KB:-1} YARD_CFGS=${YARD_CFGS:-27,29} YARD_SHAPES=${YARD_SHAPES:-8192x8192x8192,65536x2304x768,65536x768x3072}
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $OUT -o y -- python3 tools/gemm_square_yardstick.py > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_tn" not in n and "Cijk" not in n: continue
        m = re.search(r"(gemm_tn_\w+?_kernel)", n)
        key = ((m.group(1) if m else "hipBLASLt " + n[:40]) + " grid " + r.get("Grid_Size", "?"))
        a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open("$OUT/summary.txt", "w") as out:
    for k, cs in sorted(acc.items()):
        g, m = cs.get("GRBM_GUI_ACTIVE"), cs.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if not g or not m: continue
        cyc = g[0] / 8.0
        line = "%-70s n=%3d  cycles/launch %10.0f  mfma_busy %.3f" % (k, g[1], cyc / g[1], m[0] / (cyc * 256 * 4))
        print(line); out.write(line + "\n")
PY
