#!/bin/bash
# Which kernels does the vendor library (torch.matmul -> hipBLASLt) pick for the step's bf16 GEMM shapes?  Kernel trace of tools/gemm_yardstick.py:
# the Tensile kernel names carry the macro tile (MT), the MFMA shape (MI), depthU and the wave tiling.  tools/ only.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/vendor_names
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o y -- python3 tools/gemm_yardstick.py --rounds 2 --iters 5 > $OUT/yardstick.log 2>&1
cat $OUT/yardstick.log | tail -20
python3 - <<PY
import csv, glob, collections
acc = collections.OrderedDict()
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "uia" in k or "gemm_tn" in k or "elementwise" in k or "copy" in k.lower():
            continue
        d = acc.setdefault(k, [0, 0.0, r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count")])
        d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
for k, d in acc.items():
    print(f"{d[0]:4d} launches  avg {d[1] / d[0]:8.1f} us  grid {d[2]} wg {d[3]} lds {d[4]} vgpr {d[5]} agpr {d[6]}  {k}")
PY
