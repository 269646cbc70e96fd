#!/bin/bash
# PMC counters for one GEMM config: tools/pmc_gemm.sh <cfg> [M N K]
# (every pass under its own timeout: a counter set rocprofv3 rejects aborts and then hangs until the box limit)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$(echo "$@" | tr " " "_")
mkdir -p $OUT
BIN=$GRAFT_REPO_ROOT/nextgen-uia_amd/csrc/tests/test_gemm
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 150 rocprofv3 --pmc $set --output-format csv -d $OUT -o $tag -- $BIN one "$@" > $OUT/$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, (v, n) in acc.items():
        print(f"{k:36s} per-launch {v/n:16.1f}  (n={n})")
PY
grep BENCH $OUT/*.log | head -3
