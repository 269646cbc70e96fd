cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3r
mkdir -p $OUT
cd $GRAFT_REPO_ROOT/nextgen-uia_amd/csrc/tests
for shape in "65536 3072 768" "65536 2304 768"; do
for gm in 255 2 4 8 16 32; do
  cfg=$((8 + gm*256 + 3*65536))
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o f_${gm}_$(echo $shape | tr ' ' '_') -- ./test_gemm_exp one $cfg $shape > $OUT/log_${gm}.txt 2>&1
  grep BENCH $OUT/log_${gm}.txt | tail -1
done; done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,os,re
for f in sorted(glob.glob('gpurun_out/r3r/*counter_collection.csv')):
    v=[];d=[]
    for r in csv.DictReader(open(f)):
        if 'ring' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
            v.append(float(r['Counter_Value'])); d.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    if v: print(os.path.basename(f)[:40], 'launches', len(v), 'FETCHx2 MB', round(2*sum(v)/len(v)/1024,1), 'dur us (pmc pass)', round(sum(d)/len(d)/1e3,1))
PY
