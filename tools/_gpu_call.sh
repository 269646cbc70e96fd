set -o pipefail
mkdir -p gpurun_out/r3f
cd nextgen-uia_amd/csrc/tests
for bin in test_gemm_stamps; do
  timeout 120 ./$bin 8 49152 3072 768 3 0 200 >> ../../../gpurun_out/r3f/$bin.log 2>&1
  timeout 120 ./$bin 8 65536 768 768 5 0 200 >> ../../../gpurun_out/r3f/$bin.log 2>&1
  timeout 120 ./$bin 8 65536 768 3072 5 0 200 >> ../../../gpurun_out/r3f/$bin.log 2>&1
  timeout 120 ./$bin 8 43520 768 3072 2 0 200 >> ../../../gpurun_out/r3f/$bin.log 2>&1
  timeout 120 ./$bin 8 43520 768 768 5 0 200 >> ../../../gpurun_out/r3f/$bin.log 2>&1
done
cd ../../..
echo ---- pipelined + touch;    grep -hE "^STAMPS" gpurun_out/r3f/test_gemm_stamps.log | cut -c1-190
grep -hE "^BENCH" gpurun_out/r3f/test_gemm_stamps.log | awk '{print $3,$4,$5,$7,$8,$9,$10,$11}' | sort | uniq -c | sort -k2 | awk '{a[$2" "$3" "$4" "$5]=a[$2" "$3" "$4" "$5]" "$6}END{for(k in a)print k, a[k]}' | cut -c1-200
python -m pytest tests -m gpu -x -q > gpurun_out/r3f/gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3f/gputest.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3f/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','loss')}, d['roofline']['frac'], d['roofline']['gemm_family'])
for s in d['roofline']['per_shape'][:14]: print(s['kernel'][20:], s['M'], s['N'], s['K'], s['launches'], s['avg_us'], s['tflops'])
PY
