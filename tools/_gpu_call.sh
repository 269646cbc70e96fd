mkdir -p gpurun_out/r3s
python -m pytest tests -m gpu -x -q > gpurun_out/r3s/gputest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r3s/gputest.log
for f in "" "--no-half-height-short-k" "" "--no-half-height-short-k"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also-streams 0 $f > gpurun_out/r3s/bench.json 2> gpurun_out/r3s/bench.err; echo "bench [$f] rc=$?"
  python -c "
import json
d=json.loads(open('gpurun_out/r3s/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','loss')}, d['roofline']['gemm_family']['ms_per_step'])
for s in d['roofline']['per_shape']:
    if s['N']==768 and s['K']==768: print('   ', s['kernel'][20:], s['M'], s['launches'], s['avg_us'])"
done
