mkdir -p gpurun_out/r3o
python -m pytest tests -m gpu -x -q > gpurun_out/r3o/gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3o/gputest.log
python tools/parity_at_bench_batch.py --batches 256 --adapters init --out gpurun_out/r3o/parity_init.json > gpurun_out/r3o/parity_init.log 2>&1; echo rc=$?
grep "^{" gpurun_out/r3o/parity_init.log | cut -c1-700
python tools/parity_at_bench_batch.py --batches 64,256 --stress --out gpurun_out/r3o/parity_scaled.json > gpurun_out/r3o/parity_scaled.log 2>&1; echo rc=$?
grep "^{" gpurun_out/r3o/parity_scaled.log | cut -c1-700
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also-streams 0 > gpurun_out/r3o/bench.json 2> gpurun_out/r3o/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3o/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','loss')}, d['roofline']['traffic'], d['roofline']['traffic_source'][:40])"
