set -o pipefail
mkdir -p gpurun_out/r3m
python -m pytest tests -m gpu -x -q > gpurun_out/r3m/gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3m/gputest.log
python bench.py > gpurun_out/r3m/bench_default.json 2> gpurun_out/r3m/bench_default.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3m/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','loss','rccl_world')}, d['roofline']['frac'], d['roofline']['gemm_family'], d['multi_stream'], d['cpu_baseline'])"
bash tools/prof_step.sh > gpurun_out/r3m/prof_step.log 2>&1; echo "prof rc=$?"; tail -3 gpurun_out/r3m/prof_step.log
bash tools/pmc_traffic.sh > gpurun_out/r3m/pmc_traffic.log 2>&1; echo "traffic rc=$?"; tail -25 gpurun_out/r3m/pmc_traffic.log
python bench.py --config clipseg --steps 10 --warmup 3 > gpurun_out/r3m/bench_clipseg.json 2> gpurun_out/r3m/bench_clipseg.err; echo "clipseg rc=$?"; tail -c 600 gpurun_out/r3m/bench_clipseg.json
python bench.py --config vitl_lora --steps 5 --warmup 2 > gpurun_out/r3m/bench_vitl.json 2> gpurun_out/r3m/bench_vitl.err; echo "vitl rc=$?"; tail -c 600 gpurun_out/r3m/bench_vitl.json
