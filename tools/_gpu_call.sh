set -o pipefail
mkdir -p gpurun_out/r3t
python -m pytest tests -m gpu -x -q > gpurun_out/r3t/gputest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r3t/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3t/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3t/smoke.log
python bench.py > gpurun_out/r3t/bench_default.json 2> gpurun_out/r3t/bench_default.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3t/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','loss','rccl_world')}, d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['traffic'], d['roofline']['gemm_family'], d['multi_stream'], d['cpu_baseline']['value'])"
bash tools/prof_step.sh > gpurun_out/r3t/prof_step.log 2>&1; echo "prof rc=$?"
cp gpurun_out/prof/step_kernel_stats.csv gpurun_out/r3t/kernel_stats.csv; cp gpurun_out/prof/bench.log gpurun_out/r3t/prof_bench.log
bash tools/pmc_step.sh > gpurun_out/r3t/pmc_step.log 2>&1; echo "pmc rc=$?"; tail -16 gpurun_out/r3t/pmc_step.log
cp gpurun_out/pmc_step/summary.json gpurun_out/r3t/pmc_summary.json
