set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
echo "== torchrun world=1 sanity"; timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-entry-point --also-streams 0 2> gpurun_out/r5p/torchrun.err | tail -1 | cut -c1-300
echo "== kernel trace (serial step)"; timeout -k 10 300 bash tools/prof_step.sh > gpurun_out/r5p/prof_step.log 2>&1; tail -3 gpurun_out/r5p/prof_step.log
echo "== pmc step"; timeout -k 10 600 bash tools/pmc_step.sh > gpurun_out/r5p/pmc_step.log 2>&1; tail -30 gpurun_out/r5p/pmc_step.log
echo "== pmc traffic"; timeout -k 10 400 bash tools/pmc_traffic.sh > gpurun_out/r5p/pmc_traffic.log 2>&1; tail -12 gpurun_out/r5p/pmc_traffic.log
