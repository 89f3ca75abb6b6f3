# round-2 GPU pass: tests -> bench -> rocprof kernel stats (arguments: tag)
TAG=${1:-a}
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 --tb=short -x 2>&1 | grep -vE "^E   +(\+|where)" | tail -150 > gpurun_out/pytest_gpu_$TAG.log; tail -8 gpurun_out/pytest_gpu_$TAG.log
cp gpurun_out/observed_tolerances.json gpurun_out/observed_tolerances_$TAG.json 2>/dev/null
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
head -14 $R/gpurun_out/prof_$TAG/bench_kernel_stats.csv | cut -c1-170
cd $R; timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline | tail -c 400
