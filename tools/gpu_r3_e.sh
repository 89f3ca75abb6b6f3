# tests -> bench -> rocprof kernel stats -> PMC passes (argument: tag)
TAG=${1:-r3e}
export PCLIP_OBSERVED_JSON=1
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 --tb=short 2>&1 | grep -vE "^E   +(\+|where)" > gpurun_out/pytest_gpu_$TAG.log; grep -v "of the bound" gpurun_out/pytest_gpu_$TAG.log | tail -25
cp gpurun_out/observed_tolerances.json gpurun_out/observed_tolerances_$TAG.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; python -c "import json; d=json.load(open('gpurun_out/bench_$TAG.json')); print('BENCH', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'], d['sclk_mhz_under_load'], d['power_w'])"; tail -2 gpurun_out/bench_$TAG.err
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
head -16 $R/gpurun_out/prof_$TAG/bench_kernel_stats.csv | cut -c1-220
