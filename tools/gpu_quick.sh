# quick GPU pass: selected tests (arg 2, default all) -> bench; arguments: tag [pytest-selection]
TAG=${1:-q}; SEL=${2:-tests}
mkdir -p gpurun_out
timeout 1700 python -m pytest $SEL -m gpu -q --timeout 900 --tb=short 2>&1 | grep -vE "^E   +(\+|where)" > gpurun_out/pytest_gpu_$TAG.log; grep -v "of the bound" gpurun_out/pytest_gpu_$TAG.log | tail -60
cp gpurun_out/observed_tolerances.json gpurun_out/observed_tolerances_$TAG.json 2>/dev/null
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'])"; tail -3 gpurun_out/bench_$TAG.err
