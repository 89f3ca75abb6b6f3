# round-3 GPU pass C: autograd drop-ins + the re-gated e2e suite (LayerNorm fold opt-in) + full suite + bench
export PCLIP_OBSERVED_JSON=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_autograd.py tests/test_gpu_e2e.py tests/test_gpu_parity.py -m gpu -q --timeout 600 --tb=short -x 2>&1 | grep -vE "^E   +(\+|where)" > gpurun_out/pytest_gpu_r3c1.log; grep -v "of the bound" gpurun_out/pytest_gpu_r3c1.log | tail -40
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 --tb=short 2>&1 | grep -vE "^E   +(\+|where)" > gpurun_out/pytest_gpu_r3c.log; grep -v "of the bound" gpurun_out/pytest_gpu_r3c.log | tail -15
cp gpurun_out/observed_tolerances.json gpurun_out/observed_tolerances_r3c.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r3c.json 2> gpurun_out/bench_r3c.err; python -c "import json; d=json.load(open('gpurun_out/bench_r3c.json')); print('BENCH unfolded', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['sclk_mhz_under_load'], d['power_w'])"; tail -2 gpurun_out/bench_r3c.err
PCLIP_LN_FOLD=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r3c_fold.json 2> gpurun_out/bench_r3c_fold.err; python -c "import json; d=json.load(open('gpurun_out/bench_r3c_fold.json')); print('BENCH folded', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['sclk_mhz_under_load'], d['power_w'])"
