# Round-6 call B: the mid-N one-launch classification (tests + small_bench) and the full-size default-routing test
mkdir -p gpurun_out
python tools/mid_probe.py 0 1 2>&1 | grep -v amdgpu
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 --tb=short -x -k "full_size_default or mid or fuzz or alpha_outside or reference_fixture or run_proto_clip or zero_shot" 2>&1 | grep -vE "of the bound" | tail -40 > gpurun_out/r06_b_pytest.log; tail -12 gpurun_out/r06_b_pytest.log
python tools/small_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_small_bench_v1.txt; cat gpurun_out/r06_small_bench_v1.txt
