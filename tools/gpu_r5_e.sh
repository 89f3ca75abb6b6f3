# round 5: fused row-panel classification — parity tests + timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q 2>&1 | grep -v "of the bound" | tail -12
timeout 600 python tools/small_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_small_bench.txt | tail -6
