mkdir -p gpurun_out
(timeout 600 python tools/encoder_bench.py 2>&1 | grep -v amdgpu.ids | tail -12) | tee gpurun_out/encoder_bench_r3.log
(timeout 600 python tools/serving_latency.py 2>&1 | grep -v amdgpu.ids | tail -14) | tee gpurun_out/serving_r3.log
(timeout 300 python tools/train_bench.py 2>&1 | grep -v amdgpu.ids | tail -6) | tee gpurun_out/train_bench_r3.log
(timeout 300 python tools/kernel_bench.py --what gemm,attn,ln --imgs 1024 2>&1 | grep -v amdgpu.ids | tail -24) | tee gpurun_out/kernel_bench_r3.log
(timeout 300 python tools/text_bench.py 2>&1 | grep -v amdgpu.ids | tail -4) | tee gpurun_out/text_bench_r3.log
