mkdir -p gpurun_out
timeout 600 python tools/ab_multi.py ln ln0 base 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_ln2.log
timeout 600 python tools/ab_multi.py gemm base rpf 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_res_pf.log
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q --timeout 600 --tb=short -k "attention or layernorm or towers or full_size" 2>&1 | grep -v "of the bound" | tail -12
