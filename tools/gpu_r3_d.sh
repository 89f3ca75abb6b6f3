mkdir -p gpurun_out
timeout 600 python tools/ab_multi.py attn base av1 av3 av6 av7 av8 av15 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_attn_var.log
timeout 600 python tools/ab_multi.py gemm base gp1 gp2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_gemm_prio.log
for c in 256 512; do PCLIP_VIT_CHUNK=$c timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CHUNK $c', d['value'], d['ms_per_step'])"; done | tee gpurun_out/chunk_bench.log
