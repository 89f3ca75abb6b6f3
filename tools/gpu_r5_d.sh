# round 5: new gemm4w tests + the encoder suite + bench with the four-wave kernel as the default route
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_encoder.py -x -q 2>&1 | tail -4
for v in 0 1; do PCLIP_GEMM_4W=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH 4W=$v', round(d['value']), d['ms_per_step'], d['self_check'], d['roofline']['frac'], d['roofline'].get('kernel'))"; done
