# round 5, call A: the four-wave asm-loop GEMM — bit-identity, race-stress build, timing against the eight-wave kernel, bench with / without it
mkdir -p gpurun_out
timeout 600 python tools/gemm4w_check.py > gpurun_out/r05_gemm4w_check.txt 2>&1; echo "check rc=$?"
grep -v "^check" gpurun_out/r05_gemm4w_check.txt | tail -12; grep -c "identical=True" gpurun_out/r05_gemm4w_check.txt; grep "identical=False" gpurun_out/r05_gemm4w_check.txt | head -5
PCLIP_RACE_STRESS=1 timeout 600 python tools/gemm4w_check.py --no-bench > gpurun_out/r05_gemm4w_stress.txt 2>&1; echo "stress rc=$?"; tail -2 gpurun_out/r05_gemm4w_stress.txt
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -k "gemm or linear" 2>&1 | tail -3
for v in 0 1 0 1; do PCLIP_GEMM_4W=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH 4W=$v', round(d['value']), d['ms_per_step'], d['self_check'], d['roofline']['frac'])"; done
