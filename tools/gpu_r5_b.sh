# round 5, quick iteration: four-wave GEMM identity (+ stress build) + variants timing in one process
mkdir -p gpurun_out
timeout 600 python tools/gemm4w_check.py --no-bench > gpurun_out/r05_gemm4w_check.txt 2>&1; echo "check rc=$?"
grep -v "^check" gpurun_out/r05_gemm4w_check.txt | tail -3; grep -c "identical=True" gpurun_out/r05_gemm4w_check.txt; grep "identical=False" gpurun_out/r05_gemm4w_check.txt | head -5
PCLIP_RACE_STRESS=1 timeout 600 python tools/gemm4w_check.py --no-bench 2>&1 | tail -1
timeout 900 python tools/gemm4w_check.py --variants ${1:-2,3,4,5} --rounds 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_gemm4w_variants.txt
