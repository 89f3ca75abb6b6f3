# round 5: schedule variants of the four-wave K-loop, same process
mkdir -p gpurun_out
timeout 900 python tools/gemm4w_check.py --variants ${1:-2,3,4,5} --rounds 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_gemm4w_variants.txt
