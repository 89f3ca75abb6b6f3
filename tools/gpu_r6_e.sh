# Round-6 call E: per-tile second pass of the fused classification (tests + ImageNet timing), four-wave GEMM: tile-phase stamps, band tile order A/B + FETCH_SIZE
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -q --timeout 600 --tb=short -k "fused or full_size or fuzz or jitter" 2>&1 | grep -vE "of the bound" | tail -12
python tools/small_bench.py 2>&1 | grep -v amdgpu.ids | tail -5
python tools/gemm4w_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_gemm4w_stamps.txt
python tools/ab_band4w.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_ab_band4w.txt
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for b in 0 4 6; do
  PMC_BAND=$b timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_band_$b -o b -- python $R/tools/ab_band4w.py pmc > /dev/null 2>&1
  F=$(find $R/gpurun_out/pmc_band_$b -name "*counter_collection.csv" | head -1); echo "band $b"; python $R/tools/pmc_kernels.py $F linear4w | cut -c1-160
  rm -rf $R/gpurun_out/pmc_band_$b
done 2>&1 | tee $R/gpurun_out/r06_band4w_fetch.txt
