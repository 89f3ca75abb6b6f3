# rocprofv3 PMC passes over the bench (separate passes; kernel-trace only, no sys/hip trace — gpurun refuses those with --pmc) -> gpurun_out/pmc_*
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_mfma.log 2>&1
cd $R; ls gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma; python tools/pmc_summary.py 2>&1 | tail -30; cp profiles/r03_pmc_*.json gpurun_out/
