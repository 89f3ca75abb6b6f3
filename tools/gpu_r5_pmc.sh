# round 5: SQ counters of the eight-wave and the four-wave GEMM kernels on the bench's shapes (kernel-trace only beside --pmc)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_g4 -o b -- python $R/tools/gemm4w_pmc.py > $R/gpurun_out/pmc_g4.log 2>&1
cd $R; F=$(find gpurun_out/pmc_g4 -name "*counter_collection.csv" | head -1); python tools/pmc_kernels.py $F linear gpurun_out/r05_pmc_gemm4w.json | cut -c1-400; rm -rf gpurun_out/pmc_g4
