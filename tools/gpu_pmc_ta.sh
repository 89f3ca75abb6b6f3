# rocprofv3 PMC passes over the bench for the vector-memory address path (TA / TCP) and the SQ's VMEM issue: separate passes, kernel-trace only -> gpurun_out/pmc_ta_*
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ta_$tag -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_ta_$tag.log 2>&1; ls $R/gpurun_out/pmc_ta_$tag 2>/dev/null | head -3; tail -2 $R/gpurun_out/pmc_ta_$tag.log | cut -c1-200; }
run sq SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run ta1 TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE
run ta2 TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run ta3 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run tcp1 TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
run tcp2 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
run tcp3 TCP_TOTAL_ACCESSES_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE
cd $R; python tools/pmc_ta_summary.py 2>&1 | tail -40
