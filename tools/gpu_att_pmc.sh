# PMC passes over the attention kernel alone (tools/att_pmc.py) -> gpurun_out/attpmc_<n>/ ; summary printed by tools/att_pmc_summary.py
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES" \
           "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/attpmc_$i -o a -- python $R/tools/att_pmc.py > $R/gpurun_out/attpmc_$i.log 2>&1
  echo "pass $i rc $?"
done
cd $R; python tools/att_pmc_summary.py
