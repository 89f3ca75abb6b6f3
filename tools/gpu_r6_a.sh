# Round-6 call A: the new full-size routing tests, a baseline bench line of this round's box, and the ResNet-50 evidence: rocprofv3 --stats and
# SQ / traffic counters of tools/rn_prof.py (RN50, batch 256, three encode_image passes).
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 --tb=short -k "full_size or alpha_outside or fused" 2>&1 | tail -8 > gpurun_out/r06_a_pytest.log; tail -5 gpurun_out/r06_a_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_v0.json 2> gpurun_out/r06_bench_v0.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r06_bench_v0.json')); print('BENCH', d['value'] and round(d['value']), d['ms_per_step'], d['self_check'], d['roofline']['achieved'], d['roofline']['frac'], d.get('sclk_mhz_under_load'))"
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rn50 -o rn -- python $R/tools/rn_prof.py > $R/gpurun_out/prof_rn50.log 2>&1
cp $R/gpurun_out/prof_rn50/rn_kernel_stats.csv $R/gpurun_out/r06_rn50_kernel_stats.csv 2>/dev/null
head -16 $R/gpurun_out/r06_rn50_kernel_stats.csv | cut -d, -f1-5 | cut -c1-200
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_rn -o b -- python $R/tools/rn_prof.py > $R/gpurun_out/pmc_rn.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_rn_f -o b -- python $R/tools/rn_prof.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_rn_w -o b -- python $R/tools/rn_prof.py > /dev/null 2>&1
cd $R
F=$(find gpurun_out/pmc_rn -name "*counter_collection.csv" | head -1); python tools/pmc_kernels.py $F "" gpurun_out/r06_pmc_rn50_sq.json | cut -c1-300 | head -30
for d in f w; do F=$(find gpurun_out/pmc_rn_$d -name "*counter_collection.csv" | head -1); python tools/pmc_kernels.py $F "" gpurun_out/r06_pmc_rn50_$d.json | cut -c1-200 | head -24; done
rm -rf gpurun_out/pmc_rn gpurun_out/pmc_rn_f gpurun_out/pmc_rn_w gpurun_out/prof_rn50
( echo "== encoder_bench"; python tools/encoder_bench.py; echo "== small_bench"; python tools/small_bench.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_tool_benches_v0.txt; tail -30 gpurun_out/r06_tool_benches_v0.txt
