# Round-6 evidence run (one gpurun call): full GPU suite with the tolerance ledger, the bench line, rocprofv3 kernel stats of the same command, tool benches,
# and LAST — on the tree that was just timed — the PMC passes (separate --pmc runs, kernel-trace only).  Argument: tag.  Everything lands in gpurun_out/.
TAG=${1:-v4}
mkdir -p gpurun_out
PCLIP_OBSERVED_JSON=1 timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --tb=short 2>&1 | grep -vE "^E   +(\+|where)" > gpurun_out/r06_pytest_gpu_$TAG.log
grep -v "of the bound" gpurun_out/r06_pytest_gpu_$TAG.log | tail -12
cp gpurun_out/observed_tolerances.json gpurun_out/r06_observed_tolerances_$TAG.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_$TAG.json 2> gpurun_out/r06_bench_$TAG.err; echo "bench rc=$?"
python -c "import sys,json; d=json.load(open('gpurun_out/r06_bench_$TAG.json')); print('BENCH', d['value'] and round(d['value']), d['ms_per_step'], d['median_ms'], d['p10_ms'], d['p90_ms'], d['self_check'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['roofline'].get('per_variant'), d['extra'].get('c3_classify'), d['extra'].get('c2_kernels'))"
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06_$TAG -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_r06_$TAG.log 2>&1
cp $R/gpurun_out/prof_r06_$TAG/bench_kernel_stats.csv $R/gpurun_out/r06_bench_${TAG}_kernel_stats.csv 2>/dev/null
head -14 $R/gpurun_out/r06_bench_${TAG}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-180
python $R/tools/roofline_check.py $R/gpurun_out/r06_bench_$TAG.json $R/gpurun_out/r06_bench_${TAG}_kernel_stats.csv | tee $R/gpurun_out/r06_roofline_check_$TAG.txt
cd $R
( echo "== small_bench"; python tools/small_bench.py; echo "== encoder_bench"; python tools/encoder_bench.py; echo "== adapter_bench"; python tools/adapter_bench.py; echo "== train_bench"; python tools/train_bench.py; echo "== conv_strip_bench"; python tools/conv_strip_bench.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_tool_benches_$TAG.txt
tail -12 gpurun_out/r06_tool_benches_$TAG.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r06_bench_torchrun_$TAG.json 2> gpurun_out/r06_bench_torchrun_$TAG.err; python -c "import json; d=json.load(open('gpurun_out/r06_bench_torchrun_$TAG.json')); print('TORCHRUN', round(d['value']), d['rccl_world_size'], d['self_check'])"
# ---- PMC passes LAST, same tree
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/pmc_mfma.log 2>&1
# the fused classification's HBM traffic (ImageNet split): FETCH / WRITE of classify_panel_kernel vs the two stages
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_cls_f -o b -- python $R/tools/classify_pmc.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_cls_w -o b -- python $R/tools/classify_pmc.py > /dev/null 2>&1
cd $R; python tools/pmc_summary.py r06 2>&1 | tail -16; cp profiles/r06_pmc_traffic.json profiles/r06_pmc_mfma.json gpurun_out/ 2>/dev/null
for d in f w; do F=$(find gpurun_out/pmc_cls_$d -name "*counter_collection.csv" | head -1); python tools/pmc_kernels.py $F "" gpurun_out/r06_pmc_classify_$d.json | grep -A1 -E "classify_panel|sqdist|fuse_probs" | cut -c1-200; done
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/prof_r06_$TAG gpurun_out/pmc_cls_f gpurun_out/pmc_cls_w
