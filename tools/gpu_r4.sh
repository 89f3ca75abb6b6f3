# Round-4 evidence run (one gpurun call): full GPU suite with the tolerance ledger, the bench line, rocprofv3 kernel stats of the same command,
# PMC passes (separate --pmc runs, kernel-trace only), tool benches.  Argument: tag.  Everything lands in gpurun_out/ (copy what is cited into profiles/).
TAG=${1:-v1}
mkdir -p gpurun_out
PCLIP_OBSERVED_JSON=1 timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --tb=short 2>&1 | grep -vE "^E   +(\+|where)" > gpurun_out/r04_pytest_gpu_$TAG.log
grep -v "of the bound" gpurun_out/r04_pytest_gpu_$TAG.log | tail -12
cp gpurun_out/observed_tolerances.json gpurun_out/r04_observed_tolerances_$TAG.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_$TAG.json 2> gpurun_out/r04_bench_$TAG.err
python -c "import sys,json; d=json.load(open('gpurun_out/r04_bench_$TAG.json')); print('BENCH', round(d['value']), d['ms_per_step'], d['self_check'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['extra'].get('folded_value'))"
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04_$TAG -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_r04_$TAG.log 2>&1
cp $R/gpurun_out/prof_r04_$TAG/bench_kernel_stats.csv $R/gpurun_out/r04_bench_${TAG}_kernel_stats.csv 2>/dev/null
head -14 $R/gpurun_out/r04_bench_${TAG}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-180
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/pmc_mfma.log 2>&1
cd $R; python tools/pmc_summary.py 2>&1 | tail -25; cp profiles/r04_pmc_traffic.json profiles/r04_pmc_mfma.json gpurun_out/ 2>/dev/null; rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/prof_r04_$TAG
( echo "== adapter_bench (MFMA kernels)"; python tools/adapter_bench.py; echo "== adapter_bench PCLIP_ADAPTER_MFMA=0 (VALU kernels)"; PCLIP_ADAPTER_MFMA=0 python tools/adapter_bench.py;
  echo "== train_bench (MFMA adapter backward)"; python tools/train_bench.py; echo "== train_bench PCLIP_ADAPTER_MFMA=0"; PCLIP_ADAPTER_MFMA=0 python tools/train_bench.py;
  echo "== small_bench"; python tools/small_bench.py; echo "== encoder_bench"; python tools/encoder_bench.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_tool_benches_$TAG.txt
tail -30 gpurun_out/r04_tool_benches_$TAG.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r04_bench_torchrun_$TAG.json 2> gpurun_out/r04_bench_torchrun_$TAG.err; python -c "import json; d=json.load(open('gpurun_out/r04_bench_torchrun_$TAG.json')); print('TORCHRUN', round(d['value']), d['rccl_world_size'], d['self_check'])"
