mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --tb=short 2>&1 | grep -vE "^E   +(\+|where)" | tail -15 > gpurun_out/pytest_gpu.log; tail -5 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_r1.log 2>&1
head -12 $R/gpurun_out/prof_r1/bench_kernel_stats.csv | cut -c1-160
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_mfma.log 2>&1
cd $R; python - <<'PY'
import csv, collections
def agg(path, names):
    rows = csv.DictReader(open(path))
    a = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); disp=set()
    for r in rows:
        k = r["Kernel_Name"][:60]
        a[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in disp: disp.add((k, r["Dispatch_Id"])); n[k]+=1
    for k in a:
        if any(x in k for x in names): print(k, n[k], {c: v/n[k] for c, v in a[k].items()})
for d in ("pmc_fetch","pmc_write","pmc_mfma"):
    try: agg(f"gpurun_out/{d}/b_counter_collection.csv", ["linear_fast", "attention", "sqdist", "layernorm"])
    except Exception as e: print(d, "failed", e)
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/bench_torchrun.json 2> gpurun_out/bench_torchrun.err; tail -c 600 gpurun_out/bench_torchrun.json
