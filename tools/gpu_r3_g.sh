mkdir -p gpurun_out
timeout 600 python tools/ab_band.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_band.log
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for b in 0 6 3; do
  PCLIP_GEMM_BAND=$b timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_band$b -o b -- python $R/tools/ab_band.py once > $R/gpurun_out/pmc_band$b.log 2>&1
  python - $b <<'PY'
import csv, sys, os, collections
b = sys.argv[1]; root = os.environ["GRAFT_REPO_ROOT"]
tot, n = collections.defaultdict(float), collections.Counter()
seen = set()
for r in csv.DictReader(open(f"{root}/gpurun_out/pmc_band{b}/b_counter_collection.csv")):
    if r["Counter_Name"] != "FETCH_SIZE" or "linear_fast" not in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"][-60:], r["Grid_Size"]) if "Grid_Size" in r else r["Kernel_Name"][-60:]
    tot[key] += float(r["Counter_Value"])
    if (key, r["Dispatch_Id"]) not in seen: seen.add((key, r["Dispatch_Id"])); n[key] += 1
for k in tot: print(f"band {b}: {k}: launches {n[k]}, FETCH_SIZE x2 = {2 * tot[k] / n[k] * 1024 / 1e9:.3f} GB per launch")
PY
done 2>&1 | tee $R/gpurun_out/pmc_band.log
