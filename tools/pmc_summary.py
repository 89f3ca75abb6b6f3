#!/usr/bin/env python3
"""Builds profiles/r01_pmc_traffic.json from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes that
tools/gpu_round.sh leaves under gpurun_out/ (units KB; FETCH_SIZE doubled: gfx950 reports half of wide coalesced
reads, MI355X_MICROARCH.md §HBM).  Per-launch averages per kernel + the launch-weighted mean over the GEMM kernels."""
import collections, csv, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def per_kernel(path, counter):
    tot, n, seen = collections.defaultdict(float), collections.Counter(), set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}

f = per_kernel(os.path.join(root, "gpurun_out/pmc_fetch/b_counter_collection.csv"), "FETCH_SIZE")
w = per_kernel(os.path.join(root, "gpurun_out/pmc_write/b_counter_collection.csv"), "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python bench.py --steps 2 --warmup 1 "
               "--no-cpu-baseline`; units KB; FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, "
               "MI355X_MICROARCH.md §HBM); per launch averages", "kernels": {}}
gb, gn = 0.0, 0
for k in sorted(f):
    if k not in w: continue
    hbm = (2.0 * f[k][0] + w[k][0]) * 1024.0
    out["kernels"][k] = {"launches": f[k][1], "fetch_kb_raw": f[k][0], "write_kb": w[k][0], "hbm_bytes_per_launch": hbm}
    if "linear_fast_kernel" in k:
        gb += hbm * f[k][1]; gn += f[k][1]
out["linear_kernel_hbm_bytes_per_launch"] = gb / max(gn, 1)
json.dump(out, open(os.path.join(root, "profiles/r01_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k[:90]: v for k, v in out["kernels"].items() if "linear" in k or "attention" in k or "layernorm" in k}, indent=1)[:3000])
print("GEMM mean bytes/launch", out["linear_kernel_hbm_bytes_per_launch"])
