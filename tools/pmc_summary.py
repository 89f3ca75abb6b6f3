#!/usr/bin/env python3
"""Builds profiles/<tag>_pmc_traffic.json (tag = argv[1], default r05) from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes that
tools/gpu_round.sh leaves under gpurun_out/ (units KB; FETCH_SIZE doubled: gfx950 reports half of wide coalesced
reads, MI355X_MICROARCH.md §HBM).  Per-launch averages per kernel + the launch-weighted mean over the GEMM kernels."""
import collections, csv, json, os, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
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
    if "linear_fast_kernel" in k or "linear_small_kernel" in k or "linear4w_kernel" in k:
        gb += hbm * f[k][1]; gn += f[k][1]
out["linear_kernel_hbm_bytes_per_launch"] = gb / max(gn, 1)
json.dump(out, open(os.path.join(root, f"profiles/{TAG}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k[:90]: v for k, v in out["kernels"].items() if "linear" in k or "attention" in k or "layernorm" in k}, indent=1)[:3000])
print("GEMM mean bytes/launch", out["linear_kernel_hbm_bytes_per_launch"])

# ---- MFMA utilisation / LDS conflicts / wait fraction from the third pass (SQ_VALU_MFMA_BUSY_CYCLES counts SIMD cycles:
# 16 per v_mfma_f32_16x16x32_f16, 32 per 32x32x16; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs)
mpath = os.path.join(root, "gpurun_out/pmc_mfma/b_counter_collection.csv")
if os.path.exists(mpath):
    tot, n, seen = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter(), set()
    for r in csv.DictReader(open(mpath)):
        k = r["Kernel_Name"]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1
    res = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT "
                   "SQ_LDS_IDX_ACTIVE over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline`; mfma_util = MFMA busy SIMD-cycles / "
                   "(1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), i.e. the fraction of the dense MFMA rate AT THE CLOCK THE KERNEL RAN AT "
                   "(the GEMM runs power-throttled, DESIGN §5); per-launch averages", "kernels": {}}
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        g = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if g <= 0 or n[k] == 0: continue
        res["kernels"][k] = {"launches": n[k], "gpu_cycles_per_launch": g / n[k],
                             "mfma_util": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * g),
                             "lds_bank_conflict_frac": v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0),
                             "wave_wait_frac": v.get("SQ_WAIT_ANY", 0.0) / max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)}
    json.dump(res, open(os.path.join(root, f"profiles/{TAG}_pmc_mfma.json"), "w"), indent=1)
    for k, v in list(res["kernels"].items())[:6]:
        print(f"{k[:80]:80s} mfma_util {v['mfma_util']:.3f} lds_conflict {v['lds_bank_conflict_frac']:.3f} wait {v['wave_wait_frac']:.3f}")
