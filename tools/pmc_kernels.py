#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 counter_collection.csv + the derived figures used in DESIGN.md:
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES; issue_stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES;
active = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES.  usage: pmc_kernels.py <csv> [substring filter] [out.json]"""
import collections, csv, json, sys

path, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
tot, n, seen = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter(), set()
grid = {}
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if filt and filt not in k: continue
    key = k + " grid=" + r.get("Grid_Size", "?")
    tot[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if (key, r["Dispatch_Id"]) not in seen:
        seen.add((key, r["Dispatch_Id"])); n[key] += 1
out = {}
for k, v in tot.items():
    d = {c: x / n[k] for c, x in v.items()}
    g = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if g > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in d: d["mfma_util"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * g)
    if wc > 0:
        for name, c in (("wait", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"), ("active", "SQ_ACTIVE_INST_ANY"), ("lds_stall", "SQ_WAIT_INST_LDS")):
            if c in d: d[name] = d[c] / wc
    d["launches"] = n[k]
    out[k] = d
for k, d in sorted(out.items()):
    print(k[:110])
    print("   " + "  ".join(f"{c}={x:.4g}" for c, x in sorted(d.items())))
if len(sys.argv) > 3: json.dump(out, open(sys.argv[3], "w"), indent=1)
