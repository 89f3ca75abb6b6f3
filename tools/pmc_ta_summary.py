#!/usr/bin/env python3
"""Per-kernel, per-launch averages of the TA / TCP / SQ-VMEM counters tools/gpu_pmc_ta.sh collected (gpurun_out/pmc_ta_*), with the ratios the
address-path argument of DESIGN section 5 items 13 / 16 rests on -> profiles/r03_pmc_addr_path.json."""
import collections, csv, glob, json, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tot, n = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(collections.Counter)
for path in sorted(glob.glob(os.path.join(root, "gpurun_out/pmc_ta_*/b_counter_collection.csv"))):
    tag = os.path.basename(os.path.dirname(path))
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        c = r["Counter_Name"] if r["Counter_Name"] != "GRBM_GUI_ACTIVE" else "GRBM_GUI_ACTIVE@" + tag
        tot[k][c] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"], c) not in seen:
            seen.add((k, r["Dispatch_Id"], c)); n[k][c] += 1
out = {"note": "rocprofv3 --pmc passes (tools/gpu_pmc_ta.sh) over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline`; per-launch averages; *_sum counters are "
               "summed over the chip's TA / TCP instances (one per CU: 256), SQ counters over the SQs; gpu_cycles = GRBM_GUI_ACTIVE / 8 XCDs", "kernels": {}}
for k, v in tot.items():
    if not any(s in k for s in ("linear_fast", "attention_kernel", "layernorm_pf", "vit_embed", "im2col")): continue
    a = {c: v[c] / max(n[k][c], 1) for c in v}
    g = next((a[c] for c in a if c.startswith("GRBM_GUI_ACTIVE@")), 0.0) / 8.0
    if g <= 0: continue
    e = {"launches": max(n[k].values()), "gpu_cycles": g}
    e.update({c: a[c] for c in a if not c.startswith("GRBM")})
    if "TA_TA_BUSY_sum" in a: e["ta_busy_frac"] = a["TA_TA_BUSY_sum"] / (256.0 * g)
    if "TA_BUFFER_TOTAL_CYCLES_sum" in a: e["ta_buffer_cycles_frac"] = a["TA_BUFFER_TOTAL_CYCLES_sum"] / (256.0 * g)
    if "TA_ADDR_STALLED_BY_TC_CYCLES_sum" in a: e["ta_addr_stalled_by_tc_frac"] = a["TA_ADDR_STALLED_BY_TC_CYCLES_sum"] / (256.0 * g)
    if "TA_DATA_STALLED_BY_TC_CYCLES_sum" in a: e["ta_data_stalled_by_tc_frac"] = a["TA_DATA_STALLED_BY_TC_CYCLES_sum"] / (256.0 * g)
    if "TCP_PENDING_STALL_CYCLES_sum" in a: e["tcp_pending_stall_frac"] = a["TCP_PENDING_STALL_CYCLES_sum"] / (256.0 * g)
    if "TCP_TCP_TA_ADDR_STALL_CYCLES_sum" in a: e["tcp_ta_addr_stall_frac"] = a["TCP_TCP_TA_ADDR_STALL_CYCLES_sum"] / (256.0 * g)
    if "TCP_TCC_READ_REQ_LATENCY_sum" in a and a.get("TCP_TCC_READ_REQ_sum"): e["tcc_read_latency_cycles"] = a["TCP_TCC_READ_REQ_LATENCY_sum"] / a["TCP_TCC_READ_REQ_sum"]
    if "SQ_WAVE_CYCLES" in a and a["SQ_WAVE_CYCLES"]:
        e["vmem_issue_frac_of_wave_cycles"] = a.get("SQ_ACTIVE_INST_VMEM", 0.0) / a["SQ_WAVE_CYCLES"]
        e["wait_inst_any_frac"] = a.get("SQ_WAIT_INST_ANY", 0.0) / a["SQ_WAVE_CYCLES"]
    if a.get("SQ_INSTS_VMEM_RD"): e["cycles_per_vmem_rd_inst"] = a.get("SQ_INST_CYCLES_VMEM_RD", 0.0) / a["SQ_INSTS_VMEM_RD"]
    if a.get("SQ_INSTS_VMEM_WR"): e["cycles_per_vmem_wr_inst"] = a.get("SQ_INST_CYCLES_VMEM_WR", 0.0) / a["SQ_INSTS_VMEM_WR"]
    out["kernels"][k] = e
json.dump(out, open(os.path.join(root, "profiles/r03_pmc_addr_path.json"), "w"), indent=1)
for k, e in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["gpu_cycles"] * kv[1]["launches"])[:8]:
    print(k[:100])
    print("   ", {c: (round(x, 3) if isinstance(x, float) and x < 1000 else (int(x) if isinstance(x, float) else x)) for c, x in e.items()})
