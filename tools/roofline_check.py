#!/usr/bin/env python3
"""Does the bench line's `roofline` follow from the rocprofv3 summary of the same command?  usage: roofline_check.py <bench.json> <kernel_stats.csv>
Compares, per GEMM variant, the bench's avg_call_us (the GEMM kernels' own begin / end stamps inside an instrumented step: pclip_gemm_timing) with rocprofv3's average duration of that kernel, and
the step totals (sum of every GEMM kernel in the trace / steps traced)."""
import csv, json, sys
line = json.load(open(sys.argv[1]))
rows = list(csv.DictReader(open(sys.argv[2])))
rf = line["roofline"]
pv = rf.get("per_variant") or {}
def stat(substr):
    sel = [r for r in rows if substr in r["Name"]]
    return sum(int(r["Calls"]) for r in sel), sum(float(r["TotalDurationNs"]) for r in sel)
# linear4w_kernel<ACT, HAS_BIAS, VAR>: ACT 1 = c_fc, ACT 0 + bias = in_proj, ACT 6 = out_proj + c_proj together
names = {"c_fc": "linear4w_kernelILi1ELb1E", "in_proj": "linear4w_kernelILi0ELb1E", "residual (out_proj + c_proj)": "linear4w_kernelILi6ELb1E"}
for nm, sub in names.items():
    calls, tot = stat(sub)
    if not calls: continue
    if nm.startswith("residual"):
        b_calls = sum(pv[k]["calls"] for k in ("out_proj", "c_proj") if k in pv)
        b_ms = sum(pv[k]["ms_per_step"] for k in ("out_proj", "c_proj") if k in pv)
        b_us = 1e3 * b_ms / max(b_calls, 1)
    else:
        b_us = pv.get(nm, {}).get("avg_call_us", float("nan"))
    print(f"{nm:30s} rocprofv3 avg {tot / calls / 1e3:8.1f} us ({calls} launches)   bench avg_call_us {b_us:8.1f}   ratio {b_us / (tot / calls / 1e3):.3f}")
gemm_ns = sum(float(r["TotalDurationNs"]) for r in rows if "linear4w_kernel" in r["Name"] or "linear_fast_kernel" in r["Name"] or "linear_small_kernel" in r["Name"] or "splitk" in r["Name"])
att = [r for r in rows if "attention_kernelILi8" in r["Name"]]
steps = round(sum(int(r["Calls"]) for r in att) / 11) if att else 0
if steps:
    print(f"GEMM kernels, rocprofv3 sum / {steps} steps = {gemm_ns / steps / 1e6:.2f} ms per step   bench gemm_ms_per_step {rf['gemm_ms_per_step']:.2f}   ratio {rf['gemm_ms_per_step'] / (gemm_ns / steps / 1e6):.3f}")
    print(f"=> roofline.achieved from the trace: {rf['algorithmic_gflop_per_step'] / (gemm_ns / steps / 1e6):.0f} TFLOP/s; the line says {rf['achieved']:.0f} (frac {rf['frac']:.3f})")
