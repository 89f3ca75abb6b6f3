"""Per-launch averages of the PMC passes of tools/gpu_att_pmc.sh (attention kernels only)."""
import csv, glob, collections, json, os
out = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/attpmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attention" not in k:
            continue
        key = ("B16" if "ILi8E" in k else "text", r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"]); acc[key][1] += 1
    for (kern, ctr), (s, n) in acc.items():
        out[kern][ctr] = s / n
for kern, d in out.items():
    print(kern)
    for c, v in sorted(d.items()):
        print(f"  {c:34s} {v:16.0f}")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/att_pmc.json", "w"), indent=1)
