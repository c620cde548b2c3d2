"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (name + grid), mean of every counter and duration."""
import csv, sys, collections, re

def short(name):
    m = re.search(r"modconv_mfma_kernel<([^>]*)>", name)
    if m: return "modconv<" + m.group(1).replace(" ", "") + ">"
    return name.split("(")[0].replace("gance::", "").replace("void ", "")[:40]

# Kernels launched several times per step with the same grid (the fused up kernel and the Winograd kernel: one launch
# per resolution) are split by their launch order within a step: PMC_SPLIT="name:launches_per_step;..." (default below).
import os
SPLIT = dict(item.rsplit(":", 1) for item in os.environ.get(
    "PMC_SPLIT", "upfir_fused_kernel:4;upfir_fused_pre_kernel:4;upfir16_fused_kernel:4;upfir16_fused_pre_kernel:4;upfir16x_fused_pre_kernel:4;upfir16x_fused_pre_noise_kernel:4;upfirs_fused_pre_kernel:4;upfirs_fused_pre_noise_kernel:4;upfirs_fused_kernel:4;upfirr_fused_kernel:4;upfirr_fused_noise_kernel:4;tile_gemm_kernel:4;winograd64_rgb_kernel:1;winograd43_rgb_kernel:5"
).split(";") if item)
seen = collections.Counter()
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        records = sorted(csv.DictReader(f), key=lambda r: (int(r["Start_Timestamp"]), r["Counter_Name"]))
        dispatch_slot = {}
        for r in records:
            name = short(r["Kernel_Name"])
            if name in SPLIT:
                did = r.get("Dispatch_Id", r["Start_Timestamp"])
                if (name, did) not in dispatch_slot:
                    dispatch_slot[(name, did)] = seen[name] % int(SPLIT[name])
                    seen[name] += 1
                name = f"{name}#launch{dispatch_slot[(name, did)]}"
            key = (name, int(r["Grid_Size"]))
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            rows[key]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
counters = sorted({c for v in rows.values() for c in v if c != "_dur_us"})
print("kernel,grid,calls,dur_us," + ",".join(counters))
for key, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]["_dur_us"])):
    ncalls = len(v[counters[0]]) if counters and v[counters[0]] else 0
    mean = lambda xs: sum(xs) / len(xs) if xs else float("nan")
    print(f"{key[0]},{key[1]},{ncalls},{mean(v['_dur_us']):.1f}," + ",".join(f"{mean(v[c]):.4g}" for c in counters))
