"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (name + grid), mean of every counter and duration."""
import csv, sys, collections, re

def short(name):
    m = re.search(r"modconv_mfma_kernel<([^>]*)>", name)
    if m: return "modconv<" + m.group(1).replace(" ", "") + ">"
    return name.split("(")[0].replace("gance::", "").replace("void ", "")[:40]

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            rows[key]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
counters = sorted({c for v in rows.values() for c in v if c != "_dur_us"})
print("kernel,grid,calls,dur_us," + ",".join(counters))
for key, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]["_dur_us"])):
    ncalls = len(v[counters[0]]) if counters and v[counters[0]] else 0
    mean = lambda xs: sum(xs) / len(xs) if xs else float("nan")
    print(f"{key[0]},{key[1]},{ncalls},{mean(v['_dur_us']):.1f}," + ",".join(f"{mean(v[c]):.4g}" for c in counters))
