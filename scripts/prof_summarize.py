"""Condense rocprofv3 CSV output (kernel stats + PMC counter collection) into small per-kernel tables.

usage: python scripts/prof_summarize.py <rocprof_out_dir> <summary_out_dir> [name_filter=bff]
Keeps: header lines, every kernel whose name contains the filter, and the 12 heaviest other kernels.
"""
import csv
import glob
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else "bff"
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    return name.split("(")[0][:110]


for path in glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(path)))
    tag = os.path.relpath(path, src).split(os.sep)[0]
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
        if rows:
            keys = list(rows[0].keys())
            f.write(",".join(keys) + "\n")
            mine = [r for r in rows if flt in r.get("Name", "")]
            rest = [r for r in rows if flt not in r.get("Name", "")][:12]
            for r in mine + rest:
                r = dict(r); r["Name"] = '"' + short(r["Name"]) + '"'
                f.write(",".join(str(r[k]) for k in keys) + "\n")

for path in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
    tag = os.path.relpath(path, src).split(os.sep)[0]
    agg = defaultdict(lambda: defaultdict(list))
    rd = csv.DictReader(open(path))
    header = rd.fieldnames
    for r in rd:
        name = r.get("Kernel_Name", "")
        if flt not in name:
            continue
        agg[short(name)][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", "nan")))
    with open(os.path.join(dst, f"{tag}_pmc.csv"), "w") as f:
        f.write("# source columns: " + " ".join(header or []) + "\n")
        f.write("kernel,counter,dispatches,mean,min,max\n")
        for k, cs in sorted(agg.items()):
            for c, vals in sorted(cs.items()):
                f.write(f'"{k}",{c},{len(vals)},{sum(vals) / len(vals):.6g},{min(vals):.6g},{max(vals):.6g}\n')

for path in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
    tag = os.path.relpath(path, src).split(os.sep)[0]
    agg = defaultdict(list)
    extra = {}
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        if flt not in name:
            continue
        agg[short(name)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        extra[short(name)] = (r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")), r.get("SGPR_Count", ""),
                              r.get("LDS_Block_Size", ""), r.get("Grid_Size", r.get("Grid_Size_X", "")),
                              r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")))
    with open(os.path.join(dst, f"{tag}_kernel_trace_summary.csv"), "w") as f:
        f.write("kernel,dispatches,mean_us,min_us,max_us,vgpr,sgpr,lds,grid,workgroup\n")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            f.write(f'"{k}",{len(v)},{sum(v) / len(v):.2f},{min(v):.2f},{max(v):.2f},' + ",".join(map(str, extra[k])) + "\n")
print("summaries:", sorted(os.listdir(dst)))
