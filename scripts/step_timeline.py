"""From a rocprofv3 kernel trace of bench.py: timeline of the LAST n steps (kernel start offsets, durations, idle
gaps), every kernel of the process listed (library sorts and fills included).  usage: step_timeline.py <dir> [n=1] [first]
(first: index of the first step to print instead of the last n -- bench.py runs diagnostic launches after its timed loop)"""
import csv, glob, os, sys
src = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 1
path = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at rle_to_maskbits
starts = [i for i, r in enumerate(rows) if "rle_to_maskbits" in r["Kernel_Name"]]
first = int(sys.argv[3]) if len(sys.argv) > 3 else max(0, len(starts) - n_last)
for si in range(first, min(len(starts), first + n_last)):
    a = starts[si]
    b = starts[si + 1] if si + 1 < len(starts) else len(rows)
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    prev_end = t0
    busy = 0
    print(f"--- step {si}")
    print(f"{'t_us':>8} {'dur_us':>8} {'gap_us':>8}  kernel")
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bff::", "")[:60]
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:8.1f}  {name}")
        busy += e - s
        prev_end = max(prev_end, e)
    print(f"step span {(prev_end - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
