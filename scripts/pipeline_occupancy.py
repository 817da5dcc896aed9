"""From a rocprofv3 kernel trace of the pipelined bench.py: how busy is the GPU over the last steps?
Prints the union-busy fraction (time with >= 1 kernel running), the overlap (>= 2 running) and the idle gaps by the
kernel that ends them -- says whether the two-stream pipeline is GPU-bound or host-bound."""
import csv, glob, os, sys
src, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
path = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
starts = [i for i, r in enumerate(rows) if "rle_to_maskbits" in r[2]]
a, b = starts[-nsteps - 1], starts[-1]
win = rows[a:b]
t0, t1 = win[0][0], rows[b][0]
ev = []
for s, e, _ in win:
    ev.append((s, 1)); ev.append((min(e, t1), -1))
ev.sort()
depth, last, busy1, busy2 = 0, t0, 0, 0
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
span = t1 - t0
print(f"{nsteps} steps, {span / nsteps / 1e3:.1f} us/step: >=1 kernel running {busy1 / span:.1%}, >=2 running {busy2 / span:.1%}")
# idle gaps attributed to the kernel that starts after them
gaps = {}
end = t0
for s, e, name in win:
    if s > end:
        k = name.split("(")[0].replace("void ", "").replace("bff::", "")[:40]
        gaps[k] = gaps.get(k, 0) + (s - end)
    end = max(end, e)
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  idle before {k:42s} {v / nsteps / 1e3:8.1f} us/step")
ksum = {}
for s, e, name in win:
    k = name.split("(")[0].replace("void ", "").replace("bff::", "")[:40]
    ksum[k] = ksum.get(k, 0) + (e - s)
print("kernel time per step (us):")
for k, v in sorted(ksum.items(), key=lambda kv: -kv[1])[:30]:
    print(f"  {k:42s} {v / nsteps / 1e3:8.1f}")
