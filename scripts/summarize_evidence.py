"""Markdown summary of one evidence directory (profiles/<name>): the numbers DESIGN.md section 4 quotes.

usage: python scripts/summarize_evidence.py profiles/r02_final
"""
import csv
import json
import os
import sys

src = sys.argv[1].rstrip("/")


def bench(name):
    p = os.path.join(src, f"bench_{name}.json")
    if not os.path.exists(p):
        return None
    t = open(p).read().strip()
    for line in reversed(t.splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return None


def table(name):
    p = os.path.join(src, name)
    if not os.path.exists(p):
        return []
    return list(csv.DictReader(l for l in open(p) if not l.startswith("#")))


print(f"evidence: {src}\n")
print("| run | scenes/s | ms/step | sweep ms (in loop / alone) | tile pass ms (in loop / alone) | host: issue + back - wait = work (ms) |")
print("|---|---|---|---|---|---|")
for name in ("c2", "c2_with_cpu_baseline", "c2_under_rocprof", "c2_no_pipeline", "c2_include_upload", "c1", "c4", "c4_no_pipeline"):
    d = bench(name)
    if d is None:
        continue
    k, r, rm, h = d.get("kernels_ms") or {}, d.get("roofline") or {}, d.get("roofline_merge") or {}, d.get("host_ms") or {}
    alone = (r.get("alone_on_chip") or {}).get("avg_launch_ms")
    print(f"| {name} | {d['value']:.1f} | {d['ms_per_step']:.3f} | {k.get('project_views', 0):.3f} / {alone if alone is None else round(alone, 3)} | "
          f"{k.get('merge_components', 0):.3f} / {round(rm.get('alone_on_chip_ms', 0), 3)} | "
          f"{h.get('front_issue')} + {h.get('back')} - {h.get('of_which_waiting_for_gpu')} = {h.get('host_work')} |")
for name in ("c2", "c4"):
    d = bench(name)
    if d is None:
        continue
    r = d["roofline"]
    c = r.get("compulsory") or {}
    print(f"\n{name} sweep roofline: algorithmic {r['algorithmic_bytes_per_launch'] / 1e9:.3f} GB / {r['avg_launch_ms']:.3f} ms = "
          f"{r['achieved']:.0f} GB/s = {r['frac']:.3f} of 8 TB/s (in the loop); alone {r['alone_on_chip']['avg_launch_ms']:.3f} ms -> "
          f"{r['alone_on_chip']['frac']:.3f}; compulsory {c.get('bytes', 0) / 1e9:.3f} GB -> {c.get('achieved', 0):.0f} GB/s = {c.get('frac', 0):.3f}; "
          f"PMC traffic {None if r.get('traffic') is None else round(r['traffic'] / 1e9, 3)} GB")
    rm = d.get("roofline_merge") or {}
    print(f"{name} tile pass: {rm.get('avg_launch_ms', 0):.3f} ms in the loop, {rm.get('alone_on_chip_ms', 0):.3f} alone; staged {rm.get('l2_bytes_staged', 0) / 1e9:.3f} GB "
          f"-> {rm.get('achieved', 0):.0f} GB/s of L2->LDS; tile pairs {rm.get('tile_pairs')}, chunk visits {rm.get('chunk_visits')}, "
          f"candidate pairs {rm.get('candidate_pairs')}, unions {rm.get('unions')}")
    print(f"{name} device span of the scene call: {d.get('scene_call_device_span_ms')}; result {d.get('result')}")
d = bench("c2_with_cpu_baseline")
if d and d.get("cpu_baseline"):
    cb = d["cpu_baseline"]
    print(f"\ncpu_baseline: {cb['value']:.4f} {cb['unit']} on {cb['cores']} cores ({cb['kind']}); stages {cb.get('stages_sample_s')}")
d = bench("c2_include_upload")
if d:
    for key in ("host_inclusive", "include_upload", "upload_leg"):
        if d.get(key):
            print(f"\nhost-inclusive: {json.dumps(d[key])}")
d = bench("c5")
if d:
    print(f"\nc5: {d['value']:.0f} {d['unit']}, {d['ms_per_step'] * 1e3:.1f} us per GEMM, {d['roofline']['achieved']:.1f} TFLOP/s")

print("\n| kernel | launches | mean us | min | max | VGPR | LDS B |")
print("|---|---|---|---|---|---|---|")
for r in table("stats_kernel_trace_summary.csv")[:22]:
    print(f"| {r['kernel']} | {r['dispatches']} | {r['mean_us']} | {r['min_us']} | {r['max_us']} | {r['vgpr']} | {r['lds']} |")
for tag in ("pmc_fetch", "pmc_write", "pmc_fetch_c4", "pmc_write_c4"):
    rows = [r for r in table(f"{tag}_pmc.csv") if any(k in r["kernel"] for k in ("project_views", "rle_to_maskbits", "merge_components_kernel<0>", "or_reduce"))]
    for r in rows:
        print(f"{tag}: {r['kernel']} {r['counter']} mean {float(r['mean']) / 1024:.1f} MiB (x{r['dispatches']}, {float(r['min']) / 1024:.1f} .. {float(r['max']) / 1024:.1f})")
