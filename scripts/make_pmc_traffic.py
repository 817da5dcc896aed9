"""profiles/pmc_traffic.json from the counter summaries of one evidence directory (scripts/collect_profiles.sh).

usage: python scripts/make_pmc_traffic.py profiles/r02_final

Per configuration: HBM-side bytes per launch of the projection sweep = FETCH_SIZE x correction + WRITE_SIZE (both
in KiB per dispatch, mean over the timed dispatches), the correction being what the same session's calibration
kernels give: scripts/diag_membw.py runs access patterns with a KNOWN number of distinct 128-byte lines -- contiguous
floats and one 4-byte gather per 64-byte half line / per 128-byte line / per image row -- and the factor is
known_line_bytes / counted_bytes of each.  (MI355X_MICROARCH.md: gfx950 tallies a 128-byte request as 64 B.)
"""
import csv
import json
import os
import sys

src = sys.argv[1].rstrip("/")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc(path):
    out = {}
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(l for l in open(path) if not l.startswith("#")):
        out[(r["kernel"], r["counter"])] = (float(r["mean"]), int(r["dispatches"]), float(r["min"]), float(r["max"]))
    return out


cal = pmc(os.path.join(src, "cal_fetch_pmc.csv"))
# known distinct 128-B lines per launch of scripts/diag_membw.py (8 Mi lanes; stride 1296: 1 Mi lanes)
known = {"gather_stride_kernel<1>": ("contiguous floats", 8388608 * 4),
         "gather_stride_kernel<16>": ("one 4-B gather per 64-B half line", 4194304 * 128),
         "gather_stride_kernel<32>": ("one 4-B gather per 128-B line", 8388608 * 128),
         "gather_stride_kernel<0>": ("one 4-B gather per line, 5184 B apart (image rows)", 1048576 * 128)}
calibration, factors = {}, []
for k, (what, line_bytes) in known.items():
    if (k, "FETCH_SIZE") in cal:
        counted = cal[(k, "FETCH_SIZE")][0] * 1024
        calibration[k] = {"pattern": what, "known_line_bytes": line_bytes, "fetch_size_bytes": round(counted),
                          "factor": round(line_bytes / counted, 4)}
        factors.append(line_bytes / counted)
corr = round(sum(factors) / len(factors), 3) if factors else 2.0

out = {}
for shape, suffix in (("c2", ""), ("c4", "_c4")):
    f = pmc(os.path.join(src, f"pmc_fetch{suffix}_pmc.csv"))
    w = pmc(os.path.join(src, f"pmc_write{suffix}_pmc.csv"))
    # the timed loop's variant is the one with the most dispatches (a diagnostic after the loop launches another once)
    fk = sorted((k for k in f if "project_views_kernel" in k[0]), key=lambda k: -f[k][1])
    wk = sorted((k for k in w if "project_views_kernel" in k[0]), key=lambda k: -w[k][1])
    if not fk or not wk:
        continue
    fetch_kb, nf = f[fk[0]][0], f[fk[0]][1]
    write_kb = w[wk[0]][0]
    # config 2 keeps its depth at the sensor's resolution (bench.py looks the entry up as ..._u16), config 4 as float32 (H, W)
    out[f"project_views_{shape}" + ("_u16" if shape == "c2" else "")] = {
        "bytes": int(round(fetch_kb * 1024 * corr + write_kb * 1024)),
        "fetch_kb_raw": round(fetch_kb), "write_kb": round(write_kb), "fetch_correction": corr, "dispatches": nf,
        "fetch_kb_min_max": [round(f[fk[0]][2]), round(f[fk[0]][3])],
        "source": f"{os.path.relpath(src, ROOT)}/pmc_fetch{suffix}_pmc.csv, pmc_write{suffix}_pmc.csv (project_views_kernel, "
                  f"mean over the dispatches of the rotating scenes)",
        "note": "FETCH_SIZE x correction + WRITE_SIZE.  The correction is calibrated in the same evidence directory on "
                "4-byte gathers with a known number of distinct 128-B lines (see `calibration`): every distinct line "
                "costs 128 B whatever part of it is used, and the counter tallies it as 64 B.  FETCH_SIZE counts L2 "
                "misses, i.e. requests to the fabric: lines served by the Infinity Cache are included, so this is an "
                "upper bound of the HBM bytes."}
out["calibration"] = {"kernels": calibration, "correction_used": corr,
                      "source": f"{os.path.relpath(src, ROOT)}/cal_fetch_pmc.csv (scripts/diag_membw.py under --pmc FETCH_SIZE)"}
dst = os.path.join(ROOT, "profiles", "pmc_traffic.json")
with open(dst, "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1))
