import os, time, numpy as np, torch, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from beyond_fixed_forms_amd.ingest import host_lib
lib = host_lib()
rng = np.random.default_rng(0)
n = 9000; hw = 968*1296
rles = []
for i in range(n):
    k = 377
    starts = np.sort(rng.choice(hw - 10, size=k, replace=False)).astype(np.int64) + 1
    lens = np.minimum(np.diff(np.append(starts, hw)), 5)
    c = np.empty(2 * k, np.int64); c[0::2] = starts; c[1::2] = lens
    rles.append({"length": hw, "counts": c})
cap = 4_000_000
rs = np.empty(cap, np.int32); re = np.empty(cap, np.int32); offs = np.empty(n + 1, np.int32)
for thr in (1, 2, 4, 8, 16):
    for _ in range(2):
        t = time.perf_counter()
        got = lib.bff_host_pack_rles(rles, rs.ctypes.data, re.ctypes.data, cap, offs.ctypes.data, hw, thr)
        dt = time.perf_counter() - t
    print("threads", thr, "runs", got, "ms", round(dt * 1e3, 2))
t = time.perf_counter(); got = lib.bff_host_pack_rles(rles, rs.ctypes.data, re.ctypes.data, 10, offs.ctypes.data, hw, 4); dt = time.perf_counter() - t
print("capacity too small (first loop only)", got, "ms", round(dt * 1e3, 2))
