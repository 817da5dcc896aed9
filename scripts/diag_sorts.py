"""Time the two library sorts at config-2 sizes (200k f32 values, 9000 int64 signatures)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_fixed_forms_amd import _lib
_lib.load()
dev = "cuda"
def timeit(f, reps=200):
    for _ in range(10): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
v = (torch.randint(0, 300, (200_000,), device=dev).float() / torch.randint(1, 300, (200_000,), device=dev).float())
k = torch.randint(0, 1 << 62, (9000,), device=dev, dtype=torch.int64)
print("sort_f32 200k      %.1f us" % timeit(lambda: _lib.sort_f32(v)))
print("torch.sort 200k    %.1f us" % timeit(lambda: torch.sort(v)))
print("argsort_i64 9000   %.1f us" % timeit(lambda: _lib.argsort_i64(k)))
print("argsort 30 bits    %.1f us" % timeit(lambda: _lib.argsort_i64(k >> 33, 30)))
print("torch.argsort 9000 %.1f us" % timeit(lambda: torch.argsort(k, stable=True)))
