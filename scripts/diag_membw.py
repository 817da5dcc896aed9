"""Streaming write / read / copy rates on this GPU (what a pure HBM-bound kernel can reach)."""
import torch
def timeit(f, reps=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
for gb in (0.225, 1.16, 3.0):
    n = int(gb * 1e9 / 4)
    x = torch.empty(n, dtype=torch.int32, device="cuda")
    y = torch.empty(n, dtype=torch.int32, device="cuda")
    tw = timeit(lambda: x.zero_())
    tf = timeit(lambda: x.fill_(7))
    tr = timeit(lambda: x.sum())
    tc = timeit(lambda: y.copy_(x))
    print(f"{gb:5.3f} GB: memset {gb / tw / 1e3:.2f} TB/s ({tw * 1e6:.0f} us), fill kernel {gb / tf / 1e3:.2f} TB/s, "
          f"read(sum) {gb / tr / 1e3:.2f} TB/s, copy {2 * gb / tc / 1e3:.2f} TB/s (r+w)")

# ---- FETCH_SIZE calibration for 4-byte gathers (run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`): every
# launch touches n_lanes distinct elements `stride` floats apart; known distinct 128-B lines / 64-B halves per launch:
#   stride  1: n_lanes * 4 bytes contiguous          stride 16: one lane per 64-B half line
#   stride 32: one lane per 128-B line               stride 1296: one lane per line, 5184 B apart (image rows)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beyond_fixed_forms_amd import _lib
_lib.load()
n_lanes = 8 * 1024 * 1024
for stride in (1, 16, 32, 1296):
    lanes = n_lanes if stride <= 32 else 1024 * 1024
    src = torch.zeros(lanes * stride, dtype=torch.float32, device="cuda")
    out = torch.zeros(lanes, dtype=torch.float32, device="cuda")
    f = lambda: _lib.call("bff_diag_gather", _lib._ptr(src), lanes, stride, _lib._ptr(out))
    t = timeit(f, reps=5)
    lines128 = lanes * 4 / 128 if stride == 1 else lanes / (2 if stride == 16 else 1)
    print(f"gather stride {stride:5d}: {lanes} lanes, {lines128:.0f} distinct 128-B lines ({lines128 * 128 / 1e6:.1f} MB of lines, "
          f"{lanes * 4 / 1e6:.1f} MB useful), {t * 1e6:.0f} us -> {lanes / t / 1e9:.2f} G gathers/s, {lines128 * 128 / t / 1e12:.2f} TB/s of lines")
    del src, out
