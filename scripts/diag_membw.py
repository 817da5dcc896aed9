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
