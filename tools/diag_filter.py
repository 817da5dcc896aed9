import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beyond_fixed_forms_amd import _lib
from beyond_fixed_forms_amd._lib import _ptr, call, c_void_p, i32, f32
dev = "cuda:0"
n = 200_000
g = torch.Generator(device=dev).manual_seed(0)
masked = (torch.randint(0, 700, (n,), device=dev, generator=g) * (torch.rand(n, device=dev, generator=g) < 0.5)).to(torch.int32)
viewed = torch.randint(0, 300, (n,), device=dev, generator=g).to(torch.int32)
def timeit(name, fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); print(f"{name:40s} {1e6*(time.perf_counter()-t)/reps:8.1f} us")
m_max, v_max = 9000, 300
n_cells = (m_max+1)*(v_max+1)
for cap in (1<<18, 1<<16):
    bitmap = torch.empty((n_cells+31)//32, dtype=i32, device=dev); cells = torch.empty(cap+1, dtype=i32, device=dev); info = torch.empty(3, dtype=i32, device=dev)
    vals = torch.empty(cap, dtype=f32, device=dev); thr = torch.empty(1, dtype=f32, device=dev)
    timeit(f"compact cap={cap}", lambda: call("bff_count_lattice_compact", _ptr(masked), _ptr(viewed), n, m_max, v_max, _ptr(bitmap), _ptr(cells), cap, c_void_p(info.data_ptr()+8)))
    print("   pairs:", int(cells[0]))
    timeit(f"values cap={cap}", lambda: call("bff_lattice_values", _ptr(cells), cap, v_max, 1, _ptr(vals)))
    timeit(f"torch.sort cap={cap}", lambda: torch.sort(vals))
    sv = torch.sort(vals).values
    timeit(f"select cap={cap}", lambda: call("bff_select_unique_rank", _ptr(sv), cap, 0.38, _ptr(thr), c_void_p(info.data_ptr()+4)))
    timeit(f"whole lattice_threshold cap={cap}", lambda: _lib.lattice_threshold(masked, viewed, m_max, v_max, 0.38, cap=cap))
timeit("torch.unique(ratio) for comparison", lambda: (masked.float()/(viewed.float()+1)).unique())
