// Host-side cost of enqueueing small work on a stream: hipMemsetAsync vs a fill kernel vs an empty kernel
// (hipcc --offload-arch=gfx950 -O2 scripts/micro/launch_cost.hip -o /tmp/launch_cost && /tmp/launch_cost)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void fill_kernel(uint32_t *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = 0; }
__global__ void empty_kernel() {}
int main()
{
    hipStream_t st; hipStreamCreate(&st);
    uint32_t *p; size_t n = 1 << 20; hipMalloc(&p, n * 4);
    const int reps = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        for (int warm = 0; warm < 2; ++warm) {
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i) {
                if (mode == 0) hipMemsetAsync(p, 0, 4096, st);
                else if (mode == 1) hipMemsetAsync(p, 0, n * 4, st);
                else if (mode == 2) fill_kernel<<<(unsigned)(n / 256), 256, 0, st>>>(p, n);
                else empty_kernel<<<1, 64, 0, st>>>();
            }
            auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(st);
            auto t2 = std::chrono::steady_clock::now();
            if (warm) printf("mode %d (%s): enqueue %.2f us each, drained after %.2f us each\n", mode,
                             mode == 0 ? "memset 4 KB" : mode == 1 ? "memset 4 MB" : mode == 2 ? "fill kernel 4 MB" : "empty kernel",
                             std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
                             std::chrono::duration<double, std::micro>(t2 - t0).count() / reps);
        }
    }
    return 0;
}
