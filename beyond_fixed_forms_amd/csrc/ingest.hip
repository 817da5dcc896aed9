// Scene ingestion on the device: the spatial sort of the cloud and its layout.
//
// scene.morton_order (host, NumPy) restated as kernels: bounding box -> 30-bit Morton code of every point (10 bits
// per axis over the box, NaN / inf coordinates count as 0) -> stable radix sort -> structure-of-arrays cloud in
// sorted order, the inverse permutation (`unsort`) that maps results back to the caller's point order, and the
// bounding boxes of the sweep's point tiles.  Same codes and a stable sort: the permutation equals the host's.
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

using namespace bff;

namespace {

__device__ __forceinline__ double finite_or_zero(double v) { return (v - v == 0.0) ? v : 0.0; }   // NaN, +-inf -> 0

// one block: min / max of the three coordinates over all points (points are few hundred thousand .. a million)
__global__ __launch_bounds__(1024) void cloud_bounds_kernel(const double *__restrict__ pts, int64_t n, int64_t stride,
                                                             double *__restrict__ box /* lo[3], hi[3] */)
{
    __shared__ double s_lo[16][3], s_hi[16][3];
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = threadIdx.x; i < n; i += 1024)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double v = finite_or_zero(pts[i * stride + a]);
            lo[a] = fmin(lo[a], v);
            hi[a] = fmax(hi[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            lo[a] = fmin(lo[a], __shfl_xor(lo[a], d));
            hi[a] = fmax(hi[a], __shfl_xor(hi[a], d));
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_lo[wave][a] = lo[a]; s_hi[wave][a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double l = s_lo[0][threadIdx.x], h = s_hi[0][threadIdx.x];
        for (int w = 1; w < 16; ++w) { l = fmin(l, s_lo[w][threadIdx.x]); h = fmax(h, s_hi[w][threadIdx.x]); }
        box[threadIdx.x] = l;
        box[3 + threadIdx.x] = h;
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void morton_codes_kernel(const double *__restrict__ pts, int64_t n, int64_t stride,
                                    const double *__restrict__ box, uint32_t *__restrict__ codes, int32_t *__restrict__ iota)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double lo = box[a], ext = fmax(box[3 + a] - lo, 1e-300);
        const double t = (finite_or_zero(pts[i * stride + a]) - lo) / ext * 1023.0;       // same expression as the host
        const uint32_t u = (uint32_t)t;                                                   // t in [0, 1023]: truncation
        q[a] = u > 1023u ? 1023u : u;
    }
    codes[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    iota[i] = (int32_t)i;
}

__global__ void cloud_gather_kernel(const double *__restrict__ pts, int64_t n, int64_t stride, int64_t n_pad,
                                    const int32_t *__restrict__ perm, double *__restrict__ soa, int32_t *__restrict__ unsort)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // sorted position
    if (s >= n_pad) return;
    if (s < n) {
        const int32_t o = perm ? perm[s] : (int32_t)s;
#pragma unroll
        for (int a = 0; a < 3; ++a) soa[a * n_pad + s] = pts[(int64_t)o * stride + a];
        if (unsort) unsort[o] = (int32_t)s;
    } else {
#pragma unroll
        for (int a = 0; a < 3; ++a) soa[a * n_pad + s] = 0.0;
    }
}

}  // namespace

// pts: float64 [n][stride] (stride >= 3: the (N, 6) xyz+rgb array of <scene>.npy works as is).  Outputs: soa float64
// [3][n_pad], unsort int32 [n] (position of original point o in the sorted cloud), perm int32 [n] (scratch / inverse),
// codes uint32 [2 n] scratch, box float64 [6] scratch, temp for the sort (size query when temp == NULL).
// sort == 0: keep the caller's order (unsort is not written).
extern "C" int bff_cloud_layout(const double *pts, int64_t n, int64_t stride, int64_t n_pad, int32_t sort, double *soa,
                                int32_t *unsort, int32_t *perm, uint32_t *codes, double *box, void *temp,
                                size_t *temp_bytes, void *stream)
{
    BFF_REQUIRE(n >= 0 && stride >= 3 && n_pad >= n && temp_bytes, "bff_cloud_layout: bad sizes");
    BFF_LIMIT(n < (1ll << 31), "bff_cloud_layout: too many points");
    hipStream_t st = as_stream(stream);
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, need, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                             (const int32_t *)nullptr, (int32_t *)nullptr, (unsigned)n, 0, 30, st);
    if (e != hipSuccess) return fail((int)e, "bff_cloud_layout: %s", hipGetErrorString(e));
    const size_t iota_bytes = ((size_t)n * sizeof(int32_t) + 255) & ~(size_t)255;
    if (!temp) { *temp_bytes = need + iota_bytes; return BFF_OK; }
    if (n == 0) return BFF_OK;
    BFF_REQUIRE(pts && soa, "bff_cloud_layout: null pointer");
    if (sort) {
        BFF_REQUIRE(unsort && perm && codes && box && *temp_bytes >= need + iota_bytes, "bff_cloud_layout: scratch missing");
        int32_t *iota = reinterpret_cast<int32_t *>(temp);
        cloud_bounds_kernel<<<1, 1024, 0, st>>>(pts, n, stride, box);
        morton_codes_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(pts, n, stride, box, codes, iota);
        e = rocprim::radix_sort_pairs(reinterpret_cast<char *>(temp) + iota_bytes, need, codes, codes + n, iota, perm,
                                      (unsigned)n, 0, 30, st);
        if (e != hipSuccess) return fail((int)e, "bff_cloud_layout: %s", hipGetErrorString(e));
    }
    cloud_gather_kernel<<<(unsigned)ceil_div(n_pad, 256), 256, 0, st>>>(pts, n, stride, n_pad, sort ? perm : nullptr, soa,
                                                                       sort ? unsort : nullptr);
    return launched("bff_cloud_layout");
}
