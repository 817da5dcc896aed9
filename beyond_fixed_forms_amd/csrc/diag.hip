// Measurement aids (scripts/diag_membw.py, bench.py): access patterns with a KNOWN number of distinct cache lines, to
// calibrate the FETCH_SIZE counter for the sweep's 4-byte gathers (MI355X_MICROARCH.md: only wide coalesced reads are
// calibrated -- x2 on gfx950 -- "calibrate on a known byte count in your own access pattern").
#include "common.h"

using namespace bff;

namespace {

// Every lane reads ONE float at element  ((wave * 64 + lane) * stride + jitter(lane))  of src, with
//   stride = 1   : 256 contiguous bytes per wave (2 x 128-B lines): the coalesced reference point
//   stride = 16  : one lane per 64-B half line
//   stride = 32  : one lane per 128-B line
//   stride = 1296: consecutive lanes in consecutive image rows (the worst case of the sweep's depth gather)
// Each element is touched exactly once per launch, the footprint is n_lanes * stride * 4 bytes (>> every cache).
template <int kTag>
__global__ void gather_stride_kernel(const float *__restrict__ src, int64_t n_lanes, int64_t stride, float *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_lanes) return;
    const float v = src[t * stride];
    if (v == 12345.678f) out[t] = v;           // never true for the zero-filled table: keeps the load alive, writes nothing
}

}  // namespace

// pattern: 1, 16, 32 or any other stride (tagged 0 in the kernel name).  src must hold n_lanes * stride floats.
extern "C" int bff_diag_gather(const float *src, int64_t n_lanes, int64_t stride, float *out, void *stream)
{
    BFF_REQUIRE(src && out && n_lanes > 0 && stride > 0, "bff_diag_gather: bad arguments");
    const unsigned grid = (unsigned)ceil_div(n_lanes, 256);
    hipStream_t st = as_stream(stream);
    if (stride == 1) gather_stride_kernel<1><<<grid, 256, 0, st>>>(src, n_lanes, stride, out);
    else if (stride == 16) gather_stride_kernel<16><<<grid, 256, 0, st>>>(src, n_lanes, stride, out);
    else if (stride == 32) gather_stride_kernel<32><<<grid, 256, 0, st>>>(src, n_lanes, stride, out);
    else gather_stride_kernel<0><<<grid, 256, 0, st>>>(src, n_lanes, stride, out);
    return launched("bff_diag_gather");
}
