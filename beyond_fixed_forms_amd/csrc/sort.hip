// Library sorts behind the C ABI (rocPRIM radix sort), so that a host without PyTorch can run the whole path:
// the ascending sort of the per-point filter values (bff_select_unique_rank consumes it) and the stable
// argsort of the row signatures / label ids that orders the Gram tiles.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

using namespace bff;

namespace {
__global__ void iota_kernel(int32_t *v, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}
}  // namespace

// Ascending sort of n floats.  temp == NULL: only *temp_bytes is written (size query, host side, no launch).
extern "C" int bff_sort_f32(const float *keys_in, float *keys_out, int64_t n, void *temp, size_t *temp_bytes,
                            void *stream)
{
    BFF_REQUIRE(n >= 0 && temp_bytes, "bff_sort_f32: bad arguments");
    BFF_LIMIT(n < (1ll << 31), "bff_sort_f32: too many keys");
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, need, keys_in, keys_out, (unsigned)n, 0, 32, as_stream(stream));
    if (e != hipSuccess) return fail((int)e, "bff_sort_f32: %s", hipGetErrorString(e));
    if (!temp) { *temp_bytes = need; return BFF_OK; }
    BFF_REQUIRE(*temp_bytes >= need && (n == 0 || (keys_in && keys_out)), "bff_sort_f32: temp storage too small");
    if (n == 0) return BFF_OK;
    e = rocprim::radix_sort_keys(temp, need, keys_in, keys_out, (unsigned)n, 0, 32, as_stream(stream));
    if (e != hipSuccess) return fail((int)e, "bff_sort_f32: %s", hipGetErrorString(e));
    return BFF_OK;
}

// order_out = stable ascending argsort of the int64 keys (ties keep their index order; radix sort is stable).
// key_bits = 64: full signed order; key_bits < 64: the keys are non-negative and < 2^key_bits, only those bits
// are sorted (fewer radix passes).  keys_scratch: int64 [n] for the sorted keys.  temp == NULL: size query.
extern "C" int bff_argsort_i64(const int64_t *keys, int64_t *keys_scratch, int32_t *order_out, int32_t n,
                               int32_t key_bits, void *temp, size_t *temp_bytes, void *stream)
{
    BFF_REQUIRE(n >= 0 && temp_bytes && key_bits >= 1 && key_bits <= 64, "bff_argsort_i64: bad arguments");
    size_t need = 0;
    const size_t iota_bytes = ((size_t)n * sizeof(int32_t) + 255) & ~(size_t)255;     // identity permutation lives in temp
    hipError_t e = rocprim::radix_sort_pairs(nullptr, need, keys, keys_scratch, (const int32_t *)nullptr, order_out,
                                             (unsigned)n, 0, (unsigned)key_bits, as_stream(stream));
    if (e != hipSuccess) return fail((int)e, "bff_argsort_i64: %s", hipGetErrorString(e));
    if (!temp) { *temp_bytes = need + iota_bytes; return BFF_OK; }
    BFF_REQUIRE(*temp_bytes >= need + iota_bytes && (n == 0 || (keys && keys_scratch && order_out)),
                "bff_argsort_i64: temp storage too small");
    if (n == 0) return BFF_OK;
    int32_t *iota = reinterpret_cast<int32_t *>(temp);
    iota_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, as_stream(stream)>>>(iota, n);
    e = rocprim::radix_sort_pairs(reinterpret_cast<char *>(temp) + iota_bytes, need, keys, keys_scratch, iota, order_out,
                                  (unsigned)n, 0, (unsigned)key_bits, as_stream(stream));
    if (e != hipSuccess) return fail((int)e, "bff_argsort_i64: %s", hipGetErrorString(e));
    return BFF_OK;
}
