// Shared helpers for libbff_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "bff_hip.h"

namespace bff {

constexpr int kWave = 64;   // CDNA wavefront
constexpr int kCW = 8;      // words per row chunk (512 points): granularity of the rows' occupancy masks

char *err_buf();

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int launched(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return BFF_OK;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// bff_scene_project clears every scratch buffer of its steps with ONE fill (they are laid out in one block,
// bff_scene_workspace) and sets this flag for the duration of the call: the steps' own clears become no-ops.
int &scratch_prezeroed();
inline hipError_t zero_async(void *p, size_t bytes, hipStream_t st)
{
    return scratch_prezeroed() ? hipSuccess : hipMemsetAsync(p, 0, bytes, st);
}

#define BFF_REQUIRE(cond, ...) \
    do { if (!(cond)) return ::bff::fail(BFF_E_ARG, __VA_ARGS__); } while (0)
#define BFF_LIMIT(cond, ...) \
    do { if (!(cond)) return ::bff::fail(BFF_E_LIMIT, __VA_ARGS__); } while (0)

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

__device__ __forceinline__ int popc64(uint64_t v) { return __popcll(v); }

}  // namespace bff
