// Point filters (include/bff_hip.h: a14, a15): the detection-ratio / occurrence thresholds of
// projection_2d_to_3d.py:512-578 without sorting N floats: the ratio masked/(viewed+1) only takes
// as many distinct values as there are distinct (masked, viewed) integer pairs, so the kernel marks
// the pairs that occur and the host does `unique()[floor(t*n)]` over that small set.
#include "common.h"

namespace bff {

__global__ void count_lattice_kernel(const int32_t *__restrict__ masked, const int32_t *__restrict__ viewed,
                                     int64_t n, int v_max, uint8_t *__restrict__ presence)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t cell = (int64_t)masked[i] * (v_max + 1) + (viewed ? viewed[i] : 0);
    presence[cell] = 1;         // racing writers all store the same byte
}

__global__ void ratio_keep_kernel(const int32_t *__restrict__ masked, const int32_t *__restrict__ viewed, int64_t n,
                                  float thr, int use_thr, int64_t nw, uint64_t *__restrict__ keep)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool k = false;
    if (i < n) {
        const int m = masked[i];
        k = m > 0;
        if (k && use_thr) {
            const float r = viewed ? __fdiv_rn((float)m, __fadd_rn((float)viewed[i], 1.0f)) : (float)m;
            k = !(r < thr);
        }
    }
    const uint64_t bal = __ballot(k);
    if (lane_id() == 0 && (i >> 6) < nw) keep[i >> 6] = bal;
}

}  // namespace bff

using namespace bff;

extern "C" int bff_count_lattice(const int32_t *masked, const int32_t *viewed, int64_t n_points, int32_t m_max,
                                 int32_t v_max, uint8_t *presence, void *stream)
{
    BFF_REQUIRE(n_points >= 0 && m_max >= 0 && v_max >= 0, "bff_count_lattice: bad sizes");
    if (n_points == 0) return BFF_OK;
    BFF_REQUIRE(masked && presence, "bff_count_lattice: null pointer");
    (void)m_max;   // bounds are the caller's contract: masked <= m_max, viewed <= v_max
    count_lattice_kernel<<<(unsigned)ceil_div(n_points, 256), 256, 0, as_stream(stream)>>>(masked, viewed, n_points, v_max, presence);
    return launched("bff_count_lattice");
}

extern "C" int bff_ratio_keep(const int32_t *masked, const int32_t *viewed, int64_t n_points, float thr,
                              int32_t use_thr, int64_t nw, uint64_t *keep, void *stream)
{
    BFF_REQUIRE(n_points >= 0 && nw == ceil_div(n_points, 64), "bff_ratio_keep: bad sizes");
    if (n_points == 0) return BFF_OK;
    BFF_REQUIRE(masked && keep, "bff_ratio_keep: null pointer");
    ratio_keep_kernel<<<(unsigned)ceil_div(nw * 64, 256), 256, 0, as_stream(stream)>>>(masked, viewed, n_points, thr, use_thr, nw, keep);
    return launched("bff_ratio_keep");
}
