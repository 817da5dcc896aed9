// Point filters (include/bff_hip.h: a14, a15): the detection-ratio / occurrence thresholds of
// projection_2d_to_3d.py:512-578 without sorting N floats: the ratio masked/(viewed+1) only takes
// as many distinct values as there are distinct (masked, viewed) integer pairs, so the kernel marks
// the pairs that occur and the host does `unique()[floor(t*n)]` over that small set.
#include "common.h"

namespace bff {

// vals[i] = float32 filter statistic of point i: masked/(viewed+1) (P:571) or masked (P:513)
__global__ void point_values_kernel(const int32_t *__restrict__ masked, const int32_t *__restrict__ viewed, int64_t n,
                                    float *__restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float m = (float)masked[i];
    vals[i] = viewed ? __fdiv_rn(m, __fadd_rn((float)viewed[i], 1.0f)) : m;
}

// sorted ascending -> thr = (distinct values)[floor(frac * n_distinct)], exactly the reference's
// `x.unique()[math.floor(frac * x.unique().shape[0])]` (P:516-518, 574-576; frac * n is a float64 product).
// Pass 1: every block counts the positions that start a run of equal values.  Pass 2 (one block): prefix
// over the block counts gives n_distinct and the block holding the wanted rank; that block's slice is
// scanned by one wave.
constexpr int kSelBlock = 1024;      // elements per block in pass 1

__global__ __launch_bounds__(256) void distinct_count_kernel(const float *__restrict__ sorted, int64_t n,
                                                              int32_t *__restrict__ block_count)
{
    __shared__ int part[4];
    const int64_t base = (int64_t)blockIdx.x * kSelBlock;
    int c = 0;
#pragma unroll
    for (int k = 0; k < kSelBlock / 256; ++k) {
        const int64_t i = base + threadIdx.x + k * 256;
        if (i < n) c += (i == 0) || (sorted[i] != sorted[i - 1]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
    if (lane_id() == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(1024) void distinct_select_kernel(const float *__restrict__ sorted, int64_t n,
                                                                const int32_t *__restrict__ block_count, int n_blocks,
                                                                double frac, float *__restrict__ thr,
                                                                int32_t *__restrict__ n_unique)
{
    __shared__ int wsum[16];
    __shared__ int s_total, s_block, s_before;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n_blocks + 1023) / 1024;
    const int lo = tid * per, hi = min(n_blocks, lo + per);
    int mine = 0;
    for (int b = lo; b < hi; ++b) mine += block_count[b];
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
    if (lane == 63) wsum[wave] = incl;
    if (tid == 0) s_block = -1;
    __syncthreads();
    int before = incl - mine;
    for (int q = 0; q < wave; ++q) before += wsum[q];
    if (tid == 1023) s_total = before + mine;
    __syncthreads();
    const int total = s_total;
    const long long rank = (long long)floor(frac * (double)total);
    if (rank >= 0 && rank < total && rank >= before && rank < before + mine) {
        int acc = before;
        for (int b = lo; b < hi; ++b) {
            if (rank < acc + block_count[b]) { s_block = b; s_before = acc; break; }
            acc += block_count[b];
        }
    }
    __syncthreads();
    if (tid == 0) { *n_unique = total; if (s_block < 0) *thr = __builtin_nanf(""); }
    if (s_block < 0 || wave != 0) return;
    // one wave walks the 1024 elements of the chosen block, 64 at a time
    const int64_t base = (int64_t)s_block * kSelBlock;
    int seen = s_before;
    for (int k = 0; k < kSelBlock / 64; ++k) {
        const int64_t i = base + k * 64 + lane;
        const bool start = i < n && ((i == 0) || (sorted[i] != sorted[i - 1]));
        const uint64_t bal = __ballot(start);
        const int my = seen + __popcll(bal & ((1ull << lane) - 1));
        if (start && my == (int)rank) *thr = sorted[i];
        seen += __popcll(bal);
        if (seen > rank) break;
    }
}

// ---- the same threshold without sorting N values --------------------------------------------------------
// The filter statistic of a point is a function of two small integers, (masked, viewed): float32(masked) /
// (float32(viewed) + 1) (or float32(masked) alone).  A scene has ~10^5..10^6 points but only ~10^3..10^4 distinct
// values, so: (1) every block of 1024 points collects the distinct values of ITS points in an LDS hash set (most
// points repeat a value -- half of them are 0 -- and a repeated value costs one probe) and writes them to its own
// slice of a scratch table: no global atomics, nothing to clear; (2) 64 blocks, one per 1/64 of the hash space, merge
// the slices into LDS sets -- together exactly x.unique() of the reference -- and append what they hold to one list;
// (3) one block selects the value of rank floor(frac * n_distinct) with a 4-pass byte-wise radix select (values are
// >= 0, so the order of the bit patterns is the order of the values).  Three launches instead of the 12 of the sorting
// path (value pass, 9-launch radix sort, two selection passes); bit-identical threshold.  A partition with more
// distinct values than its set holds: *overflow = 1 and the caller sorts.
constexpr uint32_t kLocalSlots = 2048;                        // LDS set of one block (1024 points -> <= 1024 values)
constexpr uint32_t kHashEmpty = 0xFFFFFFFFu;                  // not a value: a NaN pattern (the statistic is never NaN)
constexpr int kSliceWords = 1025;                             // per block: count, then <= 1024 values

__device__ __forceinline__ float pair_value(int m, int v, bool ratio)
{
    return ratio ? __fdiv_rn((float)m, __fadd_rn((float)v, 1.0f)) : (float)m;
}

__global__ __launch_bounds__(1024) void block_value_sets_kernel(const int32_t *__restrict__ masked,
                                                                 const int32_t *__restrict__ viewed, int64_t n,
                                                                 uint32_t *__restrict__ slices)
{
    __shared__ uint32_t local[kLocalSlots];
    __shared__ uint32_t s_n;
    const int tid = threadIdx.x;
    for (int q = tid; q < (int)kLocalSlots; q += 1024) local[q] = kHashEmpty;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 1024 + tid;
    if (i < n) {
        const uint32_t bits = __float_as_uint(pair_value(masked[i], viewed ? viewed[i] : 0, viewed != nullptr));
        uint32_t h = (bits * 2654435761u) >> (32 - 11);
        for (;;) {                                             // <= 1024 values in 2048 slots: always terminates
            uint32_t old = local[h];
            if (old == kHashEmpty) old = atomicCAS(&local[h], kHashEmpty, bits);
            if (old == kHashEmpty || old == bits) break;
            h = (h + 1) & (kLocalSlots - 1);
        }
    }
    __syncthreads();
    uint32_t *out = slices + (int64_t)blockIdx.x * kSliceWords;
    for (int q = tid; q < (int)kLocalSlots; q += 1024) {
        const uint32_t bits = local[q];
        const uint64_t bal = __ballot(bits != kHashEmpty);
        uint32_t base = 0;
        if (lane_id() == 0 && bal) base = atomicAdd(&s_n, (uint32_t)__popcll(bal));
        base = __shfl(base, 0);
        if (bits != kHashEmpty) out[1 + base + __popcll(bal & ((1ull << lane_id()) - 1))] = bits;
    }
    __syncthreads();
    if (tid == 0) out[0] = s_n;
}

// (2) kParts blocks, each owning 1/kParts of the hash space: a block reads every slice (they sit in L2) and keeps the
// values of its partition in an LDS set; what it holds at the end is appended to `values` and counted in *n_unique.
// One merging block alone spends ~100 us at config 2 on the dependent LDS probes of ~10^5 inserts; 64 blocks share them.
constexpr int kParts = 64;
constexpr uint32_t kPartSlots = 8192;                         // LDS set of one partition (32 KiB)
constexpr uint32_t kPartMax = kPartSlots / 4 * 3;             // distinct values a partition accepts

__global__ __launch_bounds__(1024) void partition_sets_kernel(const uint32_t *__restrict__ slices, int n_slices,
                                                               uint32_t *__restrict__ values, int32_t *__restrict__ n_unique,
                                                               int32_t *__restrict__ overflow, uint32_t part_max)
{
    __shared__ uint32_t set[kPartSlots];
    __shared__ uint32_t s_count, s_full, s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t part = blockIdx.x;
    for (uint32_t q = tid; q < kPartSlots; q += 1024) set[q] = kHashEmpty;
    if (tid == 0) { s_count = 0; s_full = 0; }
    __syncthreads();
    // wave w reads slices w, w + 16, ...: a slice's <= 1024 values in ONE round trip (16 per lane in flight together,
    // the next slice's length with them)
    constexpr int kPerLane = 1024 / kWave;
    uint32_t cnt = wave < n_slices ? slices[(int64_t)wave * kSliceWords] : 0;
    for (int b = wave; b < n_slices; b += 16) {
        const uint32_t *sl = slices + (int64_t)b * kSliceWords;
        uint32_t v[kPerLane];
#pragma unroll
        for (int q = 0; q < kPerLane; ++q) {
            const uint32_t k = lane + kWave * q;
            v[q] = k < cnt ? sl[1 + k] : kHashEmpty;
        }
        const uint32_t cnt_next = b + 16 < n_slices ? slices[(int64_t)(b + 16) * kSliceWords] : 0;
#pragma unroll
        for (int q = 0; q < kPerLane; ++q) {
            const uint32_t bits = v[q];
            const uint32_t hash = bits * 2654435761u;
            if (bits == kHashEmpty || (hash >> 26) != part) continue;              // top 6 bits: the partition
            uint32_t h = (hash >> 13) & (kPartSlots - 1);                          // next 13 bits: the slot
            for (uint32_t probe = 0; probe < kPartSlots; ++probe) {
                uint32_t old = set[h];
                if (old == kHashEmpty) {
                    if (*(volatile uint32_t *)&s_count >= part_max) { s_full = 1; break; }   // <= 1024 inserts slip past: there is room
                    old = atomicCAS(&set[h], kHashEmpty, bits);
                    if (old == kHashEmpty) { atomicAdd(&s_count, 1u); break; }
                }
                if (old == bits) break;
                h = (h + 1) & (kPartSlots - 1);
            }
        }
        cnt = cnt_next;
    }
    __syncthreads();
    const uint32_t n = s_count;
    if (s_full || n > part_max) {                              // block-uniform
        if (tid == 0) *overflow = 1;
        return;
    }
    if (tid == 0) s_base = (uint32_t)atomicAdd(n_unique, (int32_t)n);
    if (tid == 0) s_count = 0;
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t q = tid; q < kPartSlots; q += 1024) {
        const uint32_t bits = set[q];
        const uint64_t bal = __ballot(bits != kHashEmpty);
        uint32_t at = 0;
        if (lane == 0 && bal) at = atomicAdd(&s_count, (uint32_t)__popcll(bal));
        at = __shfl(at, 0);
        if (bits != kHashEmpty) values[base + at + __popcll(bal & ((1ull << lane) - 1))] = bits;
    }
}

// (3) one block: the value of rank floor(frac * n) among the n distinct bit patterns (all >= 0 as floats), by four 8-bit
// radix-select passes, most significant byte first
__global__ __launch_bounds__(1024) void select_rank_kernel(const uint32_t *__restrict__ values, const int32_t *__restrict__ n_unique,
                                                            const int32_t *__restrict__ overflow, double frac,
                                                            float *__restrict__ thr)
{
    // one histogram per wave: the values of a scene share their high bytes, and ~10^4 LDS atomics on ONE counter cost
    // the block ~30 us per launch; 16 counters take them side by side, a second step adds the 16 up
    __shared__ uint32_t hist[16][256];
    __shared__ uint32_t total[256];
    __shared__ uint32_t s_prefix, s_rank;
    const int tid = threadIdx.x, wave = tid >> 6;
    if (*overflow) { if (tid == 0) *thr = __builtin_nanf(""); return; }             // block-uniform: the caller sorts
    const uint32_t n = (uint32_t)*n_unique;
    if (tid == 0) {
        const long long r = (long long)floor(frac * (double)n);
        s_rank = (r >= 0 && r < (long long)n) ? (uint32_t)r : 0xFFFFFFFFu;
        s_prefix = 0;
    }
    __syncthreads();
    if (s_rank == 0xFFFFFFFFu) { if (tid == 0) *thr = __builtin_nanf(""); return; }     // block-uniform
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int q = tid; q < 16 * 256; q += 1024) (&hist[0][0])[q] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix, mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (uint32_t i = tid; i < n; i += 1024) {
            const uint32_t v = values[i];
            if ((v & mask) == prefix) atomicAdd(&hist[wave][(v >> shift) & 255u], 1u);
        }
        __syncthreads();
        // bin of the rank: inclusive prefix over the 256 bins by the first four waves (one serial walk by one thread cost
        // ~7 us per pass: 255 dependent LDS reads), then the one bin whose prefix interval holds the rank reports itself
        uint32_t mine = 0, incl = 0;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < 16; ++w) mine += hist[w][tid];
            incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d); if ((tid & 63) >= d) incl += up; }
            if ((tid & 63) == 63) total[tid >> 6] = incl;              // the four waves' sums
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t before = 0;
            for (int w = 0; w < (tid >> 6); ++w) before += total[w];
            incl += before;
            const uint32_t r = s_rank;                                  // read by all before anyone writes (barrier below)
            const bool here = r < incl && r >= incl - mine;
            __syncthreads();
            if (here) { s_rank = r - (incl - mine); s_prefix = prefix | ((uint32_t)tid << shift); }
        } else {
            __syncthreads();
        }
        __syncthreads();
    }
    if (tid == 0) *thr = __uint_as_float(s_prefix);
}

__global__ void ratio_keep_kernel(const int32_t *__restrict__ masked, const int32_t *__restrict__ viewed, int64_t n,
                                  float thr_imm, const float *__restrict__ thr_dev, int use_thr, int64_t nw,
                                  uint64_t *__restrict__ keep)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float thr = thr_dev ? *thr_dev : thr_imm;
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

// ---- depth ingestion ------------------------------------------------------
// raw 16-bit depth (millimetres, source resolution) -> float32 metres at (H, W): `png.astype(f32) / 1000`
// followed by the 2-tap horizontal then vertical passes of a bilinear resize with half-pixel centres and
// edge clamp (reference P:432-436 does this with cv2.imread + cv2.resize; restated on the host in
// io.resize_bilinear_f32, to which this kernel is bit-identical: same float32 operation order, tap tables
// computed by the host in float64).  One thread per output pixel; the 4 source texels are L2-resident.
__global__ void depth_resize_kernel(const uint16_t *__restrict__ src, int hs, int ws, int n_frames,
                                    const int32_t *__restrict__ x0, const int32_t *__restrict__ x1,
                                    const float *__restrict__ ax, const int32_t *__restrict__ y0,
                                    const int32_t *__restrict__ y1, const float *__restrict__ ay, int H, int W,
                                    float scale, float *__restrict__ dst)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= W) return;
    const uint16_t *img = src + (int64_t)f * hs * ws;
    const float a = ax[x], b = ay[y];
    const float one_a = __fsub_rn(1.0f, a), one_b = __fsub_rn(1.0f, b);
    const int xa = x0[x], xb = x1[x];
    auto tap = [&](int yy) {
        const float s0 = __fdiv_rn((float)img[(int64_t)yy * ws + xa], scale);
        const float s1 = __fdiv_rn((float)img[(int64_t)yy * ws + xb], scale);
        return __fadd_rn(__fmul_rn(s0, one_a), __fmul_rn(s1, a));
    };
    const float r0 = tap(y0[y]), r1 = tap(y1[y]);
    dst[((int64_t)f * H + y) * W + x] = __fadd_rn(__fmul_rn(r0, one_b), __fmul_rn(r1, b));
}

__global__ void depth_scale_kernel(const uint16_t *__restrict__ src, int64_t n, float scale, float *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = __fdiv_rn((float)src[i], scale);
}

}  // namespace bff

using namespace bff;

extern "C" int bff_point_values(const int32_t *masked, const int32_t *viewed, int64_t n_points, float *vals, void *stream)
{
    BFF_REQUIRE(n_points >= 0, "bff_point_values: bad size");
    if (n_points == 0) return BFF_OK;
    BFF_REQUIRE(masked && vals, "bff_point_values: null pointer");
    point_values_kernel<<<(unsigned)ceil_div(n_points, 256), 256, 0, as_stream(stream)>>>(masked, viewed, n_points, vals);
    return launched("bff_point_values");
}

extern "C" int bff_select_unique_rank(const float *sorted, int64_t n, double fraction, int32_t *block_scratch,
                                      float *thr, int32_t *n_unique, void *stream)
{
    BFF_REQUIRE(n >= 0 && sorted && block_scratch && thr && n_unique, "bff_select_unique_rank: bad arguments");
    const int nb = (int)ceil_div(n > 0 ? n : 1, kSelBlock);
    if (n > 0) distinct_count_kernel<<<nb, 256, 0, as_stream(stream)>>>(sorted, n, block_scratch);
    distinct_select_kernel<<<1, 1024, 0, as_stream(stream)>>>(sorted, n, block_scratch, n > 0 ? nb : 0, fraction, thr, n_unique);
    return launched("bff_select_unique_rank");
}

extern "C" int bff_ratio_keep(const int32_t *masked, const int32_t *viewed, int64_t n_points, float thr,
                              const float *thr_dev, int32_t use_thr, int64_t nw, uint64_t *keep, void *stream)
{
    BFF_REQUIRE(n_points >= 0 && nw == ceil_div(n_points, 64), "bff_ratio_keep: bad sizes");
    if (n_points == 0) return BFF_OK;
    BFF_REQUIRE(masked && keep, "bff_ratio_keep: null pointer");
    ratio_keep_kernel<<<(unsigned)ceil_div(nw * 64, 256), 256, 0, as_stream(stream)>>>(masked, viewed, n_points, thr, thr_dev, use_thr, nw, keep);
    return launched("bff_ratio_keep");
}

extern "C" int bff_depth_from_u16(const uint16_t *src, int32_t n_frames, int32_t h_src, int32_t w_src,
                                  const int32_t *x0, const int32_t *x1, const float *ax,
                                  const int32_t *y0, const int32_t *y1, const float *ay,
                                  int32_t height, int32_t width, float depth_scale, float *dst, void *stream)
{
    BFF_REQUIRE(n_frames >= 0 && h_src > 0 && w_src > 0 && height > 0 && width > 0 && depth_scale > 0,
                "bff_depth_from_u16: bad sizes");
    if (n_frames == 0) return BFF_OK;
    BFF_REQUIRE(src && dst, "bff_depth_from_u16: null pointer");
    if (h_src == height && w_src == width) {       // cv2.resize to the same size is the identity
        const int64_t n = (int64_t)n_frames * height * width;
        depth_scale_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, as_stream(stream)>>>(src, n, depth_scale, dst);
        return launched("bff_depth_from_u16");
    }
    BFF_REQUIRE(x0 && x1 && ax && y0 && y1 && ay, "bff_depth_from_u16: tap tables required when resizing");
    BFF_LIMIT(height <= 65535 && n_frames <= 65535, "bff_depth_from_u16: grid limits");
    dim3 grid((unsigned)ceil_div(width, 256), (unsigned)height, (unsigned)n_frames);
    depth_resize_kernel<<<grid, 256, 0, as_stream(stream)>>>(src, h_src, w_src, n_frames, x0, x1, ax, y0, y1, ay, height,
                                                             width, depth_scale, dst);
    return launched("bff_depth_from_u16");
}

extern "C" int64_t bff_point_threshold_scratch_words(int64_t n_points)
{
    return ceil_div(n_points > 0 ? n_points : 1, 1024) * kSliceWords + (int64_t)kParts * kPartSlots;
}

static uint32_t g_part_max = kPartMax;
extern "C" int32_t bff_point_threshold_capacity(void) { return (int32_t)g_part_max; }
// test hook: a smaller capacity makes ordinary scenes exercise the overflow -> sorting fallback; 0 restores the default
extern "C" int32_t bff_point_threshold_capacity_set(int32_t cap)
{
    g_part_max = (cap > 0 && (uint32_t)cap < kPartMax) ? (uint32_t)cap : kPartMax;
    return (int32_t)g_part_max;
}

// thr / n_unique as bff_point_values + bff_sort_f32 + bff_select_unique_rank deliver them, from the distinct values
// that occur.  scratch: uint32 [bff_point_threshold_scratch_words(n_points)], needs no clearing.  *overflow (device
// int32, NOT cleared here) is set to 1 when one of the 64 hash partitions holds more distinct values than
// bff_point_threshold_capacity() (scenes with several 10^5 distinct values): thr is then NaN and the caller must take
// the sorting path.  *n_unique is the counter the partitions add to: cleared here (bff_scene_project: by its one fill).
extern "C" int bff_point_threshold_pairs(const int32_t *masked, const int32_t *viewed, int64_t n_points, double fraction,
                                         uint32_t *scratch, float *thr, int32_t *n_unique, int32_t *overflow, void *stream)
{
    BFF_REQUIRE(n_points >= 0 && scratch && thr && n_unique && overflow, "bff_point_threshold_pairs: bad arguments");
    hipStream_t st = as_stream(stream);
    const int64_t n_slices = n_points > 0 ? ceil_div(n_points, 1024) : 0;
    BFF_LIMIT(n_slices < (1ll << 30), "bff_point_threshold_pairs: too many points");
    hipError_t e = zero_async(n_unique, sizeof(int32_t), st);
    if (e != hipSuccess) return fail((int)e, "bff_point_threshold_pairs: memset: %s", hipGetErrorString(e));
    uint32_t *values = scratch + n_slices * kSliceWords;
    if (n_points > 0) {
        BFF_REQUIRE(masked, "bff_point_threshold_pairs: null pointer");
        block_value_sets_kernel<<<(unsigned)n_slices, 1024, 0, st>>>(masked, viewed, n_points, scratch);
        partition_sets_kernel<<<kParts, 1024, 0, st>>>(scratch, (int)n_slices, values, n_unique, overflow, g_part_max);
    }
    select_rank_kernel<<<1, 1024, 0, st>>>(values, n_unique, overflow, fraction, thr);
    return launched("bff_point_threshold_pairs");
}
