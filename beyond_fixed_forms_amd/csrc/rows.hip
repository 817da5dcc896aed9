// Bit-row primitives (include/bff_hip.h: a8-a13, a16-a20).
//
// A boolean row over N points is nw = ceil(N/64) uint64 words.  Set algebra on rows is AND/OR/
// ANDNOT on words, cardinalities are popcounts, the {0,1} matmuls of the reference
// (F @ F.T, projection_2d_to_3d.py:159; mask_1 @ mask_2.T, refinement.py:84) are
// popcount(a & b) accumulated over words -- exact integers, 1/32 of the bytes of the float form.
#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "common.h"

static thread_local hipEvent_t g_merge_start = nullptr, g_merge_stop = nullptr;   // bff_profile_next_merge

namespace bff {

constexpr int kT = 64;        // tile of 64 x 64 row pairs per 256-thread block
constexpr int kKW = 32;       // words staged per step
constexpr int kPitch = kT + 1;

// ---- popcount of rows -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void popcount_rows_kernel(const uint64_t *__restrict__ rows,
                                                             const int32_t *__restrict__ idx, int64_t nw,
                                                             int32_t *__restrict__ area)
{
    __shared__ int part[4];
    const int r = blockIdx.x;
    const uint64_t *row = rows + (int64_t)(idx ? idx[r] : r) * nw;
    int s = 0;
    for (int64_t w = threadIdx.x; w < nw; w += blockDim.x) s += popc64(row[w]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
    if (lane_id() == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) area[r] = part[0] + part[1] + part[2] + part[3];
}

// ---- 64x64 tile of popcount(a_i & b_j) -------------------------------------------------------
// LDS images are [word][row] (pitch 65) so that the 4 rows / 4 columns a thread needs for one word
// are 32 contiguous bytes; thread (ti, tj) of the 16 x 16 thread grid owns rows 4ti..4ti+3 and
// columns 4tj..4tj+3.
__device__ __forceinline__ int64_t ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ void tile_popcount(const uint64_t *__restrict__ a, const int32_t *__restrict__ ia,
                                              int na, int i0, const uint64_t *__restrict__ b,
                                              const int32_t *__restrict__ ib, int nb, int j0, int64_t nw,
                                              int64_t k_begin, int64_t k_end,
                                              uint64_t (*sa)[kPitch], uint64_t (*sb)[kPitch], int acc[4][4])
{
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    const int lk = tid & (kKW - 1), lr = tid >> 5;          // loader: word lk of rows lr, lr+8, ...
    const uint64_t *pa[8];
    const uint64_t *pb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ra = i0 + lr + 8 * q, rb = j0 + lr + 8 * q;
        pa[q] = ra < na ? a + (int64_t)(ia ? ia[ra] : ra) * nw : nullptr;
        pb[q] = rb < nb ? b + (int64_t)(ib ? ib[rb] : rb) * nw : nullptr;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0;
    for (int64_t k0 = k_begin; k0 < k_end; k0 += kKW) {
        const bool kin = k0 + lk < nw;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sa[lk][lr + 8 * q] = (kin && pa[q]) ? pa[q][k0 + lk] : 0;
            sb[lk][lr + 8 * q] = (kin && pb[q]) ? pb[q][k0 + lk] : 0;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < kKW; ++kk) {
            uint64_t av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { av[r] = sa[kk][ti * 4 + r]; bv[r] = sb[kk][tj * 4 + r]; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] += popc64(av[r] & bv[c]);
        }
        __syncthreads();
    }
}

constexpr int kFuseMax = BFF_GROUP_CAP_MAX;      // most groups the device forms by itself

__global__ __launch_bounds__(256) void cross_popcount_kernel(const uint64_t *__restrict__ a,
                                                              const int32_t *__restrict__ ia, int na,
                                                              const uint64_t *__restrict__ b,
                                                              const int32_t *__restrict__ ib, int nb, int64_t nw,
                                                              int64_t k_split, int32_t *__restrict__ inter,
                                                              const int32_t *__restrict__ k_dev, int lim_a, int hole_hi)
{
    __shared__ uint64_t sa[kKW][kPitch], sb[kKW][kPitch];
    // k_dev (optional): only the first *k_dev rows of a (when lim_a) resp. of b's leading block [0, hole_hi) hold data,
    // the rest of those ranges is all zero: tiles that lie entirely in the zero part are skipped (the output is
    // zeroed by the host when the words are split over z; callers never read the skipped entries otherwise)
    if (k_dev) {
        const int kd = *k_dev;
        if (lim_a && (int)blockIdx.y * kT >= kd) return;
        if ((int)blockIdx.x * kT >= kd && (int)(blockIdx.x + 1) * kT <= hole_hi) return;
    }
    // blockIdx.z owns the word range [z*k_split, (z+1)*k_split): small row counts still fill the chip.
    // Partial counts are combined with integer atomics (exact, order independent) into a zeroed matrix.
    int acc[4][4];
    const int i0 = blockIdx.y * kT, j0 = blockIdx.x * kT;
    const int64_t k_begin = (int64_t)blockIdx.z * k_split;
    tile_popcount(a, ia, na, i0, b, ib, nb, j0, nw, k_begin, min(nw, k_begin + k_split), sa, sb, acc);
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = i0 + ti * 4 + r, j = j0 + tj * 4 + c;
            if (i < na && j < nb) {
                if (gridDim.z == 1) inter[(int64_t)i * nb + j] = acc[r][c];
                else if (acc[r][c]) atomicAdd(inter + (int64_t)i * nb + j, acc[r][c]);
            }
        }
}

// ---- row statistics for the block-sparse Gram -------------------------------------------------
// Per row: popcount, occupancy mask over chunks of kCW words, and the mean word position of its set
// bits (sort key that brings rows covering the same region of the -- spatially sorted -- cloud together).
constexpr int kBins = 64;     // histogram bins per row (each ceil(nw/64) words wide)

// 30-bit sort key of a row from its heavy-bin ballot: the first five heavy bins (ascending), 6 bits each, most
// significant first; unused slots = 63.  Rows of one object share the key whatever the view.
__device__ __forceinline__ int64_t heavy_signature(uint64_t heavy)
{
    uint32_t key = 0;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        int b = 63;
        if (heavy) { b = __ffsll((unsigned long long)heavy) - 1; heavy &= heavy - 1; }
        key = (key << 6) | (uint32_t)b;
    }
    return (int64_t)key;
}

__global__ __launch_bounds__(256) void row_stats_kernel(const uint64_t *__restrict__ rows, int64_t nw, int mw,
                                                         int bin_words, int32_t *__restrict__ area,
                                                         int32_t *__restrict__ mean_word,
                                                         uint64_t *__restrict__ cmask, uint32_t *__restrict__ hist,
                                                         int64_t *__restrict__ signature, uint16_t *__restrict__ cpop)
{
    extern __shared__ uint64_t s_cm[];                 // mw words
    __shared__ int part[4];
    __shared__ unsigned long long psum[4];
    __shared__ uint32_t s_hist[kBins];
    const int r = blockIdx.x, tid = threadIdx.x;
    const uint64_t *row = rows + (int64_t)r * nw;
    for (int i = tid; i < mw; i += 256) s_cm[i] = 0;
    if (tid < kBins) s_hist[tid] = 0;
    __syncthreads();
    int s = 0;
    unsigned long long ws = 0;
    for (int64_t w = tid; w < nw; w += 256) {
        const uint64_t v = row[w];
        if (v) {
            const int c = (int)(w / kCW);
            atomicOr((unsigned long long *)&s_cm[c >> 6], 1ull << (c & 63));
            const int pc = popc64(v);
            atomicAdd(&s_hist[(int)(w / bin_words)], (uint32_t)pc);
            s += pc;
            ws += (unsigned long long)pc * (unsigned long long)w;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { s += __shfl_down(s, d); ws += __shfl_down(ws, d); }
    if (lane_id() == 0) { part[tid >> 6] = s; psum[tid >> 6] = ws; }
    __syncthreads();
    for (int i = tid; i < mw; i += 256) cmask[(int64_t)r * mw + i] = s_cm[i];
    if (cpop) {                                    // points per 512-point chunk (second-level bound of the tile pass)
        const int n_chunks = (int)((nw + kCW - 1) / kCW);
        for (int c = tid; c < mw * 64; c += 256) {
            int pc = 0;
            if (c < n_chunks)
#pragma unroll
                for (int k = 0; k < kCW; ++k) { const int64_t w = (int64_t)c * kCW + k; if (w < nw) pc += popc64(row[w]); }
            cpop[(int64_t)r * mw * 64 + c] = (uint16_t)pc;
        }
    }
    const int a_all = part[0] + part[1] + part[2] + part[3];
    if (tid < kBins) {
        hist[(int64_t)r * kBins + tid] = s_hist[tid];
        // bins holding >= 15 % of the row: rows of one object share this signature whatever the view, and
        // stray "bleed" points never enter it.
        const uint64_t heavy = __ballot((uint64_t)s_hist[tid] * 100 >= (uint64_t)a_all * 15 && a_all > 0);
        if (tid == 0) signature[r] = heavy_signature(a_all ? heavy : 0);
    }
    if (tid == 0) {
        const int a = part[0] + part[1] + part[2] + part[3];
        const unsigned long long t = psum[0] + psum[1] + psum[2] + psum[3];
        area[r] = a;
        mean_word[r] = a ? (int32_t)(t / (unsigned long long)a) : 0x7fffffff;   // empty rows sort last
    }
}

// The same statistics when the rows' chunk masks are already known (the sweep flags the chunks it stores
// into): one wave per row, lane l takes the flagged chunks l, l+64, ... and reads only those 64 bytes.
__global__ __launch_bounds__(256) void row_stats_sparse_kernel(const uint64_t *__restrict__ rows, int n_rows, int64_t nw,
                                                                int mw, int bin_words, int32_t *__restrict__ area,
                                                                int32_t *__restrict__ mean_word,
                                                                const uint64_t *__restrict__ cmask,
                                                                uint32_t *__restrict__ hist, int64_t *__restrict__ signature,
                                                                uint16_t *__restrict__ cpop)
{
    __shared__ uint32_t s_hist[4][kBins];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= n_rows) return;                                   // wave-uniform; no block barrier below
    const uint64_t *row = rows + (int64_t)r * nw;
    uint32_t *hs = s_hist[wave];
    hs[lane] = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int s = 0;
    unsigned long long ws = 0;
    for (int i = 0; i < mw; ++i) {
        const uint64_t m = cmask[(int64_t)r * mw + i];
        if ((m >> lane) & 1) {
            const int64_t w0 = ((int64_t)i * 64 + lane) * kCW;
            int in_chunk = 0;
#pragma unroll
            for (int k = 0; k < kCW; ++k) {
                const int64_t w = w0 + k;
                const uint64_t v = w < nw ? row[w] : 0;
                if (v) {
                    const int pc = popc64(v);
                    atomicAdd(&hs[(int)(w / bin_words)], (uint32_t)pc);
                    s += pc;
                    in_chunk += pc;
                    ws += (unsigned long long)pc * (unsigned long long)w;
                }
            }
            if (cpop) cpop[(int64_t)r * mw * 64 + i * 64 + lane] = (uint16_t)in_chunk;     // unflagged chunks: zeroed by the caller
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { s += __shfl_xor(s, d); ws += __shfl_xor(ws, d); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint32_t hv = hs[lane];
    hist[(int64_t)r * kBins + lane] = hv;
    const uint64_t heavy = __ballot((uint64_t)hv * 100 >= (uint64_t)s * 15 && s > 0);
    if (lane == 0) {
        signature[r] = heavy_signature(s ? heavy : 0);
        area[r] = s;
        mean_word[r] = s ? (int32_t)(ws / (unsigned long long)s) : 0x7fffffff;
    }
}

// Undo what the sweep stored: zero exactly the chunks flagged in the rows' occupancy masks (one wave per row), so
// that a zero-filled row arena is all zero again after a scene without touching its other 99 %.
__global__ __launch_bounds__(256) void clear_flagged_chunks_kernel(uint64_t *__restrict__ rows, int n_rows, int64_t nw,
                                                                    const uint64_t *__restrict__ cmask, int mw,
                                                                    const int32_t *__restrict__ veto)
{
    if (veto && *veto) return;                     // the host still needs the rows (general path)
    const int lane = lane_id();
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    uint64_t *row = rows + (int64_t)r * nw;
    for (int i = 0; i < mw; ++i) {
        const uint64_t m = cmask[(int64_t)r * mw + i];
        if ((m >> lane) & 1) {
            const int64_t w0 = ((int64_t)i * 64 + lane) * kCW;
#pragma unroll
            for (int k = 0; k < kCW; ++k)
                if (w0 + k < nw) row[w0 + k] = 0;
        }
    }
}

// Per tile t (rows order[64t .. 64t+63]): tmask[t] = OR of the rows' chunk masks; optionally the sorted,
// packed histogram copy hist_sorted[bin pair][position] (two 16-bit bins per word: coalesced tile loads, one
// v_pk_min_u16 + v_dot2_u32_u16 per two bins), the bin-wise maxima and the smallest non-empty area of the tile.
// 256 threads: thread (k = tid & 63, q = tid >> 6) works on row k of the tile.
__global__ __launch_bounds__(256) void tile_masks_kernel(const uint64_t *__restrict__ cmask,
                                                          const int32_t *__restrict__ order, int n, int mw,
                                                          uint64_t *__restrict__ tmask, const uint32_t *__restrict__ hist,
                                                          uint32_t *__restrict__ hist_sorted, int n_pos,
                                                          const int32_t *__restrict__ area,
                                                          uint32_t *__restrict__ tile_hmax, int32_t *__restrict__ tile_amin,
                                                          const int32_t *__restrict__ label_id,
                                                          int32_t *__restrict__ row_sorted, int32_t *__restrict__ area_sorted,
                                                          int32_t *__restrict__ label_sorted, int32_t *__restrict__ parent_init)
{
    __shared__ uint32_t s_hmax[kBins];
    __shared__ int s_amin;
    const int t = blockIdx.x, tid = threadIdx.x, k = tid & 63, q = tid >> 6;
    const int pos = t * kT + k;
    const int row = pos < n ? (order ? order[pos] : pos) : -1;
    if (parent_init && q == 0 && row >= 0) parent_init[row] = row;     // disjoint-set forest: every row its own root
    // chunk-mask OR: lanes = rows, each wave takes every 4th mask word and OR-reduces it across the wave
    for (int i = q; i < mw; i += 4) {
        uint64_t v = row >= 0 ? cmask[(int64_t)row * mw + i] : 0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v |= __shfl_xor(v, d);
        if (k == 0) tmask[(int64_t)t * mw + i] = v;
    }
    if (!hist_sorted) return;
    if (tid < kBins) s_hmax[tid] = 0;
    if (tid == 0) s_amin = 0x7fffffff;
    __syncthreads();
    // wave q handles bin pairs q, q+4, ...: row k's two bins -> packed word, tile maxima via LDS atomics
    for (int b = q; b < kBins / 2; b += 4) {
        const uint32_t lo = row >= 0 ? hist[(int64_t)row * kBins + 2 * b] : 0;
        const uint32_t hi = row >= 0 ? hist[(int64_t)row * kBins + 2 * b + 1] : 0;
        hist_sorted[(int64_t)b * n_pos + pos] = lo | (hi << 16);
        uint32_t mlo = lo, mhi = hi;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { mlo = max(mlo, (uint32_t)__shfl_xor(mlo, d)); mhi = max(mhi, (uint32_t)__shfl_xor(mhi, d)); }
        if (k == 0) { s_hmax[2 * b] = mlo; s_hmax[2 * b + 1] = mhi; }
    }
    if (q == 0) {
        int a = row >= 0 ? area[row] : 0;
        if (row_sorted) {                          // position-indexed copies: the tile pass loads them coalesced
            row_sorted[pos] = row;
            area_sorted[pos] = a;
            label_sorted[pos] = row >= 0 ? label_id[row] : -1;
        }
        a = a > 0 ? a : 0x7fffffff;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) a = min(a, __shfl_xor(a, d));
        if (k == 0) s_amin = a;
    }
    __syncthreads();
    if (tid < kBins) tile_hmax[(int64_t)t * kBins + tid] = s_hmax[tid];
    if (tid == 0) tile_amin[t] = s_amin;
}

// Upper-triangular tile pairs of the symmetric Gram matrix.  Tile (bi, bj) covers rows
// order[64 bi ..] x order[64 bj ..]; with tile chunk masks it visits only the chunks of kCW words
// that both tiles occupy (four chunks = 32 words per LDS stage), otherwise every word.  The epilogue
// applies the reference's float32 IoU test and emits adjacency words for the tile and its mirror
// image, indexed by position in `order`.
constexpr int kMaxChunks = 4096;     // chunk list capacity (LDS): N <= 4096*512 = 2.1 M points per call
constexpr int kSplitStages = 12;     // split mode: LDS stages per part (a part = ~50 us of tile pass)
constexpr int kMaxParts = 8;         // parts per tile pair
constexpr int kMaxSlots = 512;       // tile pairs that can be split in one call (16 KiB of partial counts each)

__global__ __launch_bounds__(256) void merge_adjacency_kernel(const uint64_t *__restrict__ rows, int n, int64_t nw,
                                                               const int32_t *__restrict__ order,
                                                               const uint64_t *__restrict__ tmask, int mw,
                                                               const uint32_t *__restrict__ hist,
                                                               const int32_t *__restrict__ area,
                                                               const int32_t *__restrict__ label_id, float thr,
                                                               uint64_t *__restrict__ adj, int aw,
                                                               int32_t *__restrict__ inter, int n_tiles)
{
    __shared__ uint64_t sa[kKW][kPitch], sb[kKW][kPitch];
    __shared__ uint8_t flag[kT][kT + 4];
    __shared__ uint16_t clist[kMaxChunks];
    __shared__ int s_cnt;
    // linear upper-triangular index -> (bi <= bj)
    int t = blockIdx.x, bi = 0;
    while (t >= n_tiles - bi) { t -= n_tiles - bi; ++bi; }
    const int bj = bi + t;
    const int i0 = bi * kT, j0 = bj * kT;
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int n_chunks = (int)((nw + kCW - 1) / kCW);

    // ---- can any pair of this tile be adjacent at all?  I(i,j) <= UB = sum over 64 bins of
    // min(hist_i, hist_j); the float32 IoU expression below is monotone non-decreasing in I (a_i + a_j
    // fixed, correctly rounded ops), so "label equal and iou(min(UB, a_i, a_j)) > thr" is a sound
    // superset of the adjacent pairs.  A tile without candidates skips its word loop (all bits 0).
    bool any_candidate = true;
    if (hist) {
        uint32_t (*ha)[kBins] = reinterpret_cast<uint32_t (*)[kBins]>(&sa[0][0]);    // [bin][row], 16 KB each
        uint32_t (*hb)[kBins] = reinterpret_cast<uint32_t (*)[kBins]>(&sb[0][0]);
        {
            const int lane = tid & 63, wv = tid >> 6;
            const int ra = i0 + lane, rb = j0 + lane;
            const uint32_t *ga = ra < n ? hist + (int64_t)(order ? order[ra] : ra) * kBins : nullptr;
            const uint32_t *gb = rb < n ? hist + (int64_t)(order ? order[rb] : rb) * kBins : nullptr;
#pragma unroll
            for (int q = 0; q < kBins / 4; ++q) {
                const int b = wv * (kBins / 4) + q;
                ha[b][lane] = ga ? ga[b] : 0;
                hb[b][lane] = gb ? gb[b] : 0;
            }
        }
        __syncthreads();
        uint32_t ub[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) ub[r][c] = 0;
#pragma unroll 4
        for (int b = 0; b < kBins; ++b) {
            uint32_t av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { av[r] = ha[b][ti * 4 + r]; bv[r] = hb[b][tj * 4 + r]; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) ub[r][c] += min(av[r], bv[c]);
        }
        bool cand = false;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int si = i0 + ti * 4 + r, sj = j0 + tj * 4 + c;
                if (si < n && sj < n) {
                    const int i = order ? order[si] : si, j = order ? order[sj] : sj;
                    const int ai = area[i], aj = area[j];
                    const float fi = (float)min((int)ub[r][c], min(ai, aj));
                    const float iou = __fdiv_rn(fi, (float)ai + (float)aj - fi);
                    cand |= (label_id[i] == label_id[j]) && (iou > thr);
                }
            }
        any_candidate = __syncthreads_or(cand);
    }

    // ---- chunks to visit
    if (tid < kWave) {
        int base = 0;
        for (int m = 0; m < (n_chunks + 63) / 64; ++m) {
            const uint64_t bits = tmask ? (tmask[(int64_t)bi * mw + m] & tmask[(int64_t)bj * mw + m]) : ~0ull;
            const int c = m * 64 + tid;
            const bool on = ((bits >> tid) & 1) && c < n_chunks;
            const uint64_t bal = __ballot(on);
            if (on) clist[base + __popcll(bal & ((1ull << tid) - 1))] = (uint16_t)c;
            base += __popcll(bal);
        }
        if (tid == 0) s_cnt = base;
    }
    __syncthreads();
    const int cnt = any_candidate ? s_cnt : 0;

    int acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0;
    if (cnt) {
        const int lk = tid & (kKW - 1), lr = tid >> 5;      // loader: staged word lk of rows lr, lr+8, ...
        const int slot = lk / kCW, cw = lk % kCW;
        const uint64_t *pa[8];
        const uint64_t *pb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int ra = i0 + lr + 8 * q, rb = j0 + lr + 8 * q;
            pa[q] = ra < n ? rows + (int64_t)(order ? order[ra] : ra) * nw : nullptr;
            pb[q] = rb < n ? rows + (int64_t)(order ? order[rb] : rb) * nw : nullptr;
        }
        for (int g = 0; g < cnt; g += kKW / kCW) {
            const int64_t w = (g + slot < cnt) ? (int64_t)clist[g + slot] * kCW + cw : nw;
            const bool kin = w < nw;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                sa[lk][lr + 8 * q] = (kin && pa[q]) ? pa[q][w] : 0;
                sb[lk][lr + 8 * q] = (kin && pb[q]) ? pb[q][w] : 0;
            }
            __syncthreads();
#pragma unroll 8
            for (int kk = 0; kk < kKW; ++kk) {
                uint64_t av[4], bv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { av[r] = sa[kk][ti * 4 + r]; bv[r] = sb[kk][tj * 4 + r]; }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c] += popc64(av[r] & bv[c]);
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int si = i0 + ti * 4 + r, sj = j0 + tj * 4 + c;
            bool ok = false;
            if (si < n && sj < n) {
                const int i = order ? order[si] : si, j = order ? order[sj] : sj;
                const float fi = (float)acc[r][c];
                const float uni = (float)area[i] + (float)area[j] - fi;
                const float iou = __fdiv_rn(fi, uni);           // 0/0 -> NaN -> compares false
                ok = (label_id[i] == label_id[j]) && (iou > thr);
                if (inter) {
                    inter[(int64_t)i * n + j] = acc[r][c];
                    inter[(int64_t)j * n + i] = acc[r][c];
                }
            }
            flag[ti * 4 + r][tj * 4 + c] = ok ? 1 : 0;
        }
    __syncthreads();
    if (tid < kT) {
        const int i = i0 + tid;
        if (i < n) {
            uint64_t w = 0;
            for (int c = 0; c < kT; ++c) w |= (uint64_t)flag[tid][c] << c;
            adj[(int64_t)i * aw + bj] = w;
        }
    } else if (tid < 2 * kT && bi != bj) {
        const int c = tid - kT, j = j0 + c;
        if (j < n) {
            uint64_t w = 0;
            for (int r = 0; r < kT; ++r) w |= (uint64_t)flag[r][c] << r;
            adj[(int64_t)j * aw + bi] = w;
        }
    }
}

// ---- components without an adjacency matrix: union-find in the tile epilogue -----------------------
// parent[] is a disjoint-set forest over ROW indices (roots point to themselves, links go to the
// smaller index).  Reads bypass L1 (agent-scope relaxed atomics) so every wave sees links made by
// other CUs; a stale view can only make a tile do work it could have skipped, never change the result:
// once two rows share a root they are connected for good, and links are made with compare-and-swap.
__device__ __forceinline__ int uf_find(int32_t *parent, int x)
{
    int p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        const int g = __hip_atomic_load(parent + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // path halving: point x at its grandparent.  g is an ancestor of x, so the forest stays a forest
        // whatever other waves do meanwhile (links only ever go to smaller indices).
        if (g != p) __hip_atomic_store(parent + x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
        p = g;
    }
    return x;
}

__device__ __forceinline__ void uf_union(int32_t *parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }            // a > b: hang the larger root under the smaller
        int expected = a;
        if (__hip_atomic_compare_exchange_strong(parent + a, &expected, b, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT))
            return;
    }
}

__global__ void uf_init_kernel(int32_t *parent, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = i;
}

__global__ void uf_flatten_kernel(int32_t *parent, int n, int32_t *comp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) comp[i] = uf_find(parent, i);                      // = smallest row index of the component
}

// Skeleton pass: one wave per pair of rows `stride` apart in the sorted order.  Rows with the same signature
// show the same object, so a handful of strides links most of every large component before the tile pass
// starts, which then finds many of its possible edges already connected.  Exact test, same as the tiles.
__global__ __launch_bounds__(256) void uf_skeleton_kernel(const uint64_t *__restrict__ rows, int n, int64_t nw,
                                                           const int32_t *__restrict__ order,
                                                           const uint64_t *__restrict__ cmask, int mw,
                                                           const int32_t *__restrict__ area,
                                                           const int32_t *__restrict__ label_id, float thr,
                                                           int32_t *__restrict__ parent, int n_strides)
{
    const int lane = lane_id();
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int stride = blockIdx.y < 3 ? blockIdx.y + 1 : (blockIdx.y == 3 ? 5 : blockIdx.y == 4 ? 8 : 13 + 8 * (blockIdx.y - 5));
    (void)n_strides;
    if (p + stride >= n) return;
    const int i = order ? order[p] : p, j = order ? order[p + stride] : p + stride;
    // everything the wave needs about the two rows is requested at once (one memory round trip): labels, areas
    // and the occupancy words (lane m holds word m of both rows; mw <= 64 = kMaxChunks / 64)
    const int li = label_id[i], lj = label_id[j];
    const int ai = area[i], aj = area[j];
    const uint64_t shared_chunks = lane < mw ? (cmask[(int64_t)i * mw + lane] & cmask[(int64_t)j * mw + lane]) : 0;
    if (li != lj) return;
    const uint64_t *ri = rows + (int64_t)i * nw, *rj = rows + (int64_t)j * nw;
    int acc = 0;
    const int sub = lane >> 3, cw = lane & 7;                        // 8 chunks x 8 words per step
    for (int m = 0; m < mw; ++m) {                                   // wave-uniform walk over shared chunks
        uint64_t bits = __shfl(shared_chunks, m);
        while (bits) {
            int mine = -1;                                           // the sub-th set bit of this batch, if any
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (bits) {
                    const int c = __ffsll((unsigned long long)bits) - 1;
                    bits &= bits - 1;
                    if (k == sub) mine = m * 64 + c;
                }
            }
            if (mine >= 0) {
                const int64_t w = (int64_t)mine * kCW + cw;
                if (w < nw) acc += popc64(ri[w] & rj[w]);
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d);
    if (lane == 0) {
        const float fi = (float)acc;
        const float iou = __fdiv_rn(fi, (float)ai + (float)aj - fi);
        if (iou > thr) uf_union(parent, i, j);
    }
}

// Tile pairs are enumerated diagonal-first (|bi - bj| = 0, 1, 2, ...): with rows clustered by signature
// the first tiles discover the large components, and later tiles find most of their possible edges
// already inside one component and skip their word loop.  index t -> (bi, d = bj - bi).
__device__ __forceinline__ void tile_pair_of(int t, int n_tiles, int &bi, int &d)
{
    // offset of diagonal d: d * n_tiles - d (d - 1) / 2; invert with a float estimate and fix up
    const double a = 2.0 * n_tiles + 1.0;
    int dd = (int)((a - sqrt(a * a - 8.0 * (double)t)) * 0.5);
    dd = max(0, min(dd, n_tiles - 1));
    auto off = [&](int q) { return (int64_t)q * n_tiles - (int64_t)q * (q - 1) / 2; };
    while (dd > 0 && off(dd) > t) --dd;
    while (dd + 1 < n_tiles && off(dd + 1) <= t) ++dd;
    d = dd;
    bi = (int)(t - off(dd));
}

// Tile-level quick reject for every tile pair, ahead of the tile pass: the pairs that can hold an edge are
// appended to `list` (wave-aggregated, so the list keeps the diagonal-first order up to wave granularity).
// Every pair of rows of tiles (A, B) has I <= u = sum_b min(maxA[b], maxB[b]) and a_i + a_j >= aminA + aminB,
// hence IoU = I / (a_i + a_j - I) <= u / (aminA + aminB - u) whenever that denominator is positive (float32
// evaluation is monotone in both arguments); otherwise no conclusion.
__global__ __launch_bounds__(256) void tile_pair_filter_kernel(const uint32_t *__restrict__ tile_hmax,
                                                                const int32_t *__restrict__ tile_amin, int n_tiles,
                                                                int total, float thr, int32_t *__restrict__ list,
                                                                int32_t *__restrict__ count)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    bool possible = false;
    if (t < total) {
        int bi, d;
        tile_pair_of(t, n_tiles, bi, d);
        const int bj = bi + d;
        const int amin_a = tile_amin[bi], amin_b = tile_amin[bj];
        // a tile of empty rows has no edges at all -- unless the threshold is negative: an empty row then links to
        // every non-empty row of its label (IoU 0 > thr, P:149-166), and the bound below holds with amin = INT_MAX
        // too (quotient >= 0 > thr)
        if (0.0f > thr || (amin_a != 0x7fffffff && amin_b != 0x7fffffff)) {
            const uint4 *ha = reinterpret_cast<const uint4 *>(tile_hmax + (int64_t)bi * kBins);
            const uint4 *hb = reinterpret_cast<const uint4 *>(tile_hmax + (int64_t)bj * kBins);
            uint32_t u = 0;
#pragma unroll 4
            for (int q = 0; q < kBins / 4; ++q) {
                const uint4 x = ha[q], y = hb[q];
                u += min(x.x, y.x) + min(x.y, y.y) + min(x.z, y.z) + min(x.w, y.w);
            }
            const float fi = (float)u;
            const float den = (float)amin_a + (float)amin_b - fi;
            possible = !(den > 0.0f) || (__fdiv_rn(fi, den) > thr);
        }
    }
    const uint64_t bal = __ballot(possible);
    if (!bal) return;
    const int lane = lane_id();
    int base = 0;
    if (lane == 0) base = atomicAdd(count, __popcll(bal));
    base = __shfl(base, 0);
    if (possible) list[base + __popcll(bal & ((1ull << lane) - 1))] = t;
}

// Row-level bound for the tile pairs that survived the tile-level one: one wave per listed pair, lane k = row k of
// tile A (then of tile B).  Row i of A against the bin-wise maxima of tile B: I(i, j) <= u_i = sum_b min(h_i[b],
// maxB[b]) for every j of B and a_j >= aminB, so IoU(i, j) <= min(u_i, a_i) / (a_i + aminB - min(u_i, a_i)) whenever the
// denominator is positive (same monotone float32 expression as the exact test).  Rows that fail cannot have an edge
// into the other tile.  Pairs in which some row of A and some row of B pass are appended to list2 together with
// the two 64-bit pass masks; most pairs end here, at the cost of one wave instead of a 256-thread block.
__global__ __launch_bounds__(256) void tile_pair_rows_kernel(const uint32_t *__restrict__ hist, int n_pos,
                                                              const int32_t *__restrict__ area_sorted,
                                                              const uint32_t *__restrict__ tile_hmax,
                                                              const int32_t *__restrict__ tile_amin, int n_tiles, float thr,
                                                              const int32_t *__restrict__ list1,
                                                              const int32_t *__restrict__ count1,
                                                              int32_t *__restrict__ list2, uint64_t *__restrict__ pass2,
                                                              int32_t *__restrict__ count2,
                                                              const uint64_t *__restrict__ tmask, int mw,
                                                              int32_t *__restrict__ part2, int32_t *__restrict__ n_slots,
                                                              int list_cap)
{
    const int lane = lane_id();
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= *count1) return;                                      // wave-uniform
    const int t = list1[e];
    int bi, d;
    tile_pair_of(t, n_tiles, bi, d);
    const int bj = bi + d;
    constexpr int kBP = kBins / 2;
    // lane b holds bin b of both tiles' maxima; read back per bin pair with a wave broadcast
    const uint32_t hmA = tile_hmax[(int64_t)bi * kBins + lane], hmB = tile_hmax[(int64_t)bj * kBins + lane];
    const int aA = area_sorted[bi * kT + lane], aB = area_sorted[bj * kT + lane];
    const int aminA = tile_amin[bi], aminB = tile_amin[bj];
    uint32_t uA = 0, uB = 0;
#pragma unroll                                    // fully: the broadcasts below become v_readlane with constant lanes
    for (int b = 0; b < kBP; ++b) {
        const uint32_t wa = hist[(int64_t)b * n_pos + bi * kT + lane];
        const uint32_t wb = hist[(int64_t)b * n_pos + bj * kT + lane];
        const uint32_t mB0 = __shfl(hmB, 2 * b), mB1 = __shfl(hmB, 2 * b + 1);
        const uint32_t mA0 = __shfl(hmA, 2 * b), mA1 = __shfl(hmA, 2 * b + 1);
        uA += min(wa & 0xffffu, mB0) + min(wa >> 16, mB1);
        uB += min(wb & 0xffffu, mA0) + min(wb >> 16, mA1);
    }
    // padding positions carry area 0 and an all-zero histogram: with thr >= 0 they fail (0/x or NaN), with thr < 0
    // the tile pass ignores them by their row index (-1)
    auto passes = [&](uint32_t u, int a_self, int a_other) {
        const float fi = (float)min((int)u, a_self);
        const float den = (float)a_self + (float)a_other - fi;
        return !(den > 0.0f) || (__fdiv_rn(fi, den) > thr);
    };
    const uint64_t pa = __ballot(passes(uA, aA, aminB));
    const uint64_t pb = __ballot(passes(uB, aB, aminA));
    if (!pa || !pb) return;
    // Tile pairs that share many chunks are the longest blocks of the tile pass (up to the whole cloud: 100 LDS
    // stages); their chunk list is dealt to several blocks (kSplitStages stages each, at most kMaxParts) whose
    // partial intersections meet in a scratch slot (merge_tile_pair, split mode).
    int shared = 0;
    if (tmask && lane < mw) shared = __popcll(tmask[(int64_t)bi * mw + lane] & tmask[(int64_t)bj * mw + lane]);
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) shared += __shfl_xor(shared, dd);
    if (lane == 0) {
        int n_parts = 1, slot = 0;
        if (part2 && shared >= 2 * kSplitStages * (kKW / kCW)) {
            n_parts = min(kMaxParts, shared / (kSplitStages * (kKW / kCW)));
            slot = atomicAdd(n_slots, 1);
            if (slot >= kMaxSlots) n_parts = 1;                    // out of slots: one block, as before
        }
        const int at = atomicAdd(count2, n_parts);
        if (at + n_parts > list_cap) return;                       // cannot happen with the caps chosen by the host
        for (int p = 0; p < n_parts; ++p) {
            list2[at + p] = t;
            pass2[2 * (at + p)] = pa;
            pass2[2 * (at + p) + 1] = pb;
            if (part2) part2[at + p] = n_parts > 1 ? ((slot << 8) | (n_parts << 4) | p) : 0;
        }
    }
}

// diagnostics only: thread 0 adds the cycles since *t0 to diag[slot] (>> 6 to stay inside int32) and restarts the clock
__device__ __forceinline__ void diag_lap(int32_t *diag, int slot, long long *t0)
{
    if (diag && threadIdx.x == 0) {
        const long long now = (long long)__builtin_readcyclecounter();
        atomicAdd(diag + slot, (int)((now - *t0) >> 6));
        *t0 = now;
    }
}

// Disjoint sets over the 128 rows of ONE tile pair, in LDS (ids 0..63 = rows of A, 64..127 = rows of B; links go
// to the smaller id).  They start from the global forest (rows that share a global root share a local set) and
// absorb every edge the block proves, so that (a) a pair whose rows have meanwhile become connected inside the
// tile is dropped without finishing its intersection and (b) only the edges that merge two local sets -- at most
// 127 per tile pair -- are pushed into the global forest with compare-and-swap.
__device__ __forceinline__ int local_find(volatile int *lid, int x)
{
    int p = lid[x];
    while (p != x) { x = p; p = lid[x]; }
    return x;
}

__device__ __forceinline__ bool local_union(int *lid, int a, int b)
{
    for (;;) {
        a = local_find(lid, a);
        b = local_find(lid, b);
        if (a == b) return false;
        if (a < b) { const int t = a; a = b; b = t; }              // a > b: hang a under b
        if (atomicCAS(&lid[a], a, b) == a) return true;
    }
}

__shared__ int g_block_info[2];        // [0] candidate pairs | shared chunks << 16, [1] LDS stages executed (diagnostics)

template <bool kDiag>
__device__ __forceinline__ void merge_tile_pair(int t, uint64_t pass_a, uint64_t pass_b,
                                                const uint64_t *__restrict__ rows, int n, int64_t nw,
                                                const uint64_t *__restrict__ tmask, int mw,
                                                const uint32_t *__restrict__ hist, int n_pos,
                                                const int32_t *__restrict__ row_sorted,
                                                const int32_t *__restrict__ area_sorted,
                                                const int32_t *__restrict__ label_sorted, float thr,
                                                int32_t *__restrict__ parent, int n_tiles,
                                                int32_t *__restrict__ diag, const uint16_t *__restrict__ cpop,
                                                int part_info, int32_t *__restrict__ partial, int32_t *__restrict__ arrive)
{
    // split mode (part_info != 0): this block is part `part` of `n_parts` of the tile pair and counts only every
    // n_parts-th LDS stage of the chunk list.  All parts must agree on the candidate pairs, so those come from the
    // static bounds alone (not from the forest, which changes while the parts run), nothing is settled early, and the
    // partial counts are added into the pair's scratch slot; the part that arrives last reads the sums and decides.
    const bool split = part_info != 0;
    const int part = part_info & 15, n_parts = split ? (part_info >> 4) & 15 : 1, pslot = part_info >> 8;
    __shared__ uint64_t sa[kKW][kPitch], sb[kKW][kPitch];
    __shared__ uint16_t clist[kMaxChunks];
    __shared__ int s_cnt;
    __shared__ int rowA[kT], rowB[kT], rootA[kT], rootB[kT];
    __shared__ int areaA[kT], areaB[kT], labA[kT], labB[kT];       // fetched once per tile pair
    __shared__ int lid[2 * kT];                                    // local disjoint sets, see above
    int bi, d;
    tile_pair_of(t, n_tiles, bi, d);
    const int bj = bi + d;
    const int i0 = bi * kT, j0 = bj * kT;
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int n_chunks = (int)((nw + kCW - 1) / kCW);
    long long t_lap = kDiag ? (long long)__builtin_readcyclecounter() : 0;     // kDiag: counters + phase clocks

    // one round trip: position-indexed row / area / label of the 128 rows, then the roots of the rows that can
    // still have an edge into the other tile (the others never enter a candidate pair)
    if (tid < kT) {
        const int row = row_sorted[i0 + tid];
        const bool live = row >= 0 && ((pass_a >> tid) & 1);
        rowA[tid] = row;
        areaA[tid] = area_sorted[i0 + tid];
        labA[tid] = label_sorted[i0 + tid];
        rootA[tid] = live ? (split ? row : uf_find(parent, row)) : -1;
    } else if (tid < 2 * kT) {
        const int k = tid - kT;
        const int row = row_sorted[j0 + k];
        const bool live = row >= 0 && ((pass_b >> k) & 1);
        rowB[k] = row;
        areaB[k] = area_sorted[j0 + k];
        labB[k] = label_sorted[j0 + k];
        rootB[k] = live ? (split ? row : uf_find(parent, row)) : -2;
    }
    // histogram bound (see merge_adjacency_kernel): possible edges only.  hist holds two 16-bit bins per
    // word, so one v_pk_min_u16 + one v_dot2_u32_u16 accumulates two bins of sum_b min(hist_i, hist_j).
    constexpr int kBP = kBins / 2;
    uint32_t (*ha)[kT] = reinterpret_cast<uint32_t (*)[kT]>(&sa[0][0]);      // [bin pair][row], 8 KB each
    uint32_t (*hb)[kT] = reinterpret_cast<uint32_t (*)[kT]>(&sb[0][0]);
    {
        const int lane = tid & 63, wv = tid >> 6;          // 256-B coalesced rows of the sorted histogram
#pragma unroll
        for (int q = 0; q < kBP / 4; ++q) {
            const int b = wv * (kBP / 4) + q;
            ha[b][lane] = hist[(int64_t)b * n_pos + i0 + lane];
            hb[b][lane] = hist[(int64_t)b * n_pos + j0 + lane];
        }
    }
    __syncthreads();
    if (kDiag) diag_lap(diag, 4, &t_lap);                                     // rows, roots, histogram staging
    // local sets start from the global forest: a live row joins the first live row of the tile pair with its root
    if (tid < 2 * kT) {                            // waves 0 and 1; loops without early exit: the LDS reads pipeline
        const int mine = tid < kT ? rootA[tid] : rootB[tid - kT];
        int first = tid;
#pragma unroll 16
        for (int q = 0; q < kT; ++q)
            if (rootA[q] == mine && q < first) first = q;
        if (tid >= kT) {
#pragma unroll 16
            for (int q = 0; q < kT; ++q)
                if (rootB[q] == mine && kT + q < first) first = kT + q;
        }
        lid[tid] = mine >= 0 ? first : tid;        // roots of rows that cannot have an edge are -1 / -2: sets of their own
    }
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    uint32_t ub[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) ub[r][c] = 0;
    const us2 ones = {1, 1};
    const uint32_t mine_a = (uint32_t)(pass_a >> (ti * 4)) & 0xFu;          // 4 flags each
    const uint32_t mine_b = (uint32_t)(pass_b >> (tj * 4)) & 0xFu;
    if (mine_a && mine_b)
#pragma unroll 4
    for (int b = 0; b < kBP; ++b) {
        uint32_t av[4], bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { av[r] = ha[b][ti * 4 + r]; bv[r] = hb[b][tj * 4 + r]; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const us2 m = __builtin_elementwise_min(__builtin_bit_cast(us2, av[r]), __builtin_bit_cast(us2, bv[c]));
                ub[r][c] = __builtin_amdgcn_udot2(m, ones, ub[r][c], false);
            }
    }
    unsigned cand = 0;                                             // bit 4r+c: pair still needs the exact test
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = rowA[ti * 4 + r], j = rowB[tj * 4 + c];
            if (((mine_a >> r) & 1) && ((mine_b >> c) & 1) && i >= 0 && j >= 0 && i != j &&
                rootA[ti * 4 + r] != rootB[tj * 4 + c] && (d > 0 || ti * 4 + r < tj * 4 + c)) {
                const int ai = areaA[ti * 4 + r], aj = areaB[tj * 4 + c];
                const float fi = (float)min((int)ub[r][c], min(ai, aj));
                const float iou = __fdiv_rn(fi, (float)ai + (float)aj - fi);
                if ((labA[ti * 4 + r] == labB[tj * 4 + c]) && (iou > thr)) cand |= 1u << (4 * r + c);
            }
        }
    int any_candidate = __syncthreads_or(cand != 0);
    if (kDiag) diag_lap(diag, 5, &t_lap);                                     // per-pair histogram bound
    if (!any_candidate) return;                                    // block-uniform

    __shared__ uint16_t plist[kT * kT];
    __shared__ int wbase[4];
    const int lane = tid & 63, wv = tid >> 6;
    if (tid < kWave) {                                             // chunks both tiles occupy
        int base = 0;
        for (int m = 0; m < (n_chunks + 63) / 64; ++m) {
            const uint64_t bits = tmask ? (tmask[(int64_t)bi * mw + m] & tmask[(int64_t)bj * mw + m]) : ~0ull;
            const int c = m * 64 + tid;
            const bool on = ((bits >> tid) & 1) && c < n_chunks;
            const uint64_t bal = __ballot(on);
            if (on) clist[base + __popcll(bal & ((1ull << tid) - 1))] = (uint16_t)c;
            base += __popcll(bal);
        }
        if (tid == 0) s_cnt = base;
    }
    __syncthreads();
    // ---- second-level bound, same expression with the 512-point chunks both tiles occupy as the bins:
    // I(i, j) <= sum over shared chunks of min(points of i in the chunk, points of j in the chunk).  A bin of the
    // first bound spans several thousand points; objects that are neighbours in space (or two groups of views of one
    // object) share bins but far fewer points per chunk -- most surviving non-edges are decided here, for 2 bytes
    // per (row, chunk) instead of the chunk's 64 bytes.
    if (cpop && tmask) {
        uint32_t (*pa2)[kT] = reinterpret_cast<uint32_t (*)[kT]>(&sa[0][0]);      // [chunk pair][row]: two chunks per word
        uint32_t (*pb2)[kT] = reinterpret_cast<uint32_t (*)[kT]>(&sb[0][0]);
        constexpr int kPairsPerStep = 64;                          // 128 chunks per step (16 KB per side)
        const int cnt2 = s_cnt;
        const int64_t cstride = (int64_t)mw * 64;
        uint32_t ub2[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) ub2[r][c] = 0;
        for (int c0 = 0; c0 < cnt2; c0 += 2 * kPairsPerStep) {
            const int n_here = min(2 * kPairsPerStep, cnt2 - c0), np_here = (n_here + 1) / 2;
            // thread (row k of A or B, strided over chunk pairs): 2-byte gathers along the row's chunk table
            for (int q = tid; q < np_here * 2 * kT; q += 256) {
                const int kp = q / (2 * kT), rr = q % (2 * kT);
                const int row = rr < kT ? rowA[rr] : rowB[rr - kT];
                uint32_t v = 0;
                if (row >= 0) {
                    const uint16_t *t = cpop + (int64_t)row * cstride;
                    const int s0 = c0 + 2 * kp;
                    v = t[clist[s0]];
                    if (s0 + 1 < cnt2) v |= (uint32_t)t[clist[s0 + 1]] << 16;
                }
                if (rr < kT) pa2[kp][rr] = v; else pb2[kp][rr - kT] = v;
            }
            __syncthreads();
            if (cand)
                for (int kp = 0; kp < np_here; ++kp) {
                    uint32_t av[4], bv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { av[r] = pa2[kp][ti * 4 + r]; bv[r] = pb2[kp][tj * 4 + r]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const us2 m = __builtin_elementwise_min(__builtin_bit_cast(us2, av[r]), __builtin_bit_cast(us2, bv[c]));
                            ub2[r][c] = __builtin_amdgcn_udot2(m, ones, ub2[r][c], false);
                        }
                }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (cand & (1u << (4 * r + c))) {
                    const int ai = areaA[ti * 4 + r], aj = areaB[tj * 4 + c];
                    const float fi = (float)min((int)ub2[r][c], min(ai, aj));
                    const float iou = __fdiv_rn(fi, (float)ai + (float)aj - fi);
                    if (!(iou > thr)) cand &= ~(1u << (4 * r + c));
                }
        any_candidate = __syncthreads_or(cand != 0);
        if (kDiag) diag_lap(diag, 11, &t_lap);                                // chunk-level bound
        if (!any_candidate) return;                                // block-uniform
    }

    // ---- compact the candidate pairs of the tile: (row index in A) << 8 | (row index in B)
    const int mine_n = __popc(cand);
    int incl = mine_n;
#pragma unroll
    for (int q = 1; q < 64; q <<= 1) { const int up = __shfl_up(incl, q); if (lane >= q) incl += up; }
    if (lane == 63) wbase[wv] = incl;
    __shared__ unsigned long long s_live[2];                       // rows that still take part in a candidate pair
    if (tid < 2) s_live[tid] = 0;
    __syncthreads();
    int pos = incl - mine_n;
    for (int q = 0; q < wv; ++q) pos += wbase[q];
    const int n_pairs = wbase[0] + wbase[1] + wbase[2] + wbase[3];
    if (tid == 0) g_block_info[0] = n_pairs | (s_cnt << 16);      // timeline diagnostics (mode 2) read this
    // only the words of rows in a candidate pair are fetched below (the others stage as zeros)
    {
        unsigned cc = cand;
        unsigned ra4 = 0, rb4 = 0;                                 // which of this thread's 4 + 4 rows appear
        while (cc) {
            const int q = __ffs(cc) - 1;
            cc &= cc - 1;
            ra4 |= 1u << (q >> 2);
            rb4 |= 1u << (q & 3);
            plist[pos++] = (uint16_t)(((ti * 4 + (q >> 2)) << 8) | (tj * 4 + (q & 3)));
        }
        if (ra4) atomicOr(&s_live[0], (unsigned long long)ra4 << (ti * 4));
        if (rb4) atomicOr(&s_live[1], (unsigned long long)rb4 << (tj * 4));
    }
    __syncthreads();
    const uint64_t live_a = s_live[0], live_b = s_live[1];
    const int cnt = s_cnt;
    if (kDiag) {                                                   // diagnostics only
        if (tid == 0) { atomicAdd(diag + 0, 1); atomicAdd(diag + 1, cnt); atomicAdd(diag + 2, n_pairs); }
    }
    const int lk = tid & (kKW - 1), lr = tid >> 5;
    const int slot = lk / kCW, cw = lk % kCW;
    const uint64_t *pa[8];
    const uint64_t *pb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ra = rowA[lr + 8 * q], rb = rowB[lr + 8 * q];
        pa[q] = (ra >= 0 && ((live_a >> (lr + 8 * q)) & 1)) ? rows + (int64_t)ra * nw : nullptr;
        pb[q] = (rb >= 0 && ((live_b >> (lr + 8 * q)) & 1)) ? rows + (int64_t)rb * nw : nullptr;
    }
    // Staging is software pipelined: the words of stage g+1 are fetched into registers while stage g is
    // combined out of LDS, so a stage costs max(load latency, compute) instead of their sum.
    uint64_t ra[8], rb[8];
    auto fetch = [&](int g) {
        const int64_t w = (g + slot < cnt) ? (int64_t)clist[g + slot] * kCW + cw : nw;
        const bool kin = w < nw;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            ra[q] = (kin && pa[q]) ? pa[q][w] : 0;
            rb[q] = (kin && pb[q]) ? pb[q][w] : 0;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < 8; ++q) { sa[lk][lr + 8 * q] = ra[q]; sb[lk][lr + 8 * q] = rb[q]; }
    };
    // One pair against the intersection counted SO FAR (final = the chunk loop has ended): returns true when the
    // pair is settled.  Already in one local set: nothing to learn.  iou(partial) > thr: the final intersection is
    // at least the partial one and the float32 expression is monotone in it, so the edge exists (P:149-166) --
    // record it now (the edges that merge two local sets are queued for the global forest, see flush).
    // Otherwise keep counting; after the last chunk the count is exact and decides.
    __shared__ int edge_cnt;
    __shared__ uint8_t edge_a[2 * kT], edge_b[2 * kT];             // at most 127 local unions can succeed
    if (tid == 0) edge_cnt = 0;                                    // ordered before its first use by the barriers below
    auto settle = [&](int ia, int jb, int inter, bool final) -> bool {
        if (local_find(lid, ia) == local_find(lid, kT + jb)) return true;
        const float fi = (float)inter;
        const float iou = __fdiv_rn(fi, (float)areaA[ia] + (float)areaB[jb] - fi);
        if (iou > thr) {                                                         // labels already equal
            if (local_union(lid, ia, kT + jb)) {
                const int k = atomicAdd(&edge_cnt, 1);
                edge_a[k] = (uint8_t)ia;
                edge_b[k] = (uint8_t)jb;
            }
            return true;
        }
        return final;
    };
    int flushed = 0;
    auto flush = [&]() {                                           // call after a barrier that follows the settle phase
        const int n_edges = edge_cnt;
        const int k = flushed + tid;
        if (k < n_edges) {
            uf_union(parent, rowA[edge_a[k]], rowB[edge_b[k]]);
            if (kDiag) atomicAdd(diag + 3, 1);
        }
        flushed = n_edges;
    };
    constexpr int kStep = kKW / kCW;                               // chunks per stage
    // pair-list path: at most 7 pairs per thread (1792 of the 4096).  3 was the first choice; interleaved A/B at config 2
    // (library variants side by side, BFF_HIP_LIB): tile pass alone 0.34 ms with 3, 0.30 with 6, 0.283-0.291 with 7, 0.31-0.33
    // with 10 (a pair-list stage costs ~4 us at 3 pairs per thread and grows with them, a 4x4-block stage ~11 us);
    // 8 and more need __launch_bounds__(256, 3) to stay at three blocks per CU.  Staging tile B permuted so that the 4x4
    // pass reads 16 consecutive words instead of 16 words 32 B apart was measured too: 0.34 ms (slower).
    constexpr int kSparse = 7;
    // settle pairs every `check_every` stages: 2 at first; a settle phase that closes no pair doubles the interval
    // (tiles between two groups of one object never settle early: their phases would cost as much as the counting)
    int check_every = 2, next_check = 2;
    if (kDiag) diag_lap(diag, 6, &t_lap);                                     // pair / chunk lists
    const int g_first = part * kStep, g_step = n_parts * kStep;    // split mode: every n_parts-th stage is this block's
    __shared__ int s_last;
    // split mode, after the chunk loop: add this part's counts to the slot; the last part to arrive takes the sums.
    // Only read-modify-write atomics touch the slot (they are coherent across the chip's L2s); every thread's adds
    // are complete (agent-scope fence) before thread 0 takes the ticket.
    auto arrive_last = [&]() -> bool {
        __threadfence();
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(arrive + pslot, 1) == n_parts - 1;
        __syncthreads();
        return s_last != 0;                                        // block-uniform
    };
    int32_t *my_partial = split ? partial + (int64_t)pslot * (kT * kT) : nullptr;
    if (g_first < cnt) fetch(g_first);
    if (n_pairs <= kSparse * 256) {
        // few candidates: accumulate only those pairs (2 LDS reads per pair word)
        int pi[kSparse], pj[kSparse], accs[kSparse];
        unsigned open = 0;                                         // bit q: pair q of this thread is undecided
#pragma unroll
        for (int q = 0; q < kSparse; ++q) {
            const int p = tid + q * 256;
            const int code = p < n_pairs ? plist[p] : 0;
            pi[q] = code >> 8; pj[q] = code & 255; accs[q] = 0;
            if (p < n_pairs) open |= 1u << q;
        }
        int stages_done = 0;
        for (int g = g_first; g < cnt; g += g_step) {
            stage();
            if (tid == 0) g_block_info[1] += 1;
            __syncthreads();
            if (g + g_step < cnt) fetch(g + g_step);
#pragma unroll
            for (int q = 0; q < kSparse; ++q)
                if ((open >> q) & 1) {
                    int a2 = 0;
#pragma unroll 8
                    for (int kk = 0; kk < kKW; ++kk) a2 += popc64(sa[kk][pi[q]] & sb[kk][pj[q]]);
                    accs[q] += a2;
                }
            const bool last = g + g_step >= cnt;
            if (split) {
                __syncthreads();
            } else if (++stages_done == next_check || last) {
                const unsigned before_open = open;
#pragma unroll
                for (int q = 0; q < kSparse; ++q)
                    if (((open >> q) & 1) && settle(pi[q], pj[q], accs[q], last)) open &= ~(1u << q);
                const int any_open = __syncthreads_or(open != 0);
                const int progress = __syncthreads_or(open != before_open);
                flush();
                if (!any_open) break;                              // every pair settled: the rest of the chunks is moot
                if (!progress) check_every *= 2;
                next_check = stages_done + check_every;
            } else {
                __syncthreads();
            }
        }
        if (split) {
#pragma unroll
            for (int q = 0; q < kSparse; ++q)
                if (((open >> q) & 1) && accs[q]) atomicAdd(my_partial + pi[q] * kT + pj[q], accs[q]);
            if (arrive_last()) {
#pragma unroll
                for (int q = 0; q < kSparse; ++q)
                    if ((open >> q) & 1) settle(pi[q], pj[q], atomicAdd(my_partial + pi[q] * kT + pj[q], 0), true);
                __syncthreads();
                flush();
            }
        }
        if (kDiag) diag_lap(diag, 7, &t_lap);                                 // pair-list pass (incl. its unions)
        if (kDiag && tid == 0) atomicAdd(diag + 9, 1);
        return;
    }
    // many candidates: full 4x4 register blocks.  For a settle phase the 64 x 64 partial counts go through LDS (the
    // staging image `sa` is free between two stages) and the candidate list is walked pair by pair, thread p
    // taking pairs p, p + 256, ...: evenly spread whatever the shape of the candidate set, and a rolled loop.
    int acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0;
    int (*cnts)[kT + 1] = reinterpret_cast<int (*)[kT + 1]>(&sa[0][0]);       // 64 x 65 int32 = 16 640 B <= sizeof(sa)
    static_assert(sizeof(int) * kT * (kT + 1) <= sizeof(sa), "partial-count image must fit the staging buffer");
    const int n_mine = min(kT * kT / 256, max(0, (n_pairs - tid + 255) / 256));
    unsigned open = (1u << n_mine) - 1;                            // bit q: pair tid + 256 q of plist is undecided
    int stages_done = 0;
    for (int g = g_first; g < cnt; g += g_step) {
        stage();
        if (tid == 0) g_block_info[1] += 0x10000;                 // dense stages count in the upper half
        __syncthreads();
        if (g + g_step < cnt) fetch(g + g_step);
#pragma unroll 8
        for (int kk = 0; kk < kKW; ++kk) {
            uint64_t av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { av[r] = sa[kk][ti * 4 + r]; bv[r] = sb[kk][tj * 4 + r]; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] += popc64(av[r] & bv[c]);
        }
        __syncthreads();
        const bool last = g + g_step >= cnt;
        if (!split && (++stages_done == next_check || last)) {
            const unsigned before_open = open;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) cnts[ti * 4 + r][tj * 4 + c] = acc[r][c];
            __syncthreads();
#pragma unroll 1
            for (int q = 0; q < kT * kT / 256; ++q) {
                const int p = tid + q * 256;
                if (p >= n_pairs) break;
                if (!((open >> q) & 1)) continue;
                const int code = plist[p];
                if (settle(code >> 8, code & 255, cnts[code >> 8][code & 255], last)) open &= ~(1u << q);
            }
            const int any_open = __syncthreads_or(open != 0);
            const int progress = __syncthreads_or(open != before_open);
            flush();
            if (!any_open) break;
            if (!progress) check_every *= 2;
            next_check = stages_done + check_every;
        }
    }
    if (split) {
        // this part's counts go through the LDS image (as in a settle phase) and are added pair by pair: a rolled loop
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) cnts[ti * 4 + r][tj * 4 + c] = acc[r][c];
        __syncthreads();
#pragma unroll 1
        for (int q = 0; q < kT * kT / 256; ++q) {
            const int p = tid + q * 256;
            if (p >= n_pairs) break;
            const int code = plist[p];
            const int v = cnts[code >> 8][code & 255];
            if (v) atomicAdd(my_partial + (code >> 8) * kT + (code & 255), v);
        }
        if (arrive_last()) {
#pragma unroll 1
            for (int q = 0; q < kT * kT / 256; ++q) {
                const int p = tid + q * 256;
                if (p >= n_pairs) break;
                const int code = plist[p];
                settle(code >> 8, code & 255, atomicAdd(my_partial + (code >> 8) * kT + (code & 255), 0), true);
            }
            __syncthreads();
            flush();
        }
    }
    if (kDiag) diag_lap(diag, 8, &t_lap);                                     // dense pass (incl. its unions)
    if (kDiag && tid == 0) atomicAdd(diag + 10, 1);
}

// Tile pass over the filtered list: block b takes entry b; blocks beyond the list (its length is only known on
// the device) leave at once.  (A grid-stride or work-queue loop around the tile pair costs 60-80 registers and a
// third of the occupancy: measured slower.)  kMode: 0 production; 1 counters + phase clocks (costs registers: one
// block less per CU); 2 block timeline only -- start / end of every block at production occupancy.
template <int kMode>
__global__ __launch_bounds__(256) void merge_components_kernel(const uint64_t *__restrict__ rows, int n, int64_t nw,
                                                                const uint64_t *__restrict__ tmask, int mw,
                                                                const uint32_t *__restrict__ hist, int n_pos,
                                                                const int32_t *__restrict__ row_sorted,
                                                                const int32_t *__restrict__ area_sorted,
                                                                const int32_t *__restrict__ label_sorted, float thr,
                                                                int32_t *__restrict__ parent, int n_tiles,
                                                                int32_t *__restrict__ diag,
                                                                const int32_t *__restrict__ list,
                                                                const uint64_t *__restrict__ pass,
                                                                const int32_t *__restrict__ count,
                                                                const uint16_t *__restrict__ cpop,
                                                                const int32_t *__restrict__ part2,
                                                                int32_t *__restrict__ partial, int32_t *__restrict__ arrive)
{
    if ((int)blockIdx.x >= *count) return;                         // block-uniform
    long long t_start = 0;
    if (kMode) t_start = (long long)__builtin_amdgcn_s_memrealtime();    // 100 MHz, one clock for the whole chip
    if (threadIdx.x == 0) { g_block_info[0] = 0; g_block_info[1] = 0; }
    merge_tile_pair<kMode == 1>(list[blockIdx.x], pass[2 * blockIdx.x], pass[2 * blockIdx.x + 1], rows, n, nw, tmask, mw, hist,
                                n_pos, row_sorted, area_sorted, label_sorted, thr, parent, n_tiles, kMode == 1 ? diag : nullptr, cpop,
                                part2 ? part2[blockIdx.x] : 0, partial, arrive);
    if (kMode && threadIdx.x == 0 && diag[15] > 0 && (int)blockIdx.x < diag[15]) {
        // block timeline (diag[15] = capacity): start / end in 10-ns ticks (low 32 bits), at diag[16 + 2 b]
        diag[16 + 2 * blockIdx.x] = (int32_t)t_start;
        diag[17 + 2 * blockIdx.x] = (int32_t)(long long)__builtin_amdgcn_s_memrealtime();
        if (kMode == 2 && diag[14] > 0) {          // diag[14] != 0: two more words per block behind the timeline
            diag[16 + 2 * diag[15] + 2 * blockIdx.x] = g_block_info[0];
            diag[17 + 2 * diag[15] + 2 * blockIdx.x] = g_block_info[1];
        }
    }
}

// out bit o of row r = in bit idx[o] of row r  (bit gather; undoes the spatial point sort)
__global__ void permute_bits_kernel(const uint64_t *__restrict__ in, int64_t nw_in, const int32_t *__restrict__ idx,
                                    int64_t n_out, int64_t nw_out, uint64_t *__restrict__ out)
{
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bit = false;
    if (o < n_out) {
        const int s = idx[o];
        bit = (in[(int64_t)blockIdx.y * nw_in + (s >> 6)] >> (s & 63)) & 1;
    }
    const uint64_t bal = __ballot(bit);
    if (lane_id() == 0 && (o >> 6) < nw_out) out[(int64_t)blockIdx.y * nw_out + (o >> 6)] = bal;
}

// ---- group OR / confidence mean ---------------------------------------------------------------
constexpr int kOrSplit = 32;      // members per block along z

// Sequential mean of one group's confidences by the first wave of the calling block: all its lanes gather
// 1024 confidences into LDS at once (the gathers are the slow part), then lane 0 runs the strictly sequential
// sum the reference defines (P:225) -- one rounding in the confidence dtype per step.
template <typename T>
__device__ __forceinline__ void group_conf_mean_wave(const T *__restrict__ conf, const int32_t *__restrict__ offs,
                                                     const int32_t *__restrict__ members, int g, T *__restrict__ mean,
                                                     T *stage /* LDS [1024] */)
{
    const int lane = threadIdx.x;                  // callers pass threads 0..63 only
    const int lo = offs[g], hi = offs[g + 1];
    T s;
    if constexpr (sizeof(T) == 2) s = __float2half_rn(0.0f); else s = 0.0f;
    for (int base = lo; base < hi; base += 1024) {
        const int cnt = min(1024, hi - base);
        for (int k = lane; k < cnt; k += kWave) stage[k] = conf[members[base + k]];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // one wave: LDS ops complete in issue order
        if (lane == 0) {
#pragma unroll 8
            for (int k = 0; k < cnt; ++k) {
                if constexpr (sizeof(T) == 2) s = __hadd(s, stage[k]);        // one f16 rounding per step
                else s = __fadd_rn(s, stage[k]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (lane == 0) {
        if constexpr (sizeof(T) == 2) mean[g] = __float2half_rn(__fdiv_rn(__half2float(s), (float)(hi - lo)));
        else mean[g] = __fdiv_rn(s, (float)(hi - lo));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void or_reduce_groups_kernel(const uint64_t *__restrict__ rows, int64_t nw,
                                                                const int32_t *__restrict__ offs,
                                                                const int32_t *__restrict__ members, int n_groups,
                                                                uint64_t *__restrict__ out, const T *__restrict__ conf,
                                                                T *__restrict__ mean, const uint64_t *__restrict__ cmask, int mw)
{
    __shared__ uint32_t s_occ[kOrSplit];           // cmask given: the members' chunk flags for this block's 256 words
    // blockIdx.z takes members [z*32, z*32+32) of group blockIdx.y; partial ORs meet in the zeroed output.
    // Blocks with blockIdx.y == 0 when conf != NULL (groups then start at y = 1) do not OR anything: their first wave
    // computes the sequential confidence means of groups blockIdx.x, blockIdx.x + gridDim.x, ... so that the
    // longest chain of dependent additions runs beside the OR instead of after it.
    __shared__ T stage[1024];
    const int g = conf ? (int)blockIdx.y - 1 : (int)blockIdx.y;     // slice y = 0 is dispatched first
    if (g < 0) {
        if (blockIdx.z == 0 && threadIdx.x < kWave)
            for (int q = blockIdx.x; q < n_groups; q += gridDim.x) group_conf_mean_wave(conf, offs, members, q, mean, stage);
        return;
    }
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lo = offs[g] + blockIdx.z * kOrSplit, hi = min(offs[g + 1], lo + kOrSplit);
    if (lo >= hi) return;                                          // block-uniform
    uint64_t v = 0;
    if (cmask) {                                                   // long rows: see or_reduce_grouped_kernel
        if (threadIdx.x < hi - lo) {
            const uint64_t m64 = cmask[(int64_t)members[lo + threadIdx.x] * mw + (blockIdx.x >> 1)];
            s_occ[threadIdx.x] = (uint32_t)(m64 >> (32 * (blockIdx.x & 1)));
        }
        __syncthreads();
        if (w >= nw) return;
        const int c = threadIdx.x >> 3;
        for (int m = lo; m < hi; ++m)
            if ((s_occ[m - lo] >> c) & 1) v |= rows[(int64_t)members[m] * nw + w];
    } else {
        if (w >= nw) return;
#pragma unroll 8
        for (int m = lo; m < hi; ++m) v |= rows[(int64_t)members[m] * nw + w];
    }
    if (gridDim.z == 1) out[(int64_t)g * nw + w] = v;
    else if (v) atomicOr((unsigned long long *)(out + (int64_t)g * nw + w), (unsigned long long)v);
}

template <typename T>
__global__ __launch_bounds__(64) void group_conf_mean_kernel(const T *__restrict__ conf,
                                                             const int32_t *__restrict__ offs,
                                                             const int32_t *__restrict__ members, int n_groups,
                                                             T *__restrict__ mean)
{
    __shared__ T stage[1024];
    group_conf_mean_wave(conf, offs, members, (int)blockIdx.x, mean, stage);
}

// ---- groups on the device -------------------------------------------------------------------------
// Component ids -> the groups merge_masks keeps (P:203-226), without a host round trip: the device twin of
// bff_host_component_csr for at most `cap` groups.  comp[i] = smallest row index of i's component, so group
// order "by smallest member" is the order of the roots, and "members ascending" is the order of the rows.
//   info[0] = K (number of kept groups, may exceed cap), info[1] = flags (1: K > cap, 2: empty components survive
//   the filter, i.e. min_members <= 0 -- both mean "take the general host path"), info[2] = largest kept group,
//   info[3] = number of 32-member slices of the kept groups (work items of bff_or_reduce_grouped).
__global__ void group_count_kernel(int32_t *__restrict__ comp, int n, int32_t *__restrict__ count,
                                   int32_t *__restrict__ parent)
{
    // 64 consecutive rows (two views' masks) belong to a handful of components: one atomic per distinct root of the
    // wave instead of one per row (thousands of rows share a few dozen counters).  parent != NULL: comp is an OUTPUT,
    // the flattened disjoint-set forest (comp[i] = root of i = smallest row of its component; uf_flatten_kernel fused).
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int root = -1;
    if (i < n) {
        if (parent) { root = uf_find(parent, i); comp[i] = root; }
        else root = comp[i];
    }
    uint64_t todo = __ballot(root >= 0);
    while (todo) {
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const int r = __shfl(root, leader);
        const uint64_t same = __ballot(root == r);
        if (lane_id() == leader) atomicAdd(count + r, __popcll(same));
        todo &= ~same;
    }
}

__global__ __launch_bounds__(1024) void group_scan_kernel(const int32_t *__restrict__ comp, const int32_t *__restrict__ count,
                                                           const int32_t *__restrict__ area, int n, float thr,
                                                           int min_members, int cap, int32_t *__restrict__ info,
                                                           int32_t *__restrict__ sizes, int32_t *__restrict__ first,
                                                           int32_t *__restrict__ offs, int32_t *__restrict__ slices,
                                                           int slice_cap)
{
    __shared__ int wsum[16];
    __shared__ int s_base, s_void, s_max;
    __shared__ int s_off[kFuseMax + 1], s_soff[kFuseMax + 1], s_sz[kFuseMax];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_base = 0; s_void = 0; s_max = 0; }
    __syncthreads();
    const int need = min_members > 1 ? min_members : 1;
    const bool loops = 1.0f > thr;                                  // a non-empty row is adjacent to itself iff 1 > thr
    constexpr int kAhead = 4;                                      // chunks whose loads are in flight together
    __shared__ int wsum2[2][16];
    int base = 0, it = 0;
    for (int d0 = 0; d0 < n; d0 += 1024 * kAhead) {
        int szv[kAhead], arv[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int c = d0 + u * 1024 + tid;
            const bool root = c < n && comp[c] == c;
            szv[u] = root ? count[c] : -1;                             // -1: not a root
            arv[u] = root ? area[c] : 0;
        }
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const int c0 = d0 + u * 1024;
        if (c0 >= n) break;                                            // block-uniform
        const int c = c0 + tid;
        bool valid = false, is_void = false;
        int sz = 0;
        if (szv[u] >= 0) {
            sz = szv[u];
            is_void = sz == 1 && !(arv[u] > 0 && loops);            // isolated row without a self loop: the reference's []
            valid = !is_void && sz >= need;
        }
        const uint64_t bal = __ballot(valid);
        if (lane == 0) wsum2[it][wave] = __popcll(bal);
        if (is_void) atomicAdd(&s_void, 1);
        __syncthreads();                                   // double-buffered counters: one barrier per chunk
        int g = base + __popcll(bal & ((1ull << lane) - 1)), total = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int cq = wsum2[it][q]; total += cq; if (q < wave) g += cq; }
        if (valid) {
            if (g < cap) { sizes[g] = sz; first[g] = c; s_sz[g] = sz; }
            atomicMax(&s_max, sz);
        }
        base += total;                                     // every thread keeps the running group count
        it ^= 1;
      }
    }
    if (tid == 0) s_base = base;
    __syncthreads();
    const int k_all = s_base, k = min(k_all, cap);
    {
        // exclusive prefix sums of the groups' sizes and 32-member slice counts over the k <= 512 groups: thread g owns
        // group g (one serial walk by one thread: up to 512 dependent LDS round trips, 20+ us for scenes with many groups)
        const int sz = tid < k ? s_sz[tid] : 0, sl = (sz + kOrSplit - 1) / kOrSplit;
        int io = sz, is = sl;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int uo = __shfl_up(io, d), us = __shfl_up(is, d);
            if (lane >= d) { io += uo; is += us; }
        }
        __syncthreads();                               // wsum was last read two barriers ago; reuse it for both sums
        __shared__ int wsum3[16];
        if (lane == 63) { wsum[wave] = io; wsum3[wave] = is; }
        __syncthreads();
        int bo = 0, bs = 0;
        for (int q = 0; q < wave; ++q) { bo += wsum[q]; bs += wsum3[q]; }
        if (tid < k) { s_off[tid] = bo + io - sz; s_soff[tid] = bs + is - sl; }
        if (tid == k - 1 || (k == 0 && tid == 0)) {
            const int o = k ? bo + io : 0, so = k ? bs + is : 0;
            s_off[k] = o; s_soff[k] = so;
            info[0] = k_all;
            info[1] = (k_all > cap ? 1 : 0) | ((min_members <= 0 && s_void > 0) ? 2 : 0);
            info[2] = s_max;
            info[3] = min(so, slice_cap);
        }
    }
    __syncthreads();
    for (int g = tid; g <= k; g += 1024) offs[g] = s_off[g];
    for (int g = tid + k + 1; g <= cap; g += 1024) offs[g] = s_off[k];
    // slice s of group g covers members [offs[g] + 32 j, min(offs[g+1], ...)): table rows (group, lo, hi)
    const int n_slices = min(s_soff[k], slice_cap);
    for (int sidx = tid; sidx < n_slices; sidx += 1024) {
        int g = 0, hi = k - 1;                       // last group whose first slice is <= sidx
        while (g < hi) { const int mid = (g + hi + 1) >> 1; if (s_soff[mid] <= sidx) g = mid; else hi = mid - 1; }
        const int lo = s_off[g] + (sidx - s_soff[g]) * kOrSplit;
        slices[sidx] = g;
        slices[slice_cap + sidx] = lo;
        slices[2 * slice_cap + sidx] = min(s_off[g + 1], lo + kOrSplit);
    }
}

// members of group g in ascending row order: one block per group walks comp[] 256 rows at a time (ballot per wave,
// the four waves' counts meet in LDS)
__global__ __launch_bounds__(256) void group_members_kernel(const int32_t *__restrict__ comp, int n,
                                                             const int32_t *__restrict__ info, int cap,
                                                             const int32_t *__restrict__ first,
                                                             const int32_t *__restrict__ offs,
                                                             int32_t *__restrict__ members)
{
    __shared__ int wcnt[2][4];
    const int g = blockIdx.x;
    if (g >= min(info[0], cap)) return;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int root = first[g];
    int base = offs[g], it = 0;
    constexpr int kAhead = 4;                                  // steps whose loads are in flight together
    for (int j0 = 0; j0 < n; j0 += 256 * kAhead) {
        int cv[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int i = j0 + u * 256 + (int)threadIdx.x;
            cv[u] = i < n ? comp[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u, it ^= 1) {
            const int i = j0 + u * 256 + (int)threadIdx.x;
            if (j0 + u * 256 >= n) break;                      // block-uniform
            const bool m = cv[u] == root;                      // roots are >= 0
            const uint64_t bal = __ballot(m);
            if (lane == 0) wcnt[it][wave] = __popcll(bal);
            __syncthreads();                                   // double-buffered counters: one barrier per step
            int before = 0, total = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int c = wcnt[it][q]; total += c; if (q < wave) before += c; }
            if (m) members[base + before + __popcll(bal & ((1ull << lane) - 1))] = i;
            base += total;
        }
    }
}

// Undo the spatial point sort by SCATTER: out[r] bit perm[s] = in[r] bit s for the set bits only (aggregated rows
// hold a few percent of the points, so this touches ~1/50 of what a bit gather per output point reads).  out must be
// zero; perm[s] = original index of sorted position s.  Rows >= *k_dev (when given) are skipped.
__global__ __launch_bounds__(256) void scatter_bits_kernel(const uint64_t *__restrict__ in, int64_t nw_in,
                                                            const int32_t *__restrict__ perm, int64_t n,
                                                            int64_t nw_out, uint64_t *__restrict__ out,
                                                            const int32_t *__restrict__ k_dev)
{
    const int r = blockIdx.y;
    if (k_dev && r >= *k_dev) return;
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw_in) return;
    uint64_t v = in[(int64_t)r * nw_in + w];
    while (v) {
        const int b = __ffsll((unsigned long long)v) - 1;
        v &= v - 1;
        const int64_t s = w * 64 + b;
        if (s < n) {
            const int o = perm[s];
            atomicOr((unsigned long long *)(out + (int64_t)r * nw_out + (o >> 6)), 1ull << (o & 63));
        }
    }
}

// bff_or_reduce_groups for groups formed on the device: the work items are the 32-member slices listed by
// group_scan_kernel (their number is only known on the device: blocks beyond it leave at once); slice y = 0 of the
// grid computes the sequential confidence means, as in or_reduce_groups_kernel.
template <typename T>
__global__ __launch_bounds__(256) void or_reduce_grouped_kernel(const uint64_t *__restrict__ rows, int64_t nw,
                                                                 const int32_t *__restrict__ info, int cap,
                                                                 const int32_t *__restrict__ offs,
                                                                 const int32_t *__restrict__ members,
                                                                 const int32_t *__restrict__ slices, int slice_cap,
                                                                 uint64_t *__restrict__ out, const T *__restrict__ conf,
                                                                 T *__restrict__ mean, const uint64_t *__restrict__ cmask, int mw)
{
    __shared__ T stage[1024];
    __shared__ uint32_t s_occ[kOrSplit];           // cmask given: the 32 chunk flags of every member for this block's 256 words
    if (blockIdx.y == 0) {
        if (conf && threadIdx.x < kWave) {
            const int k = min(info[0], cap);
            for (int q = blockIdx.x; q < k; q += gridDim.x) group_conf_mean_wave(conf, offs, members, q, mean, stage);
        }
        return;
    }
    const int sidx = (int)blockIdx.y - 1;
    if (sidx >= info[3]) return;
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw && !cmask) return;
    const int g = slices[sidx], lo = slices[slice_cap + sidx], hi = slices[2 * slice_cap + sidx];
    uint64_t v = 0;
    if (cmask) {
        // long rows (~1 % occupied): a block's 256 words are 32 chunks = one 32-bit piece of a member's chunk flags;
        // a member's word is loaded only where the member has points
        if (threadIdx.x < hi - lo) {
            const uint64_t m64 = cmask[(int64_t)members[lo + threadIdx.x] * mw + (blockIdx.x >> 1)];
            s_occ[threadIdx.x] = (uint32_t)(m64 >> (32 * (blockIdx.x & 1)));
        }
        __syncthreads();
        const int c = threadIdx.x >> 3;                            // chunk of this thread's word within the block
        if (w < nw)
            for (int m = lo; m < hi; ++m)
                if ((s_occ[m - lo] >> c) & 1) v |= rows[(int64_t)members[m] * nw + w];
    } else {
#pragma unroll 8
        for (int m = lo; m < hi; ++m) v |= rows[(int64_t)members[m] * nw + w];
    }
    if (v) atomicOr((unsigned long long *)(out + (int64_t)g * nw + w), (unsigned long long)v);
}

// ---- row programs -----------------------------------------------------------------------------
// Sequential overlap decisions of solve_overlapping (P:285-299) on the device: inter is the K x K
// intersection matrix of the aggregated rows BEFORE any edit (P:289-292), size[i] the number of raw masks
// merged into row i; pairs are visited in the reference's order (i ascending, j > i ascending) and the
// and-not operations appended to `ops` ([0] = count, then (opcode, dst, src) triples).
constexpr int kOvlRows = 8192;   // rows whose pair counts fit the block's LDS; beyond that one thread walks the pairs

// One block: (1) wave w counts, for its rows i = w, w + 16, ..., the rows j > i with inter[i][j] > 0 (ballots over 64
// columns at a time), (2) a block-wide exclusive scan turns the counts into list offsets -- the reference visits the
// pairs in (i ascending, j ascending) order and that IS the order of (offset of i, rank of j within i), (3) the waves
// walk their rows again and write the triples.  The order of the list is the semantics (P:285-299); building it is
// embarrassingly parallel.
__global__ __launch_bounds__(1024) void overlap_ops_kernel(const int32_t *__restrict__ inter,
                                                            const int32_t *__restrict__ size, int k,
                                                            int32_t *__restrict__ ops)
{
    __shared__ int s_cnt[kOvlRows];
    __shared__ int s_wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (k > kOvlRows) {                            // not a realistic size: the plain ordered loop
        if (tid) return;
        int n = 0;
        for (int i = 0; i < k; ++i)
            for (int j = i + 1; j < k; ++j)
                if (inter[(int64_t)i * k + j] > 0) {
                    const bool i_wins = size[i] > size[j];            // ties: i loses (P:296-299)
                    ops[1 + 3 * n] = 0; ops[2 + 3 * n] = i_wins ? j : i; ops[3 + 3 * n] = i_wins ? i : j;
                    ++n;
                }
        ops[0] = n;
        return;
    }
    for (int i = wave; i < k; i += 16) {
        int c = 0;
        for (int j0 = (i + 1) & ~63; j0 < k; j0 += 64) {
            const int j = j0 + lane;
            c += __popcll(__ballot(j > i && j < k && inter[(int64_t)i * k + j] > 0));
        }
        if (lane == 0) s_cnt[i] = c;
    }
    __syncthreads();
    // exclusive scan of s_cnt[0..k) in place: thread t owns a contiguous run of ceil(k / 1024) rows
    const int per = (k + 1023) / 1024, lo = tid * per, hi = min(k, lo + per);
    int mine = 0;
    for (int i = lo; i < hi; ++i) mine += s_cnt[i];
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int q = 0; q < wave; ++q) base += s_wsum[q];
    if (tid == 1023) ops[0] = base + mine;
    for (int i = lo; i < hi; ++i) { const int c = s_cnt[i]; s_cnt[i] = base; base += c; }
    __syncthreads();
    for (int i = wave; i < k; i += 16) {
        int at = s_cnt[i];
        const int size_i = size[i];
        for (int j0 = (i + 1) & ~63; j0 < k; j0 += 64) {
            const int j = j0 + lane;
            const bool on = j > i && j < k && inter[(int64_t)i * k + j] > 0;
            const uint64_t bal = __ballot(on);
            if (on) {
                const int n = at + __popcll(bal & ((1ull << lane) - 1));
                const bool i_wins = size_i > size[j];                 // ties: i loses (P:296-299)
                ops[1 + 3 * n] = 0;
                ops[2 + 3 * n] = i_wins ? j : i;
                ops[3 + 3 * n] = i_wins ? i : j;
            }
            at += __popcll(bal);
        }
    }
}

__global__ void apply_row_ops_kernel(uint64_t *__restrict__ rows, int64_t nw, const int32_t *__restrict__ ops,
                                     int n_ops)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    if (n_ops < 0) { n_ops = ops[0]; ops += 1; }          // device-side list: [count, triples...]
    for (int k = 0; k < n_ops; ++k) {
        const int op = ops[3 * k], d = ops[3 * k + 1], s = ops[3 * k + 2];
        const uint64_t sv = rows[(int64_t)s * nw + w];
        uint64_t *dp = rows + (int64_t)d * nw + w;
        *dp = op == 0 ? (*dp & ~sv) : op == 1 ? (*dp | sv) : sv;
    }
}

// solve_overlapping (P:277-301) + the point filter (P:595) + both popcounts (P:592, 596) in ONE pass, for any number
// of rows.
//
// The reference lists the pairs (i < j) that share a point BEFORE any edit and visits them in (i, j) order: the row
// merged from more raw masks keeps the current overlap, the other loses it, ties go to j (P:285-299).  Seen from ONE
// point p this is a walk over S = the rows that hold p at the start (every pair inside S shares p, so every one of
// them is on the list; rows outside S neither change at p nor change others there).  A row's bit is only ever
// cleared, and a pair with a cleared bit changes nothing, so the walk is a champion scan over S in index order: the
// first row stays until it meets a row of at least its size, which then takes its place, and so on.  Champion sizes
// never decrease and a later equal size replaces the champion, hence
//     p ends up in exactly one row of S: the one with the largest size, and among those the LARGEST index.
// With the rows ordered by that priority (size descending, index descending) the whole loop is one exclusive prefix
// OR: row r keeps  r & ~(OR of the rows ranked before it).  No pair list, no intersections, no order dependence
// between words.  (tests: against the literal ordered replay, bff_overlap_ops + bff_apply_row_ops, and the oracle.)
//
// One block = 64 word columns x 16 waves; wave s owns the ranks [s L, (s+1) L), L = ceil(k / 16): it loads its rows'
// words (independent loads, all in flight), ORs them, the 16 segment sums meet in LDS, and every row is finished with
// the OR of the segments before its own plus its own exclusive prefix.  Rows that do not change are not written.
constexpr int kResWaves = 16;
constexpr int kResolveMax = 4096;          // rows: the ranks are found by counting, k^2 / 1024 comparisons per thread

__device__ __forceinline__ uint32_t wave_sum_to_lane63(uint32_t v)
{
#define BFF_DPP_ADD(ctrl, rows) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xF, false)
    BFF_DPP_ADD(0x111, 0xF);    // row_shr:1
    BFF_DPP_ADD(0x112, 0xF);    // row_shr:2
    BFF_DPP_ADD(0x114, 0xF);    // row_shr:4
    BFF_DPP_ADD(0x118, 0xF);    // row_shr:8
    BFF_DPP_ADD(0x142, 0xA);    // row_bcast:15 -> rows 1 and 3
    BFF_DPP_ADD(0x143, 0xC);    // row_bcast:31 -> rows 2 and 3
#undef BFF_DPP_ADD
    return v;                   // lane 63 holds the sum of all 64 lanes
}

// one finished row: write it if it changed, add its popcounts before / after (<= 4096 each per wave: two 16-bit fields)
__device__ __forceinline__ void resolve_emit(uint64_t *__restrict__ dst, uint64_t v, uint64_t out, bool in, int row,
                                             int32_t *__restrict__ before, int32_t *__restrict__ after, int lane)
{
    if (in && out != v) *dst = out;
    const uint32_t pc = wave_sum_to_lane63(((uint32_t)popc64(v) << 16) | (uint32_t)popc64(out));
    if (lane == kWave - 1) {
        if (pc >> 16) atomicAdd(before + row, (int)(pc >> 16));
        if (pc & 0xffffu) atomicAdd(after + row, (int)(pc & 0xffffu));
    }
}

template <int kMaxL>
__device__ __forceinline__ void resolve_segment_in_registers(uint64_t *__restrict__ rows, int64_t nw, int64_t w, bool in,
                                                             int r0, int r1, const int *s_order, uint64_t (*s_seg)[kWave],
                                                             uint64_t kp, int32_t *__restrict__ before,
                                                             int32_t *__restrict__ after, int lane, int wave)
{
    uint64_t v[kMaxL];
#pragma unroll
    for (int q = 0; q < kMaxL; ++q)                                  // r0 + q < r1 is wave-uniform
        v[q] = (r0 + q < r1 && in) ? rows[(int64_t)s_order[r0 + q] * nw + w] : 0;
    uint64_t tot = 0;
#pragma unroll
    for (int q = 0; q < kMaxL; ++q) tot |= v[q];
    s_seg[wave][lane] = tot;
    __syncthreads();
    uint64_t claimed = 0;
    for (int s = 0; s < wave; ++s) claimed |= s_seg[s][lane];
#pragma unroll
    for (int q = 0; q < kMaxL; ++q)
        if (r0 + q < r1) {
            const int row = s_order[r0 + q];
            resolve_emit(rows + (int64_t)row * nw + w, v[q], v[q] & ~claimed & kp, in, row, before, after, lane);
            claimed |= v[q];
        }
}

__global__ __launch_bounds__(1024) void resolve_priority_kernel(uint64_t *__restrict__ rows, int64_t nw, int k,
                                                                 const int32_t *__restrict__ size,
                                                                 const uint64_t *__restrict__ keep,
                                                                 int32_t *__restrict__ before, int32_t *__restrict__ after,
                                                                 const int32_t *__restrict__ k_dev)
{
    // k_dev != NULL: the row count lives on the device (groups formed there); k is then the capacity the launch
    // was sized for and a count beyond it leaves the rows alone (the host sees the count and takes the general path)
    if (k_dev) {
        const int kd = *k_dev;
        if (kd <= 0 || kd > k) return;
        k = kd;
    }
    extern __shared__ int s_res[];                                  // [k] sizes, then [k] rows by priority
    int *s_size = s_res, *s_order = s_res + k;
    __shared__ uint64_t s_seg[kResWaves][kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    for (int r = tid; r < k; r += 1024) s_size[r] = size[r];
    __syncthreads();
    for (int r = tid; r < k; r += 1024) {
        const int sr = s_size[r];
        int rank = 0;                                                // rows that take their points before row r does
#pragma unroll 8
        for (int q = 0; q < k; ++q) {
            const int sq = s_size[q];
            rank += (sq > sr || (sq == sr && q > r)) ? 1 : 0;
        }
        s_order[rank] = r;
    }
    __syncthreads();
    const int64_t w = (int64_t)blockIdx.x * kWave + lane;
    const bool in = w < nw;
    const uint64_t kp = keep ? (in ? keep[w] : 0) : ~0ull;
    const int len = (k + kResWaves - 1) / kResWaves;                 // block-uniform
    const int r0 = min(k, wave * len), r1 = min(k, r0 + len);
    if (len <= 2) {
        resolve_segment_in_registers<2>(rows, nw, w, in, r0, r1, s_order, s_seg, kp, before, after, lane, wave);
    } else if (len <= 8) {
        resolve_segment_in_registers<8>(rows, nw, w, in, r0, r1, s_order, s_seg, kp, before, after, lane, wave);
    } else if (len <= 32) {
        resolve_segment_in_registers<32>(rows, nw, w, in, r0, r1, s_order, s_seg, kp, before, after, lane, wave);
    } else {
        // more than 512 rows: two passes over the segment (the second one finds its words in the cache)
        uint64_t tot = 0;
#pragma unroll 8
        for (int r = r0; r < r1; ++r) tot |= in ? rows[(int64_t)s_order[r] * nw + w] : 0;
        s_seg[wave][lane] = tot;
        __syncthreads();
        uint64_t claimed = 0;
        for (int s = 0; s < wave; ++s) claimed |= s_seg[s][lane];
        for (int r = r0; r < r1; ++r) {
            const int row = s_order[r];
            uint64_t *dst = rows + (int64_t)row * nw + w;
            const uint64_t v = in ? *dst : 0;
            resolve_emit(dst, v, v & ~claimed & kp, in, row, before, after, lane);
            claimed |= v;
        }
    }
}

__global__ void and_rows_kernel(uint64_t *__restrict__ rows, int64_t nw, const uint64_t *__restrict__ keep)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nw) rows[(int64_t)blockIdx.y * nw + w] &= keep[w];
}

__global__ void gather_rows_kernel(const uint64_t *__restrict__ rows, const int32_t *__restrict__ idx, int64_t nw,
                                   uint64_t *__restrict__ out)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nw) out[(int64_t)blockIdx.y * nw + w] = rows[(int64_t)idx[blockIdx.y] * nw + w];
}

// ---- dense <-> bits ---------------------------------------------------------------------------
__global__ void unpack_rows_kernel(const uint64_t *__restrict__ rows, int64_t nw, int64_t n, uint8_t *__restrict__ dense)
{
    // one thread expands 8 points (one byte of the bit row) into 8 bytes
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // group of 8 points
    const int64_t p0 = g * 8;
    if (p0 >= n) return;
    const uint32_t byte = (uint32_t)(rows[(int64_t)blockIdx.y * nw + (p0 >> 6)] >> (p0 & 63)) & 0xFFu;
    // 4 bits -> 4 bytes: the partial products land on disjoint bits, so there are no carries
    const uint32_t lo = ((byte & 0xF) * 0x00204081u) & 0x01010101u;
    const uint32_t hi = ((byte >> 4) * 0x00204081u) & 0x01010101u;
    uint8_t *out = dense + (int64_t)blockIdx.y * n + p0;
    if (p0 + 8 <= n && (((uintptr_t)out) & 7) == 0) {
        *reinterpret_cast<uint64_t *>(out) = (uint64_t)lo | ((uint64_t)hi << 32);
    } else {
        for (int k = 0; k < 8 && p0 + k < n; ++k) out[k] = (byte >> k) & 1;
    }
}

__global__ void pack_rows_kernel(const uint8_t *__restrict__ dense, int64_t n, int64_t nw, uint64_t *__restrict__ rows)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool bit = p < n && dense[(int64_t)blockIdx.y * n + p] != 0;
    const uint64_t bal = __ballot(bit);
    if (lane_id() == 0 && (p >> 6) < nw) rows[(int64_t)blockIdx.y * nw + (p >> 6)] = bal;
}

// ---- per-point ids -> bit rows (evaluation consumer, scannetv2_inst_eval.py:334: `gts == instance_id`) ---
__global__ void ids_to_rows_kernel(const int64_t *__restrict__ ids, int64_t n, const int64_t *__restrict__ values,
                                   int64_t nw, uint64_t *__restrict__ rows)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool bit = p < n && ids[p] == values[blockIdx.y];
    const uint64_t bal = __ballot(bit);
    if (lane_id() == 0 && (p >> 6) < nw) rows[(int64_t)blockIdx.y * nw + (p >> 6)] = bal;
}

// ---- 1-D RLE -> bit rows ----------------------------------------------------------------------
__global__ void rle_to_rows_kernel(const int32_t *__restrict__ run_start, const int32_t *__restrict__ run_end,
                                   const int32_t *__restrict__ offs, int64_t n, int64_t nw, uint64_t *__restrict__ rows)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    const int g = blockIdx.y;
    const int64_t p0 = w * 64, p1 = p0 + 64;
    int lo = offs[g], hi = offs[g + 1], r = hi;
    while (lo < r) {                                  // first run with end > p0
        const int mid = (lo + r) >> 1;
        if ((int64_t)run_end[mid] > p0) r = mid; else lo = mid + 1;
    }
    uint64_t v = 0;
    for (; r < hi; ++r) {
        const int64_t s = run_start[r], e = run_end[r];
        if (s >= p1) break;
        const int a = (int)(max(s, p0) - p0), b = (int)(min(e, p1) - p0);    // [a, b) within the word, b > a
        const uint64_t upto_b = b >= 64 ? ~0ull : ((1ull << b) - 1);
        v |= upto_b & ~((1ull << a) - 1);
    }
    if (p1 > n) v &= (n - p0 >= 64) ? ~0ull : ((1ull << (n - p0)) - 1);
    rows[(int64_t)g * nw + w] = v;
}

// ---- bit rows -> 1-D RLE (rle_encode_batch, rle_encode_decode.py:10-32) ------------------------------
// A run starts at point p iff bit p is set and bit p-1 is not; it ends (exclusive) at e iff bit e-1 is set
// and bit e is not.  Padding bits are zero and one virtual zero word follows the row, so a run reaching the
// last point ends at N like any other.  Starts and ends alternate: the k-th end closes the k-th start.
// Pass 1 counts the starts per row; pass 2 writes counts[2k] = start+1 (1-based) and counts[2k+1] = end,
// rank by rank (block scan of the per-word counts); pass 3 turns the ends into lengths.
__device__ __forceinline__ void rle_word_edges(const uint64_t *row, int64_t w, int64_t nw, uint64_t &starts,
                                               uint64_t &ends)
{
    const uint64_t cur = w < nw ? row[w] : 0;
    const uint64_t prev_bit = w ? (row[w - 1] >> 63) : 0;
    const uint64_t shifted = (cur << 1) | prev_bit;             // bit p = value of point p-1
    starts = cur & ~shifted;
    ends = ~cur & shifted;
}

__global__ __launch_bounds__(256) void rle_count_kernel(const uint64_t *__restrict__ rows, int64_t nw,
                                                         int32_t *__restrict__ n_runs)
{
    __shared__ int part[4];
    const uint64_t *row = rows + (int64_t)blockIdx.x * nw;
    int c = 0;
    for (int64_t w = threadIdx.x; w < nw; w += 256) {
        uint64_t st, en;
        rle_word_edges(row, w, nw, st, en);
        c += popc64(st);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
    if (lane_id() == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) n_runs[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void rle_write_kernel(const uint64_t *__restrict__ rows, int64_t nw,
                                                         const int64_t *__restrict__ run_offs,
                                                         int64_t *__restrict__ counts)
{
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t *row = rows + (int64_t)blockIdx.x * nw;
    int64_t *out = counts + 2 * run_offs[blockIdx.x];
    int base_st = 0, base_en = 0;
    for (int64_t w0 = 0; w0 <= nw; w0 += 256) {                  // <= : includes the virtual word nw
        const int64_t w = w0 + tid;
        uint64_t st = 0, en = 0;
        if (w <= nw) rle_word_edges(row, w, nw, st, en);
        const int packed = popc64(st) | (popc64(en) << 16);      // <= 32 starts / ends per word, 256 words: no carry
        int incl = packed;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int excl = incl - packed;
        for (int q = 0; q < wave; ++q) excl += wsum[q];
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        int64_t rs = base_st + (excl & 0xFFFF), re = base_en + (excl >> 16);
        while (st) { const int b = __ffsll((unsigned long long)st) - 1; st &= st - 1; out[2 * rs++] = w * 64 + b + 1; }
        while (en) { const int b = __ffsll((unsigned long long)en) - 1; en &= en - 1; out[2 * re++ + 1] = w * 64 + b; }
        base_st += total & 0xFFFF;
        base_en += total >> 16;
        __syncthreads();
    }
}

__global__ void rle_lengths_kernel(int64_t *__restrict__ counts, int64_t n_runs_total)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_runs_total) counts[2 * k + 1] -= counts[2 * k] - 1;       // end - start(0-based)
}

}  // namespace bff

using namespace bff;

// dynamic LDS beyond 64 KB has to be enabled per kernel once
extern "C" int bff_popcount_rows(const uint64_t *rows, const int32_t *idx, int32_t n_rows, int64_t nw,
                                 int32_t *area, void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && nw >= 0, "bff_popcount_rows: bad sizes");
    if (n_rows == 0) return BFF_OK;
    BFF_REQUIRE(rows && area, "bff_popcount_rows: null pointer");
    popcount_rows_kernel<<<n_rows, 256, 0, as_stream(stream)>>>(rows, idx, nw, area);
    return launched("bff_popcount_rows");
}

extern "C" int bff_cross_popcount(const uint64_t *a, const int32_t *ia, int32_t na, const uint64_t *b,
                                  const int32_t *ib, int32_t nb, int64_t nw, int32_t *inter, void *stream)
{
    BFF_REQUIRE(na >= 0 && nb >= 0 && nw >= 0, "bff_cross_popcount: bad sizes");
    if (na == 0 || nb == 0) return BFF_OK;
    BFF_REQUIRE(a && b && inter, "bff_cross_popcount: null pointer");
    const int64_t tiles = ceil_div(nb, kT) * ceil_div(na, kT);
    int64_t k_split = nw;                               // words per block along z
    if (tiles < 512) {                                  // few tiles: split the words so >= ~512 blocks run
        k_split = ceil_div(ceil_div(nw * tiles, 512), kKW) * kKW;
        if (k_split < 2 * kKW) k_split = 2 * kKW;
    }
    const int64_t nz = ceil_div(nw, k_split);
    if (nz > 1) {
        hipError_t e = hipMemsetAsync(inter, 0, sizeof(int32_t) * (size_t)na * nb, as_stream(stream));
        if (e != hipSuccess) return fail((int)e, "bff_cross_popcount: memset: %s", hipGetErrorString(e));
    }
    dim3 grid((unsigned)ceil_div(nb, kT), (unsigned)ceil_div(na, kT), (unsigned)nz);
    cross_popcount_kernel<<<grid, 256, 0, as_stream(stream)>>>(a, ia, na, b, ib, nb, nw, k_split, inter, nullptr, 0, 0);
    return launched("bff_cross_popcount");
}

// bff_cross_popcount where only the first *k_dev rows of the leading `lead` rows of b (and, with limit_a != 0, of a)
// are non-zero: tiles inside the zero part are skipped.  inter is zeroed first.
extern "C" int bff_cross_popcount_dev(const uint64_t *a, int32_t na, const uint64_t *b, int32_t nb, int64_t nw,
                                      int32_t *inter, const int32_t *k_dev, int32_t limit_a, int32_t lead, void *stream)
{
    BFF_REQUIRE(na >= 0 && nb >= 0 && nw >= 0 && lead >= 0 && lead <= nb, "bff_cross_popcount_dev: bad sizes");
    if (na == 0 || nb == 0) return BFF_OK;
    BFF_REQUIRE(a && b && inter && k_dev, "bff_cross_popcount_dev: null pointer");
    const int64_t tiles = ceil_div(nb, kT) * ceil_div(na, kT);
    int64_t k_split = nw;
    // few tiles (the refinement's S1 x (K + S1) product: a dozen, most of their rows zero): split the words until ~2048
    // blocks run, down to one LDS stage each -- the partial counts meet through atomics on non-zero entries only
    // (config 2, K = 20 / 100: 40 / 44 us with 512 blocks of >= 2 stages, 26 / 35 us with 2048 of >= 1)
    constexpr int64_t target = 2048;
    if (tiles < target) {
        k_split = ceil_div(ceil_div(nw * tiles, target), kKW) * kKW;
        if (k_split < kKW) k_split = kKW;
    }
    const int64_t nz = ceil_div(nw, k_split);
    hipError_t e = zero_async(inter, sizeof(int32_t) * (size_t)na * nb, as_stream(stream));
    if (e != hipSuccess) return fail((int)e, "bff_cross_popcount_dev: memset: %s", hipGetErrorString(e));
    dim3 grid((unsigned)ceil_div(nb, kT), (unsigned)ceil_div(na, kT), (unsigned)(nz > 0 ? nz : 1));
    cross_popcount_kernel<<<grid, 256, 0, as_stream(stream)>>>(a, nullptr, na, b, nullptr, nb, nw, k_split, inter, k_dev,
                                                              limit_a, lead);
    return launched("bff_cross_popcount_dev");
}

extern "C" int bff_row_stats(const uint64_t *rows, int32_t n_rows, int64_t nw, int32_t *area, int32_t *mean_word,
                             uint64_t *chunk_mask, int32_t chunk_mask_given, uint32_t *hist, int64_t *signature,
                             uint16_t *chunk_pop, void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && nw >= 0, "bff_row_stats: bad sizes");
    if (n_rows == 0) return BFF_OK;
    BFF_REQUIRE(rows && area && mean_word && chunk_mask && hist && signature, "bff_row_stats: null pointer");
    const int n_chunks = (int)ceil_div(nw, kCW);
    BFF_LIMIT(n_chunks <= kMaxChunks, "bff_row_stats: more than %d chunks (N > %d points)", kMaxChunks, kMaxChunks * kCW * 64);
    const int mw = (int)ceil_div(n_chunks, 64);
    static_assert(kBins == kWave, "row_stats_sparse_kernel: one histogram bin per lane");
    if (chunk_mask_given) {
        if (chunk_pop) {                            // the sparse pass writes the flagged chunks only
            hipError_t e = zero_async(chunk_pop, sizeof(uint16_t) * (size_t)n_rows * mw * 64, as_stream(stream));
            if (e != hipSuccess) return fail((int)e, "bff_row_stats: memset: %s", hipGetErrorString(e));
        }
        row_stats_sparse_kernel<<<(unsigned)ceil_div(n_rows, 4), 256, 0, as_stream(stream)>>>(
            rows, n_rows, nw, mw, (int)ceil_div(nw > 0 ? nw : 1, kBins), area, mean_word, chunk_mask, hist, signature, chunk_pop);
        return launched("bff_row_stats");
    }
    row_stats_kernel<<<n_rows, 256, mw * sizeof(uint64_t), as_stream(stream)>>>(
        rows, nw, mw, (int)ceil_div(nw > 0 ? nw : 1, kBins), area, mean_word, chunk_mask, hist, signature, chunk_pop);
    return launched("bff_row_stats");
}

extern "C" int bff_chunk_mask_words(int64_t nw) { return (int)ceil_div(ceil_div(nw, kCW), 64); }

extern "C" int bff_clear_flagged_chunks(uint64_t *rows, int32_t n_rows, int64_t nw, const uint64_t *chunk_mask,
                                        void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && nw >= 0, "bff_clear_flagged_chunks: bad sizes");
    if (n_rows == 0 || nw == 0) return BFF_OK;
    BFF_REQUIRE(rows && chunk_mask, "bff_clear_flagged_chunks: null pointer");
    clear_flagged_chunks_kernel<<<(unsigned)ceil_div(n_rows, 4), 256, 0, as_stream(stream)>>>(
        rows, n_rows, nw, chunk_mask, (int)ceil_div(ceil_div(nw, kCW), 64), nullptr);
    return launched("bff_clear_flagged_chunks");
}

extern "C" int bff_clear_flagged_chunks_unless(uint64_t *rows, int32_t n_rows, int64_t nw, const uint64_t *chunk_mask,
                                               const int32_t *veto, void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && nw >= 0, "bff_clear_flagged_chunks_unless: bad sizes");
    if (n_rows == 0 || nw == 0) return BFF_OK;
    BFF_REQUIRE(rows && chunk_mask && veto, "bff_clear_flagged_chunks_unless: null pointer");
    clear_flagged_chunks_kernel<<<(unsigned)ceil_div(n_rows, 4), 256, 0, as_stream(stream)>>>(
        rows, n_rows, nw, chunk_mask, (int)ceil_div(ceil_div(nw, kCW), 64), veto);
    return launched("bff_clear_flagged_chunks_unless");
}

extern "C" int bff_merge_adjacency(const uint64_t *rows, int32_t n_rows, int64_t nw, const int32_t *order,
                                   const uint64_t *chunk_mask, uint64_t *tile_mask, const uint32_t *hist,
                                   const int32_t *area,
                                   const int32_t *label_id, float iou_thres, uint64_t *adj, int32_t *inter,
                                   void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && nw >= 0, "bff_merge_adjacency: bad sizes");
    if (n_rows == 0) return BFF_OK;
    BFF_REQUIRE(rows && area && label_id && adj, "bff_merge_adjacency: null pointer");
    BFF_REQUIRE((chunk_mask == nullptr) == (tile_mask == nullptr), "bff_merge_adjacency: chunk_mask and tile_mask go together");
    const int nt = (int)ceil_div(n_rows, kT);
    BFF_LIMIT((int64_t)nt * (nt + 1) / 2 < (1ll << 31), "bff_merge_adjacency: too many rows");
    const int n_chunks = (int)ceil_div(nw, kCW);
    BFF_LIMIT(n_chunks <= kMaxChunks, "bff_merge_adjacency: more than %d chunks (N > %d points)", kMaxChunks, kMaxChunks * kCW * 64);
    const int mw = (int)ceil_div(n_chunks, 64);
    // pairs with an empty intersection have IoU 0 (or NaN): they can only be skipped when 0 > thr is false
    const bool sparse = chunk_mask && !(0.0f > iou_thres);
    if (sparse) tile_masks_kernel<<<nt, 256, 0, as_stream(stream)>>>(chunk_mask, order, n_rows, mw, tile_mask, nullptr, nullptr, 0, nullptr, nullptr, nullptr,
                                                                     nullptr, nullptr, nullptr, nullptr, nullptr);
    const int aw = nt;   // ceil(n_rows/64) words per adjacency row
    merge_adjacency_kernel<<<(unsigned)((int64_t)nt * (nt + 1) / 2), 256, 0, as_stream(stream)>>>(
        rows, n_rows, nw, order, sparse ? tile_mask : nullptr, mw, (sparse && !inter) ? hist : nullptr, area, label_id,
        iou_thres, adj, aw, inter, nt);
    return launched("bff_merge_adjacency");
}

// The second-level (chunk) bound pays when a histogram bin is much coarser than a chunk -- clouds of ~0.5 M points and
// more (config 4: tile pass 8.4 -> 2.7 ms); on smaller clouds the pairs that survive the 64-bin bound are genuine
// near-misses that the chunk bound cannot reject either, and its table look-ups cost more than they save (config 2:
// 0.45 -> 0.52 ms).  BFF_CHUNK_BOUND=0/1 forces it off / on.
extern "C" int32_t bff_merge_uses_chunk_bound(int64_t nw)
{
    static const int forced = [] { const char *e = getenv("BFF_CHUNK_BOUND"); return e ? atoi(e) : -1; }();
    if (forced >= 0) return forced != 0;
    // from 4 chunks per histogram bin (2048 words = 131 k points).  Round 2 had it from 16 (config 4 only): at config 2 its
    // table cost what it saved then; with the shorter chain and four scenes in flight it is 0.23 vs 0.295 ms for the tile
    // pass alone and 1351-1363 vs 1205-1286 scenes/s
    return ceil_div(nw > 0 ? nw : 1, kBins) >= 4 * kCW;
}

// entries of the second tile-pair list: every surviving pair once + the extra parts of the pairs that are split
static int64_t merge_list_cap(int64_t total)
{
    return total + (int64_t)(kMaxParts - 1) * (total < kMaxSlots ? total : kMaxSlots);
}

extern "C" int64_t bff_merge_scratch_words(int32_t n_rows)
{
    const int64_t nt = ceil_div(n_rows > 0 ? n_rows : 1, kT), n_pos = nt * kT, total = nt * (nt + 1) / 2;
    const int64_t cap2 = merge_list_cap(total);
    // sorted histogram, tile maxima, tile minima, three position-indexed row tables, counters (+ padding), list 1,
    // list 2 with two 64-bit pass masks and a part word per entry, arrival counters and 64 x 64 partial counts per slot
    return (kBins / 2) * n_pos + kBins * nt + nt + 3 * n_pos + 4 + total + cap2 + 4 * cap2 + 2 + cap2 + 4 +
           (int64_t)kMaxSlots * (kT * kT + 1);
}

namespace bff {
// bff_merge_components with the tile pass on a stream of its own (`heavy`; bff_scene_project keeps the chip-filling
// kernels of the scenes in flight on shared heavy streams so that they do not run four at a time): the pre-pass and the
// two tile-pair filters run on `stream`, `before_heavy` is recorded there and awaited by `heavy`, the tile pass runs on
// `heavy`, `after_heavy` is recorded there and awaited by `stream`.  heavy == stream (events unused): one stream.
int merge_components_streams(const uint64_t *rows, int32_t n_rows, int64_t nw, const int32_t *order,
                             int32_t n_order, const uint64_t *chunk_mask, uint64_t *tile_mask,
                             const uint32_t *hist, uint32_t *scratch, const int32_t *area,
                             const int32_t *label_id, float iou_thres, int32_t *parent, int32_t init_parent,
                             int32_t *comp, int32_t *diag, const uint16_t *chunk_pop, void *stream, void *heavy_stream,
                             void *before_heavy, void *after_heavy);
}

extern "C" int bff_merge_components(const uint64_t *rows, int32_t n_rows, int64_t nw, const int32_t *order,
                                    int32_t n_order, const uint64_t *chunk_mask, uint64_t *tile_mask,
                                    const uint32_t *hist, uint32_t *scratch, const int32_t *area,
                                    const int32_t *label_id, float iou_thres, int32_t *parent, int32_t init_parent,
                                    int32_t *comp, int32_t *diag, const uint16_t *chunk_pop, void *stream)
{
    return merge_components_streams(rows, n_rows, nw, order, n_order, chunk_mask, tile_mask, hist, scratch, area, label_id,
                                    iou_thres, parent, init_parent, comp, diag, chunk_pop, stream, stream, nullptr, nullptr);
}

int bff::merge_components_streams(const uint64_t *rows, int32_t n_rows, int64_t nw, const int32_t *order,
                                  int32_t n_order, const uint64_t *chunk_mask, uint64_t *tile_mask,
                                  const uint32_t *hist, uint32_t *scratch, const int32_t *area,
                                  const int32_t *label_id, float iou_thres, int32_t *parent, int32_t init_parent,
                                  int32_t *comp, int32_t *diag, const uint16_t *chunk_pop, void *stream, void *heavy_stream,
                                  void *before_heavy, void *after_heavy)
{
    BFF_REQUIRE(n_rows >= 0 && nw >= 0 && n_order >= 0 && n_order <= n_rows, "bff_merge_components: bad sizes");
    if (n_rows == 0) return BFF_OK;
    BFF_REQUIRE(rows && chunk_mask && tile_mask && hist && scratch && area && label_id && parent &&
                (order || n_order == n_rows), "bff_merge_components: null pointer");
    const int nt = (int)ceil_div(n_order, kT);
    BFF_LIMIT((int64_t)nt * (nt + 1) / 2 < (1ll << 31), "bff_merge_components: too many rows");
    const int n_chunks = (int)ceil_div(nw, kCW);
    BFF_LIMIT(n_chunks <= kMaxChunks, "bff_merge_components: more than %d chunks (N > %d points)", kMaxChunks, kMaxChunks * kCW * 64);
    const int mw = (int)ceil_div(n_chunks, 64);
    hipStream_t st = as_stream(stream);
    if (chunk_pop && !bff_merge_uses_chunk_bound(nw)) chunk_pop = nullptr;
    // every row appears once in `order` when it lists all of them: the tile pre-pass initialises the forest on the way
    const bool init_in_tiles = init_parent && n_order == n_rows && n_order > 0;
    if (init_parent && !init_in_tiles) uf_init_kernel<<<(unsigned)ceil_div(n_rows, 256), 256, 0, st>>>(parent, n_rows);
    if (n_order > 0) {
        // an empty intersection gives IoU 0 (or NaN): such pairs can only be skipped when 0 > thr is false
        const bool sparse = !(0.0f > iou_thres);
        const int64_t n_pos = (int64_t)nt * kT, total = (int64_t)nt * (nt + 1) / 2;
        // scratch layout (int32 words), see bff_merge_scratch_words
        uint32_t *hist_sorted = scratch;
        uint32_t *tile_hmax = hist_sorted + (size_t)(kBins / 2) * n_pos;
        int32_t *tile_amin = reinterpret_cast<int32_t *>(tile_hmax + (size_t)nt * kBins);
        int32_t *row_sorted = tile_amin + nt;
        int32_t *area_sorted = row_sorted + n_pos;
        int32_t *label_sorted = area_sorted + n_pos;
        int32_t *counts = label_sorted + n_pos;                        // [0] list 1, [1] list 2
        int32_t *list1 = counts + 4;
        const int64_t cap2 = merge_list_cap(total);
        BFF_LIMIT(cap2 < (1ll << 31), "bff_merge_components: too many rows");
        int32_t *list2 = list1 + total;
        uintptr_t p2 = reinterpret_cast<uintptr_t>(list2 + cap2);
        uint64_t *pass2 = reinterpret_cast<uint64_t *>((p2 + 7) & ~(uintptr_t)7);
        int32_t *part2 = reinterpret_cast<int32_t *>(pass2 + 2 * cap2);
        int32_t *arrive = part2 + cap2;                                // [kMaxSlots], then the slots' partial counts
        int32_t *partial = arrive + kMaxSlots;
        // heavy tile pairs are split over several blocks unless switched off (BFF_MERGE_SPLIT=0); thr < 0 visits every
        // chunk of every pair anyway and keeps the simple form
        static const bool split_on = [] { const char *e = getenv("BFF_MERGE_SPLIT"); return !e || atoi(e) != 0; }();
        const bool do_split = split_on && sparse;
        tile_masks_kernel<<<nt, 256, 0, st>>>(chunk_mask, order, n_order, mw, tile_mask, hist, hist_sorted, (int)n_pos, area,
                                             tile_hmax, tile_amin, label_id, row_sorted, area_sorted, label_sorted,
                                             init_in_tiles ? parent : nullptr);
        // Pre-pass over pairs 1, 2, 3, 5 apart in the tile order: it used to shorten the tile pass when every proven
        // edge went into the global forest at once; with local sets and early settling inside the tiles it no
        // longer pays (config 2, 4 rotating scenes: 754 scenes/s with 4 strides, 768 without) -> off unless asked for.
        static const int kStrides = [] {
            const char *e = getenv("BFF_SKELETON_STRIDES");
            const int v = e ? atoi(e) : 0;
            return v < 0 ? 0 : (v > 8 ? 8 : v);
        }();
        if (kStrides > 0) {
            dim3 sgrid((unsigned)ceil_div(n_order, 4), (unsigned)kStrides);
            uf_skeleton_kernel<<<sgrid, 256, 0, st>>>(rows, n_order, nw, order, chunk_mask, mw, area, label_id, iou_thres,
                                                      parent, kStrides);
        }
        hipError_t e = zero_async(counts, 4 * sizeof(int32_t), st);
        if (e == hipSuccess && do_split) e = zero_async(arrive, sizeof(int32_t) * (size_t)kMaxSlots * (kT * kT + 1), st);
        if (e != hipSuccess) return fail((int)e, "bff_merge_components: memset: %s", hipGetErrorString(e));
        tile_pair_filter_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, st>>>(tile_hmax, tile_amin, nt, (int)total,
                                                                               iou_thres, list1, counts);
        tile_pair_rows_kernel<<<(unsigned)ceil_div(total, 4), 256, 0, st>>>(hist_sorted, (int)n_pos, area_sorted, tile_hmax,
                                                                           tile_amin, nt, iou_thres, list1, counts,
                                                                           list2, pass2, counts + 1, sparse ? tile_mask : nullptr,
                                                                           mw, do_split ? part2 : nullptr, counts + 2, (int)cap2);
        const hipEvent_t ev0 = g_merge_start, ev1 = g_merge_stop;      // attached to the dispatch itself when set
        g_merge_start = g_merge_stop = nullptr;
        const bool two = heavy_stream && heavy_stream != stream;
        hipStream_t light = st;
        if (two) {                                                     // the tile pass goes to the heavy stream
            BFF_REQUIRE(before_heavy && after_heavy, "bff_merge_components: two streams need their two events");
            hipError_t ee = hipEventRecord(reinterpret_cast<hipEvent_t>(before_heavy), light);
            if (ee == hipSuccess) ee = hipStreamWaitEvent(as_stream(heavy_stream), reinterpret_cast<hipEvent_t>(before_heavy), 0);
            if (ee != hipSuccess) return fail((int)ee, "bff_merge_components: stream hand-over: %s", hipGetErrorString(ee));
            st = as_stream(heavy_stream);
        }
        static const int diag_mode = [] { const char *e = getenv("BFF_MERGE_DIAG"); return e ? atoi(e) : 1; }();
        // BFF_MERGE_LDS_PAD=<bytes> of unused dynamic LDS: fewer blocks of the tile pass per CU (53 KB each: three fill a CU's
        // LDS and keep every other kernel in flight off that CU)
        static const int lds_pad = [] { const char *e = getenv("BFF_MERGE_LDS_PAD"); return e ? atoi(e) : 0; }();
        if (diag && diag_mode == 2)   // block timeline only (BFF_MERGE_DIAG=2): production occupancy
            hipExtLaunchKernelGGL(merge_components_kernel<2>, dim3((unsigned)cap2), dim3(256), 0, st, ev0, ev1, 0,
                rows, n_order, nw, sparse ? tile_mask : nullptr, mw, hist_sorted, (int)n_pos, row_sorted, area_sorted,
                label_sorted, iou_thres, parent, nt, diag, list2, pass2, counts + 1, chunk_pop, do_split ? part2 : nullptr, partial, arrive);
        else if (diag)  // counters + phase clocks compiled in (a couple of registers more: one wave less per SIMD)
            hipExtLaunchKernelGGL(merge_components_kernel<1>, dim3((unsigned)cap2), dim3(256), 0, st, ev0, ev1, 0,
                rows, n_order, nw, sparse ? tile_mask : nullptr, mw, hist_sorted, (int)n_pos, row_sorted, area_sorted,
                label_sorted, iou_thres, parent, nt, diag, list2, pass2, counts + 1, chunk_pop, do_split ? part2 : nullptr, partial, arrive);
        else
            hipExtLaunchKernelGGL(merge_components_kernel<0>, dim3((unsigned)cap2), dim3(256), (unsigned)lds_pad, st, ev0, ev1, 0,
                rows, n_order, nw, sparse ? tile_mask : nullptr, mw, hist_sorted, (int)n_pos, row_sorted, area_sorted,
                label_sorted, iou_thres, parent, nt, (int32_t *)nullptr, list2, pass2, counts + 1, chunk_pop, do_split ? part2 : nullptr, partial, arrive);
        if (two) {
            hipError_t ee = hipEventRecord(reinterpret_cast<hipEvent_t>(after_heavy), st);
            if (ee == hipSuccess) ee = hipStreamWaitEvent(light, reinterpret_cast<hipEvent_t>(after_heavy), 0);
            if (ee != hipSuccess) return fail((int)ee, "bff_merge_components: stream hand-over: %s", hipGetErrorString(ee));
            st = light;
        }
    }
    if (comp) uf_flatten_kernel<<<(unsigned)ceil_div(n_rows, 256), 256, 0, st>>>(parent, n_rows, comp);
    return launched("bff_merge_components");
}

// Profiling aid (bench.py): events attached to the next tile-pass dispatch of this host thread.
extern "C" int bff_profile_next_merge(void *start_event, void *stop_event)
{
    g_merge_start = reinterpret_cast<hipEvent_t>(start_event);
    g_merge_stop = reinterpret_cast<hipEvent_t>(stop_event);
    return BFF_OK;
}

extern "C" int bff_permute_bits(const uint64_t *rows_in, int32_t n_rows, int64_t nw_in, const int32_t *idx,
                                int64_t n_out, int64_t nw_out, uint64_t *rows_out, void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && n_out >= 0 && nw_out == ceil_div(n_out, 64) && nw_in >= 0, "bff_permute_bits: bad sizes");
    if (n_rows == 0 || n_out == 0) return BFF_OK;
    BFF_REQUIRE(rows_in && idx && rows_out, "bff_permute_bits: null pointer");
    dim3 grid((unsigned)ceil_div(nw_out * 64, 256), (unsigned)n_rows);
    permute_bits_kernel<<<grid, 256, 0, as_stream(stream)>>>(rows_in, nw_in, idx, n_out, nw_out, rows_out);
    return launched("bff_permute_bits");
}

// Rows at least this long are OR-ed through their chunk flags (BFF_OR_SPARSE_MIN_NW): config 4 reads 4.8 GB of rows
// that are ~1 % occupied.  Config 2 (3125 words): round 2 measured the dense pass (234 MB) and the flagged one the same end
// to end; with four scenes in flight on the shorter chain the flagged pass is 21 vs 40-45 us and worth ~5 % of the
// throughput (the dense read competed with the other scenes' kernels for HBM), so the limit is 1024 words now.
static int64_t or_sparse_min_words()
{
    static const int64_t v = [] { const char *e = getenv("BFF_OR_SPARSE_MIN_NW"); return e ? atoll(e) : 1024ll; }();
    return v;
}

extern "C" int bff_or_reduce_groups(const uint64_t *rows, int64_t nw, const int32_t *group_offs,
                                    const int32_t *members, int32_t n_groups, int32_t max_group_size,
                                    uint64_t *out, const void *conf, int32_t conf_dtype, void *conf_mean,
                                    const uint64_t *chunk_mask, void *stream)
{
    BFF_REQUIRE(n_groups >= 0 && nw >= 0, "bff_or_reduce_groups: bad sizes");
    if (n_groups == 0) return BFF_OK;
    BFF_REQUIRE(rows && group_offs && members && out, "bff_or_reduce_groups: null pointer");
    BFF_REQUIRE((conf == nullptr) == (conf_mean == nullptr) && (conf_dtype == 0 || conf_dtype == 1),
                "bff_or_reduce_groups: conf and conf_mean go together, dtype 0 (f32) or 1 (f16)");
    if (nw == 0 && !conf) return BFF_OK;
    // z covers the largest group in slices of kOrSplit members; max_group_size is a host-known bound
    const int nz = (int)ceil_div(max_group_size > 0 ? max_group_size : 1, kOrSplit);
    if (nz > 1 && nw > 0) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(uint64_t) * (size_t)n_groups * nw, as_stream(stream));
        if (e != hipSuccess) return fail((int)e, "bff_or_reduce_groups: memset: %s", hipGetErrorString(e));
    }
    dim3 grid((unsigned)ceil_div(nw > 0 ? nw : 1, 256), (unsigned)(n_groups + (conf ? 1 : 0)), (unsigned)nz);
    const uint64_t *cm = (chunk_mask && nw >= or_sparse_min_words()) ? chunk_mask : nullptr;
    const int mw = (int)ceil_div(ceil_div(nw, kCW), 64);
    if (conf_dtype == 1)
        or_reduce_groups_kernel<__half><<<grid, 256, 0, as_stream(stream)>>>(rows, nw, group_offs, members, n_groups, out,
                                                                            (const __half *)conf, (__half *)conf_mean, cm, mw);
    else
        or_reduce_groups_kernel<float><<<grid, 256, 0, as_stream(stream)>>>(rows, nw, group_offs, members, n_groups, out,
                                                                           (const float *)conf, (float *)conf_mean, cm, mw);
    return launched("bff_or_reduce_groups");
}

extern "C" int bff_group_conf_mean(const void *conf, int32_t dtype, const int32_t *group_offs,
                                   const int32_t *members, int32_t n_groups, void *mean, void *stream)
{
    BFF_REQUIRE(n_groups >= 0 && (dtype == 0 || dtype == 1), "bff_group_conf_mean: bad arguments");
    if (n_groups == 0) return BFF_OK;
    BFF_REQUIRE(conf && group_offs && members && mean, "bff_group_conf_mean: null pointer");
    const unsigned grid = (unsigned)n_groups;
    if (dtype == 1)
        group_conf_mean_kernel<__half><<<grid, 64, 0, as_stream(stream)>>>((const __half *)conf, group_offs, members,
                                                                           n_groups, (__half *)mean);
    else
        group_conf_mean_kernel<float><<<grid, 64, 0, as_stream(stream)>>>((const float *)conf, group_offs, members,
                                                                          n_groups, (float *)mean);
    return launched("bff_group_conf_mean");
}

extern "C" int bff_overlap_ops(const int32_t *inter, const int32_t *size, int32_t k, int32_t *ops, void *stream)
{
    BFF_REQUIRE(k >= 0, "bff_overlap_ops: bad size");
    BFF_REQUIRE(ops && (k == 0 || (inter && size)), "bff_overlap_ops: null pointer");
    overlap_ops_kernel<<<1, 1024, 0, as_stream(stream)>>>(inter, size, k, ops);
    return launched("bff_overlap_ops");
}

static int launch_resolve(uint64_t *rows, int k, int64_t nw, const int32_t *size, const uint64_t *keep, int32_t *before,
                          int32_t *after, const int32_t *k_dev, hipStream_t st, const char *what)
{
    hipError_t e = zero_async(before, sizeof(int32_t) * (size_t)k, st);
    if (e == hipSuccess) e = zero_async(after, sizeof(int32_t) * (size_t)k, st);
    if (e != hipSuccess) return fail((int)e, "%s: memset: %s", what, hipGetErrorString(e));
    resolve_priority_kernel<<<(unsigned)ceil_div(nw > 0 ? nw : 1, kWave), 1024, sizeof(int) * 2 * (size_t)k, st>>>(
        rows, nw, k, size, keep, before, after, k_dev);
    return launched(what);
}

extern "C" int bff_resolve_overlaps(uint64_t *rows, int32_t k, int64_t nw, const int32_t *size, const uint64_t *keep,
                                    int32_t *before, int32_t *after, void *stream)
{
    BFF_REQUIRE(k >= 0 && nw >= 0, "bff_resolve_overlaps: bad sizes");
    BFF_LIMIT(k <= kResolveMax, "bff_resolve_overlaps: more than %d rows (use bff_overlap_ops + bff_apply_row_ops)", kResolveMax);
    if (k == 0) return BFF_OK;
    BFF_REQUIRE(rows && size && before && after, "bff_resolve_overlaps: null pointer");
    return launch_resolve(rows, k, nw, size, keep, before, after, nullptr, as_stream(stream), "bff_resolve_overlaps");
}

extern "C" int bff_resolve_overlaps_dev(uint64_t *rows, int32_t k_cap, int64_t nw, const int32_t *size, const uint64_t *keep,
                                        int32_t *before, int32_t *after, const int32_t *k_dev, void *stream)
{
    BFF_REQUIRE(k_cap > 0 && nw >= 0, "bff_resolve_overlaps_dev: bad sizes");
    BFF_LIMIT(k_cap <= kResolveMax, "bff_resolve_overlaps_dev: capacity beyond %d rows", kResolveMax);
    BFF_REQUIRE(rows && size && before && after && k_dev, "bff_resolve_overlaps_dev: null pointer");
    return launch_resolve(rows, k_cap, nw, size, keep, before, after, k_dev, as_stream(stream), "bff_resolve_overlaps_dev");
}

extern "C" int bff_resolve_overlaps_max_rows(void) { return kResolveMax; }

extern "C" int bff_apply_row_ops(uint64_t *rows, int64_t nw, const int32_t *ops, int32_t n_ops, void *stream)
{
    BFF_REQUIRE(nw >= 0, "bff_apply_row_ops: bad sizes");
    if (n_ops == 0 || nw == 0) return BFF_OK;
    BFF_REQUIRE(rows && ops, "bff_apply_row_ops: null pointer");
    apply_row_ops_kernel<<<(unsigned)ceil_div(nw, 256), 256, 0, as_stream(stream)>>>(rows, nw, ops, n_ops);
    return launched("bff_apply_row_ops");
}

extern "C" int bff_and_rows(uint64_t *rows, int32_t n_rows, int64_t nw, const uint64_t *keep, void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && nw >= 0, "bff_and_rows: bad sizes");
    if (n_rows == 0 || nw == 0) return BFF_OK;
    BFF_REQUIRE(rows && keep, "bff_and_rows: null pointer");
    dim3 grid((unsigned)ceil_div(nw, 256), (unsigned)n_rows);
    and_rows_kernel<<<grid, 256, 0, as_stream(stream)>>>(rows, nw, keep);
    return launched("bff_and_rows");
}

extern "C" int bff_gather_rows(const uint64_t *rows, const int32_t *idx, int32_t n_out, int64_t nw, uint64_t *out,
                               void *stream)
{
    BFF_REQUIRE(n_out >= 0 && nw >= 0, "bff_gather_rows: bad sizes");
    if (n_out == 0 || nw == 0) return BFF_OK;
    BFF_REQUIRE(rows && idx && out, "bff_gather_rows: null pointer");
    dim3 grid((unsigned)ceil_div(nw, 256), (unsigned)n_out);
    gather_rows_kernel<<<grid, 256, 0, as_stream(stream)>>>(rows, idx, nw, out);
    return launched("bff_gather_rows");
}

extern "C" int bff_unpack_rows(const uint64_t *rows, int32_t n_rows, int64_t nw, int64_t n_points, uint8_t *dense,
                               void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && n_points >= 0 && nw == ceil_div(n_points, 64), "bff_unpack_rows: bad sizes");
    if (n_rows == 0 || n_points == 0) return BFF_OK;
    BFF_REQUIRE(rows && dense, "bff_unpack_rows: null pointer");
    dim3 grid((unsigned)ceil_div(ceil_div(n_points, 8), 256), (unsigned)n_rows);
    unpack_rows_kernel<<<grid, 256, 0, as_stream(stream)>>>(rows, nw, n_points, dense);
    return launched("bff_unpack_rows");
}

extern "C" int bff_pack_rows(const uint8_t *dense, int32_t n_rows, int64_t n_points, int64_t nw, uint64_t *rows,
                             void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && n_points >= 0 && nw == ceil_div(n_points, 64), "bff_pack_rows: bad sizes");
    if (n_rows == 0 || n_points == 0) return BFF_OK;
    BFF_REQUIRE(rows && dense, "bff_pack_rows: null pointer");
    dim3 grid((unsigned)ceil_div(nw * 64, 256), (unsigned)n_rows);
    pack_rows_kernel<<<grid, 256, 0, as_stream(stream)>>>(dense, n_points, nw, rows);
    return launched("bff_pack_rows");
}

extern "C" int bff_rle_to_rows(const int32_t *run_start, const int32_t *run_end, const int32_t *row_run_offs,
                               int32_t n_rows, int64_t n_points, int64_t nw, uint64_t *rows, void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && n_points >= 0 && nw == ceil_div(n_points, 64), "bff_rle_to_rows: bad sizes");
    if (n_rows == 0 || nw == 0) return BFF_OK;
    BFF_REQUIRE(row_run_offs && rows, "bff_rle_to_rows: null pointer");   // run arrays may be empty (NULL)
    dim3 grid((unsigned)ceil_div(nw, 256), (unsigned)n_rows);
    rle_to_rows_kernel<<<grid, 256, 0, as_stream(stream)>>>(run_start, run_end, row_run_offs, n_points, nw, rows);
    return launched("bff_rle_to_rows");
}

extern "C" int bff_rle_count_runs(const uint64_t *rows, int32_t n_rows, int64_t nw, int32_t *n_runs, void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && nw >= 0, "bff_rle_count_runs: bad sizes");
    if (n_rows == 0) return BFF_OK;
    BFF_REQUIRE(rows && n_runs, "bff_rle_count_runs: null pointer");
    rle_count_kernel<<<n_rows, 256, 0, as_stream(stream)>>>(rows, nw, n_runs);
    return launched("bff_rle_count_runs");
}

extern "C" int bff_rle_encode_rows(const uint64_t *rows, int32_t n_rows, int64_t nw, const int64_t *run_offs,
                                   int64_t n_runs_total, int64_t *counts, void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && nw >= 0 && n_runs_total >= 0, "bff_rle_encode_rows: bad sizes");
    if (n_rows == 0 || n_runs_total == 0) return BFF_OK;
    BFF_REQUIRE(rows && run_offs && counts, "bff_rle_encode_rows: null pointer");
    rle_write_kernel<<<n_rows, 256, 0, as_stream(stream)>>>(rows, nw, run_offs, counts);
    rle_lengths_kernel<<<(unsigned)ceil_div(n_runs_total, 256), 256, 0, as_stream(stream)>>>(counts, n_runs_total);
    return launched("bff_rle_encode_rows");
}

extern "C" int bff_ids_to_rows(const int64_t *ids, int64_t n_points, const int64_t *values, int32_t n_values, int64_t nw,
                               uint64_t *rows, void *stream)
{
    BFF_REQUIRE(n_points >= 0 && n_values >= 0 && nw == ceil_div(n_points, 64), "bff_ids_to_rows: bad sizes");
    if (n_values == 0 || n_points == 0) return BFF_OK;
    BFF_REQUIRE(ids && values && rows, "bff_ids_to_rows: null pointer");
    BFF_LIMIT(n_values <= 65535, "bff_ids_to_rows: too many values");
    dim3 grid((unsigned)ceil_div(nw * 64, 256), (unsigned)n_values);
    ids_to_rows_kernel<<<grid, 256, 0, as_stream(stream)>>>(ids, n_points, values, nw, rows);
    return launched("bff_ids_to_rows");
}

extern "C" int32_t bff_group_slice_cap(int32_t n_rows, int32_t cap) { return n_rows / kOrSplit + cap + 1; }

extern "C" int bff_group_components(int32_t *comp, int32_t *parent, const int32_t *area, int32_t n_rows, float iou_thres,
                                    int32_t min_members, int32_t cap, int32_t *count, int32_t count_is_zero,
                                    int32_t *info, int32_t *sizes,
                                    int32_t *first, int32_t *offs, int32_t *members, int32_t *slices, void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && cap > 0, "bff_group_components: bad sizes");
    BFF_LIMIT(cap <= kFuseMax, "bff_group_components: at most %d groups on the device", kFuseMax);
    BFF_REQUIRE(info && sizes && first && offs && slices && (n_rows == 0 || (comp && area && count && members)),
                "bff_group_components: null pointer");
    hipStream_t st = as_stream(stream);
    if (n_rows > 0) {
        if (!count_is_zero) {
            hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t) * (size_t)n_rows, st);
            if (e != hipSuccess) return fail((int)e, "bff_group_components: memset: %s", hipGetErrorString(e));
        }
        group_count_kernel<<<(unsigned)ceil_div(n_rows, 256), 256, 0, st>>>(comp, n_rows, count, parent);
    }
    group_scan_kernel<<<1, 1024, 0, st>>>(comp, count, area, n_rows, iou_thres, min_members, cap, info, sizes, first, offs,
                                         slices, bff_group_slice_cap(n_rows, cap));
    if (n_rows > 0)
        group_members_kernel<<<(unsigned)cap, 256, 0, st>>>(comp, n_rows, info, cap, first, offs, members);
    return launched("bff_group_components");
}

extern "C" int bff_or_reduce_grouped(const uint64_t *rows, int64_t nw, int32_t n_rows, const int32_t *info, int32_t cap,
                                     const int32_t *offs, const int32_t *members, const int32_t *slices, uint64_t *out,
                                     const void *conf, int32_t conf_dtype, void *conf_mean, const uint64_t *chunk_mask,
                                     void *stream)
{
    BFF_REQUIRE(nw >= 0 && n_rows >= 0 && cap > 0, "bff_or_reduce_grouped: bad sizes");
    BFF_REQUIRE(rows && info && offs && members && slices && out, "bff_or_reduce_grouped: null pointer");
    BFF_REQUIRE((conf == nullptr) == (conf_mean == nullptr) && (conf_dtype == 0 || conf_dtype == 1),
                "bff_or_reduce_grouped: conf and conf_mean go together, dtype 0 (f32) or 1 (f16)");
    hipStream_t st = as_stream(stream);
    if (nw > 0) {
        hipError_t e = zero_async(out, sizeof(uint64_t) * (size_t)cap * nw, st);
        if (e != hipSuccess) return fail((int)e, "bff_or_reduce_grouped: memset: %s", hipGetErrorString(e));
    }
    const int slice_cap = bff_group_slice_cap(n_rows, cap);
    dim3 grid((unsigned)ceil_div(nw > 0 ? nw : 1, 256), (unsigned)(slice_cap + 1));
    BFF_LIMIT(slice_cap + 1 <= 65535, "bff_or_reduce_grouped: too many member slices");
    // through the chunk flags only when the rows are long (config 2: the dense pass runs at HBM speed and the flagged
    // form was measured slower; config 4: 4.8 GB of rows, ~1 % occupied)
    const uint64_t *cm = (chunk_mask && nw >= or_sparse_min_words()) ? chunk_mask : nullptr;
    const int mw = (int)ceil_div(ceil_div(nw, kCW), 64);
    if (conf_dtype == 1)
        or_reduce_grouped_kernel<__half><<<grid, 256, 0, st>>>(rows, nw, info, cap, offs, members, slices, slice_cap, out,
                                                              (const __half *)conf, (__half *)conf_mean, cm, mw);
    else
        or_reduce_grouped_kernel<float><<<grid, 256, 0, st>>>(rows, nw, info, cap, offs, members, slices, slice_cap, out,
                                                             (const float *)conf, (float *)conf_mean, cm, mw);
    return launched("bff_or_reduce_grouped");
}

extern "C" int bff_scatter_bits(const uint64_t *rows_in, int32_t n_rows, int64_t nw_in, const int32_t *perm, int64_t n,
                                int64_t nw_out, uint64_t *rows_out, const int32_t *k_dev, void *stream)
{
    BFF_REQUIRE(n_rows >= 0 && n >= 0 && nw_out == ceil_div(n, 64) && nw_in >= 0, "bff_scatter_bits: bad sizes");
    if (n_rows == 0 || n == 0 || nw_in == 0) return BFF_OK;
    BFF_REQUIRE(rows_in && perm && rows_out, "bff_scatter_bits: null pointer");
    BFF_LIMIT(n_rows <= 65535, "bff_scatter_bits: too many rows");
    dim3 grid((unsigned)ceil_div(nw_in, 256), (unsigned)n_rows);
    scatter_bits_kernel<<<grid, 256, 0, as_stream(stream)>>>(rows_in, nw_in, perm, n, nw_out, rows_out, k_dev);
    return launched("bff_scatter_bits");
}
