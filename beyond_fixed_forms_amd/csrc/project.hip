// Per-frame projection sweep and 2-D RLE decode (include/bff_hip.h: a1, a2-a7, a15).
//
// Data layout in HBM
//   xyz        f64 [3][n_pad]            structure of arrays: a wave reads 3 x 512 B contiguous
//   depth      f32 [n_depth][H*W]        gathered at the projected pixel (4 B / visible point)
//   maskbits   u32|u64 [n_mviews][H*W]   bit b = mask b of that frame covers the pixel (one gather
//                                         returns all masks of the frame)
//   rows       u64 [n_rows][nw]          instance bit rows: a wave's 64 points are exactly one word,
//                                         so __ballot() of "bit b set" IS the output word
// Kernel shape: 256 threads = 4 waves own 1024 consecutive points (16 row words = one 128-B line per row);
// wave w owns the 4 consecutive words 4w..4w+3 of that line (its 32-B sector), 4 points per thread kept in
// registers across the frames of the block's frame tile; pose and intrinsics are wave-uniform (scalar loads /
// kernel arguments).  The frame loop has no block barrier: a wave transposes its ballots through a private LDS
// slice and stores its own sector of every row of the frame.
#include "common.h"

namespace bff {

constexpr int kBlock = 256;
constexpr int kPPT = 4;                          // points per thread
constexpr int kWordsPerBlock = (kBlock / kWave) * kPPT;   // 16 row words per block
constexpr int kPtsPerBlock = kWordsPerBlock * kWave;       // 1024

struct Intrinsics { double k[9]; };

__device__ __forceinline__ void lds_phase_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <typename WordT>
__global__ __launch_bounds__(kBlock) void project_views_kernel(
    const double *__restrict__ xyz, int64_t n_points, int64_t n_pad,
    const double *__restrict__ inv_pose, Intrinsics K, int n_frames, int frames_per_block,
    const float *__restrict__ depth, const int32_t *__restrict__ depth_index, int H, int W, double thresh,
    const WordT *__restrict__ maskbits, const uint32_t *__restrict__ segmap, int64_t seg_words,
    const int32_t *__restrict__ frame_mask,
    const int32_t *__restrict__ frame_rowbase, const int32_t *__restrict__ frame_nmask,
    const int32_t *__restrict__ frame_flags,
    uint64_t *__restrict__ rows, int64_t nw, int32_t *__restrict__ masked_count,
    int32_t *__restrict__ viewed_count)
{
    // per wave: [bit][kPPT words] transposition buffer for the wave's sector of the frame's rows
    __shared__ uint64_t stage_all[kBlock / kWave][sizeof(WordT) * 8][kPPT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t word0 = (int64_t)blockIdx.x * kWordsPerBlock + (int64_t)wave * kPPT;   // the wave's first word
    const int64_t hw = (int64_t)H * W;
    uint64_t (*stage)[kPPT] = stage_all[wave];

    double px[kPPT], py[kPPT], pz[kPPT];
    bool valid[kPPT];
    int mcount[kPPT], vcount[kPPT];
#pragma unroll
    for (int j = 0; j < kPPT; ++j) {
        const int64_t n = (word0 + j) * kWave + lane;
        valid[j] = n < n_points;
        const int64_t m = valid[j] ? n : 0;
        px[j] = xyz[m];
        py[j] = xyz[n_pad + m];
        pz[j] = xyz[2 * n_pad + m];
        mcount[j] = 0;
        vcount[j] = 0;
    }

    const int f0 = blockIdx.y * frames_per_block;
    const int f1 = min(n_frames, f0 + frames_per_block);
    const double dW = (double)W, dH = (double)H;
    for (int f = f0; f < f1; ++f) {
        const double *P = inv_pose + 16 * (int64_t)f;
        const float *dimg = depth + (int64_t)depth_index[f] * hw;
        const int mi = maskbits ? frame_mask[f] : -1;
        const bool has_masks = mi >= 0;
        const WordT *mimg = has_masks ? maskbits + (int64_t)mi * hw : nullptr;
        const uint32_t *smap = (segmap && has_masks) ? segmap + (int64_t)mi * seg_words : nullptr;
        const int nm = has_masks ? frame_nmask[f] : 0;
        const bool count_viewed = (frame_flags[f] & 1) != 0;
        // Three phases per frame so that a thread's gathers are all in flight together (the sweep is bound by
        // memory latency x concurrency, not by issue): (1) geometry of the 4 points, (2) depth + segment-bitmap
        // gathers of every in-bounds point, (3) mask-word gathers of every visible point in a masked segment.
        int pix[kPPT];
        double cz[kPPT];
#pragma unroll
        for (int j = 0; j < kPPT; ++j) {
            // k-ascending fma chains from +0.0: bit-identical to the reference's dgemm (header contract)
            const double cx = fma(P[3], 1.0, fma(P[2], pz[j], fma(P[1], py[j], fma(P[0], px[j], 0.0))));
            const double cy = fma(P[7], 1.0, fma(P[6], pz[j], fma(P[5], py[j], fma(P[4], px[j], 0.0))));
            cz[j] = fma(P[11], 1.0, fma(P[10], pz[j], fma(P[9], py[j], fma(P[8], px[j], 0.0))));
            const double p0 = fma(K.k[2], cz[j], fma(K.k[1], cy, fma(K.k[0], cx, 0.0)));
            const double p1 = fma(K.k[5], cz[j], fma(K.k[4], cy, fma(K.k[3], cx, 0.0)));
            const double u = rint(p0 / cz[j]);
            const double v = rint(p1 / cz[j]);
            const bool inb = valid[j] && (u >= 0.0) && (u < dW) && (v >= 0.0) && (v < dH);   // NaN fails
            pix[j] = inb ? (int)v * W + (int)u : -1;       // H*W < 2^31 (checked by the entry point)
        }
        float dval[kPPT];
        uint32_t sbits[kPPT];
#pragma unroll
        for (int j = 0; j < kPPT; ++j) {
            dval[j] = 0.0f;
            sbits[j] = 0xffffffffu;
            if (pix[j] >= 0) {
                dval[j] = dimg[pix[j]];
                // segments without any mask pixel were never written by the decoder: consult its bitmap
                if (smap) sbits[j] = smap[pix[j] >> 12];
            }
        }
        bool vis[kPPT];
        WordT wv[kPPT];
#pragma unroll
        for (int j = 0; j < kPPT; ++j) {
            vis[j] = (pix[j] >= 0) && (dval[j] != 0.0f) && (fabs(cz[j] - (double)dval[j]) < thresh);
            wv[j] = 0;
            if (vis[j] && mimg && ((sbits[j] >> ((pix[j] >> 7) & 31)) & 1)) wv[j] = mimg[pix[j]];
        }
        WordT present = 0;
#pragma unroll
        for (int j = 0; j < kPPT; ++j) {
            if (count_viewed) vcount[j] += vis[j] ? 1 : 0;
            mcount[j] += (sizeof(WordT) == 8) ? __popcll((uint64_t)wv[j]) : __popc((uint32_t)wv[j]);
            present |= wv[j];
        }
        if (has_masks) {                                   // wave-uniform
            // the wave's 32-B sector (kPPT words) of each of the frame's nm rows: lane -> (row lane/4 + 16 i,
            // word lane%4).  Most waves see no mask at all in a given frame: the rows were zeroed by the entry point.
            const int64_t rb = frame_rowbase[f];
            const int wd = lane & (kPPT - 1);
            const bool in_rows = word0 + wd < nw;
            if (__ballot(present != 0)) {
                // lane b collects the ballots of bit b: OR the words across the wave, visit the set bits only
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) present |= __shfl_xor(present, d);
                uint64_t mine[kPPT] = {};
                while (present) {
                    const int b = (sizeof(WordT) == 8) ? __ffsll((unsigned long long)present) - 1
                                                       : __ffs((unsigned)present) - 1;
                    present &= present - 1;
#pragma unroll
                    for (int j = 0; j < kPPT; ++j) {
                        const uint64_t bal = __ballot((wv[j] >> b) & 1);
                        if (lane == b) mine[j] = bal;
                    }
                }
                if (lane < (int)(sizeof(WordT) * 8)) {
#pragma unroll
                    for (int j = 0; j < kPPT; ++j) stage[lane][j] = mine[j];
                }
                lds_phase_fence();                         // wave-private slice: LDS ops complete in issue order
                for (int b = lane / kPPT; b < nm; b += kWave / kPPT)
                    if (in_rows) rows[(rb + b) * nw + word0 + wd] = stage[b][wd];
                lds_phase_fence();                         // reads done before the next frame's writes
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kPPT; ++j) {
        const int64_t n = (word0 + j) * kWave + lane;
        if (valid[j]) {
            if (masked_count && mcount[j]) atomicAdd(masked_count + n, mcount[j]);
            if (viewed_count && vcount[j]) atomicAdd(viewed_count + n, vcount[j]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 2-D RLE -> mask words, wave-synchronous.  Every wave owns a band of kWaveChunks x 512 pixels of one
// mask-view and a private 512-word LDS slice; lane b < n_masks keeps mask b's run cursor (current + next
// run in registers).  Per 512-pixel chunk the runs enter LDS as XOR toggles at their clipped start and end,
// an XOR prefix scan (4 + 4 words per lane, wave shuffles) turns toggles into coverage, and the chunk is
// written with two fully coalesced 1-KiB (u32) stores.  There is no block barrier: waves never wait for each
// other, LDS accesses of one wave complete in issue order.  Segments of 128 pixels without any mask pixel
// are not written when a segment bitmap is requested.  HBM traffic = at most one write of the image.
constexpr int kWaveChunk = 512;              // pixels per wave chunk (4 + 4 per lane)
constexpr int kWaveChunks = 32;              // chunks per wave band (16384 pixels)

template <typename WordT>
__global__ __launch_bounds__(kBlock) void rle_to_maskbits_kernel(
    const int32_t *__restrict__ run_start, const int32_t *__restrict__ run_end,
    const int32_t *__restrict__ mask_run_offs, const int32_t *__restrict__ view_mask_offs,
    int64_t n_pixels, WordT *__restrict__ maskbits, uint32_t *__restrict__ segmap, int64_t seg_words)
{
    __shared__ WordT lds[kBlock / kWave][kWaveChunk];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    WordT *bits = lds[wave];
    const int v = blockIdx.y;
    const int g0 = view_mask_offs[v];
    const int nm = view_mask_offs[v + 1] - g0;
    const int64_t band0 = ((int64_t)blockIdx.x * (kBlock / kWave) + wave) * kWaveChunk * kWaveChunks;
    if (band0 >= n_pixels) return;                   // wave-uniform
    WordT *img = maskbits + (int64_t)v * n_pixels;

    constexpr int64_t kNone = INT64_MAX;
    int cur = 0, hi = 0;
    int64_t rs = kNone, re = kNone, ns = kNone, ne = kNone;
    if (lane < nm) {
        int lo = mask_run_offs[g0 + lane];
        hi = mask_run_offs[g0 + lane + 1];
        int r = hi;                                   // first run with end > band0
        while (lo < r) {
            const int mid = (lo + r) >> 1;
            if ((int64_t)run_end[mid] > band0) r = mid; else lo = mid + 1;
        }
        cur = r;
        if (cur < hi) { rs = run_start[cur]; re = run_end[cur]; }
        if (cur + 1 < hi) { ns = run_start[cur + 1]; ne = run_end[cur + 1]; }
    }
    constexpr int kHalf = kWaveChunk / 2, kQ = 4;     // lane owns words [4l, 4l+4) of each 256-word half
#pragma unroll
    for (int k = 0; k < kQ; ++k) { bits[lane * kQ + k] = 0; bits[kHalf + lane * kQ + k] = 0; }
    lds_phase_fence();
    for (int c = 0; c < kWaveChunks; ++c) {
        const int64_t c0 = band0 + (int64_t)c * kWaveChunk;
        if (c0 >= n_pixels) break;
        const int64_t c1 = min(c0 + kWaveChunk, n_pixels);
        if (lane < nm) {
            const WordT bit = (WordT)1 << lane;
            while (rs < c1) {
                atomicXor(&bits[(int)(max(rs, c0) - c0)], bit);
                if (re < c1) atomicXor(&bits[(int)(re - c0)], bit);
                if (re > c1) break;                   // run continues into the next chunk
                ++cur;
                rs = ns; re = ne;
                if (cur + 1 < hi) { ns = run_start[cur + 1]; ne = run_end[cur + 1]; } else { ns = ne = kNone; }
            }
        }
        lds_phase_fence();                            // toggles of all lanes are in LDS
        WordT loc[2][kQ], tot[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            WordT acc = 0;
#pragma unroll
            for (int k = 0; k < kQ; ++k) {
                acc ^= bits[h * kHalf + lane * kQ + k];
                loc[h][k] = acc;
                bits[h * kHalf + lane * kQ + k] = 0;  // ready for the next chunk
            }
            tot[h] = acc;
        }
        lds_phase_fence();                            // re-zeroing is ordered before the next chunk's toggles
        WordT incl[2] = {tot[0], tot[1]};
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const WordT u0 = __shfl_up(incl[0], d), u1 = __shfl_up(incl[1], d);
            if (lane >= d) { incl[0] ^= u0; incl[1] ^= u1; }
        }
        const WordT half0_total = __shfl(incl[0], kWave - 1);
        const WordT carry0 = incl[0] ^ tot[0];                       // exclusive prefix within the half
        const WordT carry1 = incl[1] ^ tot[1] ^ half0_total;         // second half continues the first
        using Vec = __attribute__((ext_vector_type(4))) uint32_t;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const WordT carry = h ? carry1 : carry0;
            const int64_t p0 = c0 + h * kHalf + (int64_t)lane * kQ;
            WordT outv[kQ];
            WordT any = 0;
#pragma unroll
            for (int k = 0; k < kQ; ++k) { outv[k] = loc[h][k] ^ carry; any |= outv[k]; }
            if (segmap) {
                // 128-pixel segments = 32 consecutive lanes: all-zero segments are not stored at all, the
                // sweep learns from the bitmap (one bit per segment) that there is nothing to gather there
                const uint64_t nz = __ballot(any != 0);
                const uint32_t half_nz = (uint32_t)(nz >> (lane & 32));
                if ((lane & 31) == 0 && p0 < c1) {
                    const int64_t seg = p0 >> 7;
                    if (half_nz) atomicOr(segmap + (int64_t)v * seg_words + (seg >> 5), 1u << (seg & 31));
                }
                if (!half_nz) continue;
            }
            if (p0 + kQ <= c1) {
                const Vec *src = reinterpret_cast<const Vec *>(outv);
                Vec *dst = reinterpret_cast<Vec *>(img + p0);      // chunk starts are multiples of 512 words
#pragma unroll
                for (int k = 0; k < (int)(kQ * sizeof(WordT) / 16); ++k) dst[k] = src[k];
            } else {
#pragma unroll
                for (int k = 0; k < kQ; ++k)
                    if (p0 + k < c1) img[p0 + k] = outv[k];
            }
        }
    }
}

}  // namespace bff

using namespace bff;

extern "C" int bff_rle_to_maskbits(const int32_t *run_start, const int32_t *run_end, const int32_t *mask_run_offs,
                                   const int32_t *view_mask_offs, int32_t n_views, int64_t n_pixels,
                                   int32_t word_bits, void *maskbits, uint32_t *segmap, void *stream)
{
    BFF_REQUIRE(n_views >= 0 && n_pixels > 0, "bff_rle_to_maskbits: bad sizes");
    BFF_REQUIRE(word_bits == 32 || word_bits == 64, "bff_rle_to_maskbits: word_bits must be 32 or 64");
    BFF_LIMIT(n_pixels < (1ll << 31), "bff_rle_to_maskbits: image larger than 2^31 pixels");
    if (n_views == 0) return BFF_OK;
    BFF_REQUIRE(mask_run_offs && view_mask_offs && maskbits, "bff_rle_to_maskbits: null pointer");   // run arrays may be empty (NULL)
    dim3 grid((unsigned)ceil_div(n_pixels, (int64_t)kWaveChunk * kWaveChunks * (kBlock / kWave)), (unsigned)n_views);
    const int64_t seg_words = ceil_div(ceil_div(n_pixels, 128), 32);
    if (segmap) {
        hipError_t e = hipMemsetAsync(segmap, 0, sizeof(uint32_t) * (size_t)n_views * seg_words, as_stream(stream));
        if (e != hipSuccess) return fail((int)e, "bff_rle_to_maskbits: memset: %s", hipGetErrorString(e));
    }
    if (word_bits == 32)
        rle_to_maskbits_kernel<uint32_t><<<grid, kBlock, 0, as_stream(stream)>>>(
            run_start, run_end, mask_run_offs, view_mask_offs, n_pixels, (uint32_t *)maskbits, segmap, seg_words);
    else
        rle_to_maskbits_kernel<uint64_t><<<grid, kBlock, 0, as_stream(stream)>>>(
            run_start, run_end, mask_run_offs, view_mask_offs, n_pixels, (uint64_t *)maskbits, segmap, seg_words);
    return launched("bff_rle_to_maskbits");
}

extern "C" int bff_project_views(const double *xyz, int64_t n_points, int64_t n_pad,
                                 const double *inv_pose, const double *cam_intr_host, int32_t n_frames,
                                 const float *depth, const int32_t *depth_index, int32_t height, int32_t width,
                                 double depth_thresh,
                                 const void *maskbits, const uint32_t *segmap, int32_t word_bits,
                                 const int32_t *frame_mask, const int32_t *frame_rowbase, const int32_t *frame_nmask,
                                 const int32_t *frame_flags,
                                 uint64_t *rows, int64_t n_rows, int64_t nw,
                                 int32_t *masked_count, int32_t *viewed_count, void *stream)
{
    BFF_REQUIRE(n_points >= 0 && n_pad >= n_points && n_frames >= 0, "bff_project_views: bad sizes");
    BFF_REQUIRE(height > 0 && width > 0, "bff_project_views: bad image size");
    BFF_LIMIT((int64_t)height * width < (1ll << 31), "bff_project_views: image larger than 2^31 pixels");
    if (n_points == 0 || n_frames == 0) return BFF_OK;
    BFF_REQUIRE(xyz && inv_pose && cam_intr_host && depth && depth_index && frame_flags, "bff_project_views: null pointer");
    BFF_REQUIRE(nw == ceil_div(n_points, 64), "bff_project_views: nw must be ceil(n_points/64)");
    if (maskbits) {
        BFF_REQUIRE(word_bits == 32 || word_bits == 64, "bff_project_views: word_bits must be 32 or 64");
        BFF_REQUIRE(frame_mask && frame_rowbase && frame_nmask && rows && n_rows >= 0, "bff_project_views: mask frames need row outputs");
    }
    if (maskbits && n_rows > 0) {       // the kernel stores only the sectors in which a wave saw a mask bit
        hipError_t e = hipMemsetAsync(rows, 0, sizeof(uint64_t) * (size_t)n_rows * (size_t)nw, as_stream(stream));
        if (e != hipSuccess) return fail((int)e, "bff_project_views: memset: %s", hipGetErrorString(e));
    }
    Intrinsics K;
    for (int i = 0; i < 9; ++i) K.k[i] = cam_intr_host[i];
    const int64_t gx = ceil_div(n_points, kPtsPerBlock);
    int fpb = (int)((int64_t)n_frames * gx / 4096);      // keep >= ~4096 blocks in flight
    fpb = fpb < 1 ? 1 : (fpb > 8 ? 8 : fpb);
    dim3 grid((unsigned)gx, (unsigned)ceil_div(n_frames, fpb));
    const int64_t seg_words = ceil_div(ceil_div((int64_t)height * width, 128), 32);
    if (!maskbits || word_bits == 32)
        project_views_kernel<uint32_t><<<grid, kBlock, 0, as_stream(stream)>>>(
            xyz, n_points, n_pad, inv_pose, K, n_frames, fpb, depth, depth_index, height, width, depth_thresh,
            (const uint32_t *)maskbits, segmap, seg_words, frame_mask, frame_rowbase, frame_nmask, frame_flags, rows, nw,
            masked_count, viewed_count);
    else
        project_views_kernel<uint64_t><<<grid, kBlock, 0, as_stream(stream)>>>(
            xyz, n_points, n_pad, inv_pose, K, n_frames, fpb, depth, depth_index, height, width, depth_thresh,
            (const uint64_t *)maskbits, segmap, seg_words, frame_mask, frame_rowbase, frame_nmask, frame_flags, rows, nw,
            masked_count, viewed_count);
    return launched("bff_project_views");
}
