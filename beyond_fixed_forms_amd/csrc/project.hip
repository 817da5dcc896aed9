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
#include <hip/hip_ext.h>

#include "common.h"

namespace bff {

constexpr int kBlock = 256;
constexpr int kPPT = 4;                          // points per thread
constexpr int kWordsPerBlock = (kBlock / kWave) * kPPT;   // 16 row words per block
constexpr int kPtsPerBlock = kWordsPerBlock * kWave;       // 1024

struct Intrinsics { double k[9]; };

// Depth kept as the PNGs store it (P:431-436): uint16 millimetres at the sensor's resolution.  The sweep then evaluates
// `astype(f32) / 1000` and the two 2-tap passes of the bilinear resize to (H, W) PER POINT, at the pixel the point
// projects to, with exactly the float32 operations of depth_resize_kernel (filter.hip) / io.resize_bilinear_f32 -- the
// (H, W) float image (8x the bytes of the source) is never built.  hs == H and ws == W: the resize is the identity.
struct RawDepth {
    const uint32_t *taps;       // device table of the resize, 3 words per destination column then per destination row:
                                //   column u: (texel offset of its left tap inside a source row, of its right tap, fraction bits)
                                //   row v:    (texel offset of its upper source row, of its lower one, fraction bits)
                                // a tap's texel = row offset + column offset; offsets follow the frames' layout (row-major,
                                // or 8 x 8-texel tiles: ((y >> 3) * tw + (x >> 3)) * 64 + (y & 7) * 8 + (x & 7)); fractions and
                                // tap indices as io._axis_taps (sensor_taps_kernel below)
    int n_taps;                 // W + H
    int texel_f32;              // frames hold float32 metres (`astype(f32) / 1000` done once per texel at ingestion) instead of
                                // the uint16 millimetres themselves
    int64_t frame_stride;       // texels per frame (padded to whole tiles when tiled)
    float scale;                // 1000
};

// io._axis_taps for one destination index: f = (float)((d + 0.5) * scale - 0.5) (float64 products and differences, each
// rounded once; no contraction), i0 = floor(f), a = f - i0 in float32; x axis: border columns are copied (a = 0), y axis:
// the fraction is kept and the two row indices are clamped.
template <bool kX>
__device__ __forceinline__ void axis_tap(int d, double scale, int n_src, int &i0, int &i1, float &a)
{
    const float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
    const float fl = floorf(f);
    int s = (int)fl;
    a = __fsub_rn(f, fl);
    if (kX) {
        if (s < 0) { s = 0; a = 0.0f; }
        else if (s >= n_src - 1) { s = n_src - 1; a = 0.0f; }
        i0 = s;
        i1 = min(s + 1, n_src - 1);
    } else {
        i1 = min(max(s + 1, 0), n_src - 1);
        i0 = min(max(s, 0), n_src - 1);
    }
}

// the table RawDepth::taps points to, for source frames of hs x ws texels resized to H x W (tiled != 0: tile layout)
__global__ void sensor_taps_kernel(int hs, int ws, int H, int W, int tiled, double sx, double sy, uint32_t *__restrict__ table)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W + H) return;
    const int tw = (ws + 7) / 8;
    int i0, i1;
    float a;
    uint32_t p0, p1;
    if (i < W) {
        axis_tap<true>(i, sx, ws, i0, i1, a);
        p0 = tiled ? (uint32_t)((i0 >> 3) * 64 + (i0 & 7)) : (uint32_t)i0;
        p1 = tiled ? (uint32_t)((i1 >> 3) * 64 + (i1 & 7)) : (uint32_t)i1;
    } else {
        axis_tap<false>(i - W, sy, hs, i0, i1, a);
        p0 = tiled ? (uint32_t)((i0 >> 3) * tw * 64 + (i0 & 7) * 8) : (uint32_t)(i0 * ws);
        p1 = tiled ? (uint32_t)((i1 >> 3) * tw * 64 + (i1 & 7) * 8) : (uint32_t)(i1 * ws);
    }
    table[3 * i] = p0;
    table[3 * i + 1] = p1;
    table[3 * i + 2] = __float_as_uint(a);
}

// the resized depth value at one pixel from its four source texels (same operation order as depth_resize_kernel)
__device__ __forceinline__ float bilinear_depth(float s00, float s01, float s10, float s11, float a, float b)
{
    const float one_a = __fsub_rn(1.0f, a), one_b = __fsub_rn(1.0f, b);
    const float r0 = __fadd_rn(__fmul_rn(s00, one_a), __fmul_rn(s01, a));
    const float r1 = __fadd_rn(__fmul_rn(s10, one_a), __fmul_rn(s11, a));
    return __fadd_rn(__fmul_rn(r0, one_b), __fmul_rn(r1, b));
}

// one source texel in metres: float32 frames hold it, uint16 frames hold millimetres (P:432-435: astype(f32) / 1000)
__device__ __forceinline__ float sensor_texel(const void *frame, uint32_t t, const RawDepth &r)
{
    return r.texel_f32 ? reinterpret_cast<const float *>(frame)[t]
                       : __fdiv_rn((float)reinterpret_cast<const uint16_t *>(frame)[t], r.scale);
}

__device__ __forceinline__ void lds_phase_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <typename WordT, bool kLabels, bool kRaw>
__global__ __launch_bounds__(kBlock) void project_views_kernel(
    const double *__restrict__ xyz, int64_t n_points, int64_t n_pad,
    const double *__restrict__ inv_pose, Intrinsics K, int n_frames, int frames_per_block,
    const void *__restrict__ depth, RawDepth raw, const int32_t *__restrict__ depth_index, int H, int W, double thresh,
    const WordT *__restrict__ maskbits, const uint8_t *__restrict__ labels, int64_t label_stride,
    const uint32_t *__restrict__ segmap, int64_t seg_words,
    const int32_t *__restrict__ frame_mask,
    const int32_t *__restrict__ frame_rowbase, const int32_t *__restrict__ frame_nmask,
    const int32_t *__restrict__ frame_flags,
    uint64_t *__restrict__ rows, int64_t nw, uint64_t *__restrict__ chunk_mask, int mw,
    int32_t *__restrict__ masked_count, int32_t *__restrict__ viewed_count,
    const double *__restrict__ tile_bounds)
{
    // per wave: [bit][kPPT words] transposition buffer for the wave's sector of the frame's rows
    __shared__ uint64_t stage_all[kBlock / kWave][sizeof(WordT) * 8][kPPT];
    extern __shared__ uint32_t s_taps[];                   // kRaw: the resize's tap table (RawDepth::taps), 12 bytes per
                                                           // destination column / row: two LDS reads replace ~25 instructions
    if (kRaw) {
        // 16 bytes per load, all of a thread's loads in flight before the first store: the fill costs a block one memory
        // latency (word by word it was ~27 dependent round trips: +330 us on config 4's 73 k blocks)
        const int n_vec = (3 * raw.n_taps + 3) / 4;           // the table is padded to whole uint4
        const uint4 *src = reinterpret_cast<const uint4 *>(raw.taps);
        uint4 *dst = reinterpret_cast<uint4 *>(s_taps);
        constexpr int kFill = 1;                               // 8 x 256 x 16 B = 32 KB per round
        for (int base = 0; base < n_vec; base += kFill * kBlock) {
            uint4 v[kFill];
#pragma unroll
            for (int q = 0; q < kFill; ++q) {
                const int i = base + q * kBlock + (int)threadIdx.x;
                if (i < n_vec) v[q] = src[i];
            }
#pragma unroll
            for (int q = 0; q < kFill; ++q) {
                const int i = base + q * kBlock + (int)threadIdx.x;
                if (i < n_vec) dst[i] = v[q];
            }
        }
        __syncthreads();
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Block -> XCD mapping is the plain round-robin of the dispatch order.  Both XCD-aware mappings were measured on
    // config 2 and lost: an eighth of the (spatially sorted) cloud per XCD makes the work per XCD uneven -- frustum
    // culling empties whole regions of a frame and dispatch is in order -- (0.34 -> 0.8 ms); one frame tile per XCD, so
    // that a depth / mask line is fetched by one L2 only, ran 0.338 -> 0.345 ms alone and 0.44 -> 0.55 ms beside the other
    // scenes' kernels: the lines re-fetched by other XCDs come out of the Infinity Cache, not HBM.
    const int64_t bx = blockIdx.x;
    const int f0 = blockIdx.y * frames_per_block;
    const int64_t word0 = bx * kWordsPerBlock + (int64_t)wave * kPPT;   // the wave's first word
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

    const int f1 = min(n_frames, f0 + frames_per_block);
    const double dW = (double)W, dH = (double)H;
    // Frustum culling of the wave's 256 points against the <= 8 frames of the tile, all at once: lane = 8 * frame
    // + corner of the points' bounding box.  A point can only be visible if its pixel is in bounds, i.e. (exact
    // arithmetic) it lies in the double cone  {L_i >= 0 for all i, c_z > 0}  u  {L_i <= 0 for all i, c_z < 0}  with
    //   L_1 = p_0 + (1/2 + m) c_z,  L_2 = (W - 1/2 + m) c_z - p_0,  L_3 = p_1 + (1/2 + m) c_z,  L_4 = (H - 1/2 + m) c_z - p_1
    // (the two image borders in u and in v, widened by m = 0.01 pixel; the reference has no z > 0 test, hence both
    // cones).  Each L_i is an affine function of the world point, so its value anywhere in the box is a convex
    // combination of the 8 corner values: if some L_i < -delta at every corner the box misses the front cone, if
    // some L_j > +delta at every corner it misses the back cone; a frame is skipped only when both hold.  delta =
    // 1e-6 and m exceed the float64 rounding of these forms (~1e-11) by orders of magnitude; NaN / inf corner
    // values compare false and keep the frame.  Exact: a skipped frame has no in-bounds point in this wave.
    uint64_t culled = 0;                                   // bit 8 k set: frame f0 + k cannot see this wave's points
    if (tile_bounds) {
        const double *bb = tile_bounds + 6 * (word0 / kPPT);
        const int corner = lane & 7, fk = lane >> 3;
        const double bx = (corner & 1) ? bb[3] : bb[0], by = (corner & 2) ? bb[4] : bb[1], bz = (corner & 4) ? bb[5] : bb[2];
        double l1 = 0.0, l2 = 0.0, l3 = 0.0, l4 = 0.0;
        if (f0 + fk < f1) {
            const double *P = inv_pose + 16 * (int64_t)(f0 + fk);
            const double cx = fma(P[2], bz, fma(P[1], by, P[0] * bx)) + P[3];
            const double cy = fma(P[6], bz, fma(P[5], by, P[4] * bx)) + P[7];
            const double cz = fma(P[10], bz, fma(P[9], by, P[8] * bx)) + P[11];
            const double p0 = fma(K.k[2], cz, fma(K.k[1], cy, K.k[0] * cx));
            const double p1 = fma(K.k[5], cz, fma(K.k[4], cy, K.k[3] * cx));
            constexpr double m = 0.01;
            l1 = fma(0.5 + m, cz, p0);
            l2 = fma(dW - 0.5 + m, cz, -p0);
            l3 = fma(0.5 + m, cz, p1);
            l4 = fma(dH - 0.5 + m, cz, -p1);
        }
        constexpr double delta = 1e-6;
        auto all8 = [](uint64_t b) {                       // bit 8 k of the result = all 8 bits of byte k set
            b &= b >> 1; b &= b >> 2; b &= b >> 4;
            return b & 0x0101010101010101ull;
        };
        const uint64_t neg = all8(__ballot(l1 < -delta)) | all8(__ballot(l2 < -delta)) |
                             all8(__ballot(l3 < -delta)) | all8(__ballot(l4 < -delta));
        const uint64_t pos = all8(__ballot(l1 > delta)) | all8(__ballot(l2 > delta)) |
                             all8(__ballot(l3 > delta)) | all8(__ballot(l4 > delta));
        culled = neg & pos;
    }
    for (int f = f0; f < f1; ++f) {
        if ((culled >> (8 * (f - f0))) & 1) continue;      // wave-uniform
        const double *P = inv_pose + 16 * (int64_t)f;
        const float *dimg = kRaw ? nullptr : reinterpret_cast<const float *>(depth) + (int64_t)depth_index[f] * hw;
        const char *rimg = kRaw ? reinterpret_cast<const char *>(depth) +
                                  (int64_t)depth_index[f] * raw.frame_stride * (raw.texel_f32 ? 4 : 2) : nullptr;
        const int mi = maskbits ? frame_mask[f] : -1;
        const bool has_masks = mi >= 0;
        const WordT *mimg = has_masks ? maskbits + (int64_t)mi * hw : nullptr;
        const uint8_t *limg = (kLabels && has_masks) ? labels + (int64_t)mi * label_stride : nullptr;   // wave-uniform
        const uint32_t *smap = (segmap && has_masks) ? segmap + (int64_t)mi * seg_words * (kLabels ? 2 : 1) : nullptr;
        const int nm = has_masks ? frame_nmask[f] : 0;
        const bool count_viewed = (frame_flags[f] & 1) != 0;
        // Three phases per frame so that a thread's gathers are all in flight together (the sweep is bound by
        // memory latency x concurrency, not by issue): (1) geometry of the 4 points, (2) depth + segment-bitmap
        // gathers of every in-bounds point, (3) mask-word gathers of every visible point in a masked segment.
        int pix[kPPT];
        int pu[kPPT];                                       // kRaw: the pixel's column (its row is (pix - pu) / W)
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
            pu[j] = inb ? (int)u | ((int)v << 16) : 0;     // kRaw: H, W < 2^15 (checked by the entry point)
        }
        float dval[kPPT];
        float t00[kPPT], t01[kPPT], t10[kPPT], t11[kPPT];
        float ta[kPPT], tb[kPPT];
        uint32_t sbits[kPPT], fbits[kPPT];                 // fbits (kLabels): segments in word form
#pragma unroll
        for (int j = 0; j < kPPT; ++j) {
            dval[j] = 0.0f;
            sbits[j] = 0xffffffffu;
            fbits[j] = 0xffffffffu;
            if (kRaw) {
                t00[j] = t01[j] = t10[j] = t11[j] = 0.0f;
                ta[j] = tb[j] = 0.0f;
            }
            if (pix[j] >= 0) {
                if (kRaw) {
                    // the four source texels of the pixel (two of them when the sizes agree): all in flight together
                    const int ux = pu[j] & 0xffff, vy = pu[j] >> 16;
                    const uint32_t *tx = s_taps + 3 * ux, *ty = s_taps + 3 * (W + vy);
                    const uint32_t c0 = tx[0], c1 = tx[1], r0 = ty[0], r1 = ty[1];
                    ta[j] = __uint_as_float(tx[2]);
                    tb[j] = __uint_as_float(ty[2]);
                    t00[j] = sensor_texel(rimg, r0 + c0, raw); t01[j] = sensor_texel(rimg, r0 + c1, raw);
                    t10[j] = sensor_texel(rimg, r1 + c0, raw); t11[j] = sensor_texel(rimg, r1 + c1, raw);
                } else {
                    dval[j] = dimg[pix[j]];
                }
                // segments without any mask pixel were never written by the decoder: consult its bitmap
                if (kLabels) {
                    if (smap) {
                        const uint2 sf = reinterpret_cast<const uint2 *>(smap)[pix[j] >> 12];
                        sbits[j] = sf.x;
                        fbits[j] = sf.y;
                    }
                } else if (smap) {
                    sbits[j] = smap[pix[j] >> 12];
                }
            }
        }
        if (kRaw) {
#pragma unroll
            for (int j = 0; j < kPPT; ++j)
                if (pix[j] >= 0)
                    dval[j] = bilinear_depth(t00[j], t01[j], t10[j], t11[j], ta[j], tb[j]);
        }
        bool vis[kPPT];
        WordT wv[kPPT];
        if (kLabels) {
            // segment by segment the decoder wrote either a palette block (4-bit index per pixel + the segment's distinct
            // words, one 128-byte line) or the mask words (rle_to_maskbits_kernel<.., true>).  A point gathers its index
            // byte or its word -- both kinds in flight together --, then the palette entry out of the line the index
            // came from.
            // The palette entry's address depends on the index byte: a second, dependent round trip.  The first 16 bytes of
            // the palette (entries 0-3 of 32 bits, 0-1 of 64) are therefore fetched WITH the index byte -- same 128-byte
            // line, no extra line -- and only a pixel of a later piece pays the dependent load.
            // (32-bit words only: with 64-bit words two entries cover too few pixels -- config 4 ran 1.55 -> 1.69 ms with them)
            constexpr int kSpec = sizeof(WordT) == 4 ? 4 : 0;
            uint32_t lb[kPPT];
            uint4 first4[kPPT];
            bool pal_go[kPPT];
#pragma unroll
            for (int j = 0; j < kPPT; ++j) {
                vis[j] = (pix[j] >= 0) && (dval[j] != 0.0f) && (fabs(cz[j] - (double)dval[j]) < thresh);
                const int sb = (pix[j] >> 7) & 31;
                const bool go = vis[j] && limg && ((sbits[j] >> sb) & 1);
                const bool words = (fbits[j] >> sb) & 1;
                lb[j] = 0;
                wv[j] = 0;
                first4[j] = make_uint4(0, 0, 0, 0);
                pal_go[j] = go && !words;
                if (go && !words) {
                    lb[j] = limg[(pix[j] & ~127) + ((pix[j] & 127) >> 1)];
                    if (kSpec) first4[j] = *reinterpret_cast<const uint4 *>(limg + (pix[j] & ~127) + 64);
                }
                if (go && words) wv[j] = mimg[pix[j]];
            }
#pragma unroll
            for (int j = 0; j < kPPT; ++j) {
                const uint32_t idx = (lb[j] >> (4 * (pix[j] & 1))) & 15u;
                if (pal_go[j]) {
                    if (idx < (uint32_t)kSpec) {
                        wv[j] = (WordT)(idx == 0 ? first4[j].x : idx == 1 ? first4[j].y : idx == 2 ? first4[j].z : first4[j].w);
                    } else {
                        wv[j] = *reinterpret_cast<const WordT *>(limg + (pix[j] & ~127) + 64 + sizeof(WordT) * idx);
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kPPT; ++j) {
                vis[j] = (pix[j] >= 0) && (dval[j] != 0.0f) && (fabs(cz[j] - (double)dval[j]) < thresh);
                wv[j] = 0;
                if (vis[j] && mimg && ((sbits[j] >> ((pix[j] >> 7) & 31)) & 1)) wv[j] = mimg[pix[j]];
            }
        }
        WordT present = 0;
#pragma unroll
        for (int j = 0; j < kPPT; ++j) {
            if (count_viewed) vcount[j] += vis[j] ? 1 : 0;
            mcount[j] += (sizeof(WordT) == 8) ? __popcll((uint64_t)wv[j]) : __popc((uint32_t)wv[j]);
            present |= wv[j];
        }
#ifdef BFF_DIAG_NO_ROW_STORES
        if (false) {
#else
        if (has_masks) {                                   // wave-uniform
#endif
            // the wave's 32-B sector (kPPT words) of each of the frame's nm rows: lane -> (row lane/4 + 16 i,
            // word lane%4).  Most waves see no mask at all in a given frame: the caller zeroed the rows.
            const int64_t rb = frame_rowbase[f];
            const int wd = lane & (kPPT - 1);
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
                // only the rows that received a bit are stored (their 32-B sector), and flagged in the rows'
                // chunk occupancy masks so that later passes visit nothing else
                const int chunk = (int)(word0 / kCW);
                for (int b0 = 0; b0 < nm; b0 += kWave / kPPT) {        // wave-uniform trip count
                    const int b = b0 + lane / kPPT;
                    const uint64_t v = b < nm ? stage[b][wd] : 0;
                    const uint64_t nz = __ballot(v != 0);
                    if ((nz >> (lane & ~(kPPT - 1))) & ((1u << kPPT) - 1)) {
                        if (word0 + wd < nw) rows[(rb + b) * nw + word0 + wd] = v;     // nw need not be a multiple of 4
                        if (chunk_mask && wd == 0)
                            atomicOr((unsigned long long *)(chunk_mask + (rb + b) * mw + (chunk >> 6)), 1ull << (chunk & 63));
                    }
                }
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

// Measurement aid: which 128-byte lines of the depth images and of the mask-word images does one sweep touch?
// One thread per (frame, point) recomputes the pixel with the sweep's own arithmetic and marks bit (pixel / ppl) of the
// frame's bitmap (ppl = pixels per 128-B line: 32 for float depth and 32-bit mask words, 16 for 64-bit words; with
// label_lines the segment bitmap is bff_rle_to_labels' and segments in label form mark that bitmap, 128 pixels per line); the
// caller counts the bits.  The COMPULSORY HBM traffic of the sweep is 128 B per marked line (each line has to come
// in at least once; everything beyond that is re-fetching), plus the cloud once per frame tile and the counters.
template <typename WordT>
__global__ void sweep_lines_kernel(const double *__restrict__ xyz, int64_t n_points, int64_t n_pad,
                                   const double *__restrict__ inv_pose, Intrinsics K, int n_frames,
                                   const void *__restrict__ depth, RawDepth raw, bool is_raw,
                                   const int32_t *__restrict__ depth_index, int H, int W,
                                   double thresh, const uint32_t *__restrict__ segmap, int64_t seg_words,
                                   const int32_t *__restrict__ frame_mask, uint32_t *__restrict__ depth_lines,
                                   uint32_t *__restrict__ mask_lines, int64_t line_words, uint32_t *__restrict__ label_lines)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (n >= n_points) return;
    const double px = xyz[n], py = xyz[n_pad + n], pz = xyz[2 * n_pad + n];
    const double *P = inv_pose + 16 * (int64_t)f;
    const double cx = fma(P[3], 1.0, fma(P[2], pz, fma(P[1], py, fma(P[0], px, 0.0))));
    const double cy = fma(P[7], 1.0, fma(P[6], pz, fma(P[5], py, fma(P[4], px, 0.0))));
    const double cz = fma(P[11], 1.0, fma(P[10], pz, fma(P[9], py, fma(P[8], px, 0.0))));
    const double p0 = fma(K.k[2], cz, fma(K.k[1], cy, fma(K.k[0], cx, 0.0)));
    const double p1 = fma(K.k[5], cz, fma(K.k[4], cy, fma(K.k[3], cx, 0.0)));
    const double u = rint(p0 / cz), v = rint(p1 / cz);
    if (!((u >= 0.0) && (u < (double)W) && (v >= 0.0) && (v < (double)H))) return;
    const int pix = (int)v * W + (int)u;
    float d;
    if (is_raw) {                                                       // source texels: 64 (uint16) or 32 (float32) per 128-B line
        const char *img = reinterpret_cast<const char *>(depth) + (int64_t)depth_index[f] * raw.frame_stride * (raw.texel_f32 ? 4 : 2);
        auto mark = [&](uint32_t t) {
            const int l = (int)(t >> (raw.texel_f32 ? 5 : 6));
            atomicOr(depth_lines + (int64_t)f * line_words + (l >> 5), 1u << (l & 31));
            return sensor_texel(img, t, raw);
        };
        const uint32_t *tx = raw.taps + 3 * (int)u, *ty = raw.taps + 3 * (W + (int)v);
        const float s00 = mark(ty[0] + tx[0]), s01 = mark(ty[0] + tx[1]), s10 = mark(ty[1] + tx[0]), s11 = mark(ty[1] + tx[1]);
        d = bilinear_depth(s00, s01, s10, s11, __uint_as_float(tx[2]), __uint_as_float(ty[2]));
    } else {
        const int dl = pix >> 5;                                        // 32 floats per 128-B line
        atomicOr(depth_lines + (int64_t)f * line_words + (dl >> 5), 1u << (dl & 31));
        d = reinterpret_cast<const float *>(depth)[(int64_t)depth_index[f] * H * W + pix];
    }
    const int mi = frame_mask ? frame_mask[f] : -1;
    if (mi < 0 || d == 0.0f || !(fabs(cz - (double)d) < thresh)) return;
    if (label_lines) {
        const uint32_t *sm = segmap + 2 * ((int64_t)mi * seg_words + (pix >> 12));
        if (!((sm[0] >> ((pix >> 7) & 31)) & 1)) return;
        if (!((sm[1] >> ((pix >> 7) & 31)) & 1)) {                      // label form: 128 pixels per line
            const int ll = pix >> 7;
            atomicOr(label_lines + (int64_t)f * line_words + (ll >> 5), 1u << (ll & 31));
            return;
        }
    } else if (segmap && !((segmap[(int64_t)mi * seg_words + (pix >> 12)] >> ((pix >> 7) & 31)) & 1)) return;
    const int ml = pix / (int)(128 / sizeof(WordT));
    atomicOr(mask_lines + (int64_t)f * line_words + (ml >> 5), 1u << (ml & 31));
}

// Bounding boxes of the sweep's point tiles (one wave of project_views_kernel = kPPT words = 256 consecutive points
// of the spatially sorted cloud): bounds[t] = (xmin, ymin, zmin, xmax, ymax, zmax).  NaN coordinates are ignored
// (such a point is never in bounds), an empty tile gives (+inf, -inf) and is never culled.
__global__ __launch_bounds__(kBlock) void point_tile_bounds_kernel(const double *__restrict__ xyz, int64_t n_points,
                                                                    int64_t n_pad, int64_t n_tiles,
                                                                    double *__restrict__ bounds)
{
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int j = 0; j < kPPT; ++j) {
        const int64_t n = (t * kPPT + j) * kWave + lane;
        if (n < n_points)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double v = xyz[a * n_pad + n];
                lo[a] = fmin(lo[a], v);
                hi[a] = fmax(hi[a], v);
            }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            lo[a] = fmin(lo[a], __shfl_xor(lo[a], d));
            hi[a] = fmax(hi[a], __shfl_xor(hi[a], d));
        }
    if (lane < 3) bounds[6 * t + lane] = lane == 0 ? lo[0] : lane == 1 ? lo[1] : lo[2];
    else if (lane < 6) bounds[6 * t + lane] = lane == 3 ? hi[0] : lane == 4 ? hi[1] : hi[2];
}

// ---------------------------------------------------------------------------------------------
// 2-D RLE -> mask words, wave-synchronous.  Every wave owns a band of kWaveChunks x 1024 pixels of one
// mask-view and a private 1024-word LDS slice; lane b < n_masks keeps mask b's run cursor (current + next
// run in registers).  Per 1024-pixel chunk the runs enter LDS as XOR toggles at their clipped start and end; then
// every lane takes 16 CONSECUTIVE pixels: an XOR prefix over its toggles, one wave scan of the lanes' totals (DPP), and
// the lane holds its 16 mask words.  There is no block barrier: waves never wait for each other, LDS accesses of one
// wave complete in issue order.  Segments of 128 pixels (8 lanes) without any mask pixel are not written when a
// segment bitmap is requested.  HBM traffic = at most one write of the image.
//
// The kernel is bound by its VALU instructions (a wave64 instruction occupies a SIMD for four cycles): with 4 pixels
// per lane and four wave scans per chunk it spent ~22 instructions per pixel (0.21 ms at config 2); 16 pixels per
// lane need one wave scan and one segmented scan per chunk, and the piece numbers of 8 pixels are one 32-bit SWAR
// prefix sum.
constexpr int kWaveChunk = 1024;             // pixels per wave chunk
constexpr int kPx = kWaveChunk / kWave;      // consecutive pixels per lane; a 128-pixel segment is 8 lanes
constexpr int kWaveChunks = 16384 / kWaveChunk;   // chunks per wave band (16384 pixels)

// LDS position of chunk word i.  A lane moves its 16 consecutive words with four 16-byte accesses 64 bytes apart: left
// in place, the lanes of an access group (16 lanes for ds_read_b128 over 64 banks, 8 contiguous lanes for ds_write_b128
// over 32) would meet on the same banks four at a time.  XOR-ing slot bits (i >> 2) with lane bits -- slot bits 0-1
// with lane bits 1-2, slot bit 3 with lane bit 3 -- spreads every group over all banks; the four words of a 16-byte
// slot stay together, so the coalesced word-form read-out (lane l: words 4l..4l+3 of a 256-word block) still moves slots.
__device__ __forceinline__ int chunk_word_slot(int i) { return i ^ ((((i >> 5) & 3) | ((i >> 4) & 8)) << 2); }

// Inclusive XOR prefix over the 64 lanes of a wave in six DPP steps (row shifts 1, 2, 4, 8 inside each 16-lane
// row, then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2-3): register-to-register, no LDS
// round trips (a __shfl_up ladder is six dependent ds_bpermute latencies).
__device__ __forceinline__ uint32_t wave_xor_scan(uint32_t v)
{
#define BFF_DPP_XOR(ctrl, rows) v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xF, false)
    BFF_DPP_XOR(0x111, 0xF);    // row_shr:1
    BFF_DPP_XOR(0x112, 0xF);    // row_shr:2
    BFF_DPP_XOR(0x114, 0xF);    // row_shr:4
    BFF_DPP_XOR(0x118, 0xF);    // row_shr:8
    BFF_DPP_XOR(0x142, 0xA);    // row_bcast:15 -> rows 1 and 3
    BFF_DPP_XOR(0x143, 0xC);    // row_bcast:31 -> rows 2 and 3
#undef BFF_DPP_XOR
    return v;
}

__device__ __forceinline__ uint64_t wave_xor_scan(uint64_t v)
{
    return (uint64_t)wave_xor_scan((uint32_t)v) | ((uint64_t)wave_xor_scan((uint32_t)(v >> 32)) << 32);
}

// 1 if the word is not zero, else 0 -- as one v_min_u32 (the compiler turns min(x, 1) into compare + select through VCC,
// two instructions and a wait state per pixel in the decoder's flag loop)
__device__ __forceinline__ uint32_t nonzero_flag(uint32_t x)
{
    uint32_t r;
    asm("v_min_u32 %0, 1, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t nonzero_flag(uint64_t x) { return nonzero_flag((uint32_t)x | (uint32_t)(x >> 32)); }

// inclusive prefix sums of eight nibbles (each sum < 16), three shift-adds (left to itself the compiler multiplies by
// 0x11111111: a quarter-rate instruction)
__device__ __forceinline__ uint32_t nibble_prefix_sums(uint32_t p)
{
    asm("v_lshl_add_u32 %0, %0, 4, %0\n\tv_lshl_add_u32 %0, %0, 8, %0\n\tv_lshl_add_u32 %0, %0, 16, %0" : "+v"(p));
    return p;
}

// bit s of the result = some bit of byte s of m is set (m is wave-uniform: scalar arithmetic)
__device__ __forceinline__ uint32_t any_bit_per_byte(uint64_t m)
{
    m |= m >> 4; m |= m >> 2; m |= m >> 1;
    m &= 0x0101010101010101ull;
    return (uint32_t)((m * 0x0102040810204080ull) >> 56);        // bit 8 s -> bit 56 + s; no two products share a position
}

// kLabels: every 128-pixel segment is written in ONE of two forms, chosen here segment by segment:
//   palette form  128 bytes in `labels` (the plane has one such block per segment): 64 bytes of 4-bit indices, one per
//                 pixel, then the palette -- the words of the segment's PIECES (maximal runs of pixels with the same
//                 word) in order, 16 of 32 bits or 8 of 64 bits -- so that pixel p's word is palette[index(p)].  Mask
//                 words change only where a run of some mask starts or ends: a 128-pixel stretch of a row rarely has
//                 more than a handful of pieces, whether masks overlap or not;
//   word form     the mask words in `maskbits`, as without labels, when the segment has more pieces than that.
// The segment bitmap then has TWO uint32 per 4096 pixels: [2k] = segment holds a mask pixel, [2k + 1] = segment is in
// word form.  The sweep is bound by the number of 128-byte lines it fetches: a palette segment is ONE line instead of
// four (32-bit words) or eight.
template <typename WordT, bool kLabels>
__global__ __launch_bounds__(kBlock) void rle_to_maskbits_kernel(
    const int32_t *__restrict__ run_start, const int32_t *__restrict__ run_end,
    const int32_t *__restrict__ mask_run_offs, const int32_t *__restrict__ view_mask_offs,
    int64_t n_pixels, WordT *__restrict__ maskbits, uint32_t *__restrict__ segmap, int64_t seg_words,
    uint8_t *__restrict__ labels, int64_t label_stride)
{
    __shared__ __attribute__((aligned(32))) WordT lds[kBlock / kWave][kWaveChunk];
    using V4 = WordT __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: band and chunk bounds stay in SGPRs
    WordT *bits = lds[wave];
    const int v = blockIdx.y;
    const int g0 = view_mask_offs[v];
    const int nm = view_mask_offs[v + 1] - g0;
    const int64_t band64 = ((int64_t)blockIdx.x * (kBlock / kWave) + wave) * kWaveChunk * kWaveChunks;
    if (band64 >= n_pixels) return;                  // wave-uniform
    const int band0 = (int)band64, npx = (int)n_pixels;               // n_pixels < 2^31 (checked by the entry point)
    WordT *img = maskbits + (int64_t)v * n_pixels;

    constexpr int kNone = INT32_MAX;                 // > every pixel index
    int cur = 0, hi = 0;
    int rs = kNone, re = kNone, ns = kNone, ne = kNone;
    if (lane < nm) {
        int lo = mask_run_offs[g0 + lane];
        hi = mask_run_offs[g0 + lane + 1];
        int r = hi;                                   // first run with end > band0
        while (lo < r) {
            const int mid = (lo + r) >> 1;
            if (run_end[mid] > band0) r = mid; else lo = mid + 1;
        }
        cur = r;
        if (cur < hi) { rs = run_start[cur]; re = run_end[cur]; }
        if (cur + 1 < hi) { ns = run_start[cur + 1]; ne = run_end[cur + 1]; }
    }
    // this lane's words [16 l, 16 l + 16) of the chunk: four 4-word slots (chunk_word_slot moves whole slots)
    int own[kPx / 4];
#pragma unroll
    for (int j = 0; j < kPx / 4; ++j) own[j] = chunk_word_slot(lane * kPx + 4 * j);
#pragma unroll
    for (int j = 0; j < kPx / 4; ++j) *reinterpret_cast<V4 *>(bits + own[j]) = V4{0, 0, 0, 0};
    lds_phase_fence();
    const WordT bit = (WordT)1 << (lane & (int)(sizeof(WordT) * 8 - 1));
    uint32_t *smap = segmap ? segmap + (int64_t)v * seg_words * (kLabels ? 2 : 1) : nullptr;
    const int swz = own[0] ^ (lane * kPx);
    const int sl = lane & 7, seg_of_lane = lane >> 3;                 // lane within its segment, segment within the chunk
    const int m1 = sl >= 1 ? -1 : 0, m2 = sl >= 2 ? -1 : 0;           // lanes a segmented scan step may add into
    constexpr int kPalMax = 64 / (int)sizeof(WordT);                   // 16 palette entries of 32 bits / 8 of 64 bits
    for (int c = 0; c < kWaveChunks; ++c) {
        const int c0 = band0 + c * kWaveChunk;        // scalar
        if (c0 >= npx) break;
        const bool whole = npx - c0 >= kWaveChunk;    // every chunk but the image's last one
        const int c1 = whole ? c0 + kWaveChunk : npx;
        // no mask has a run that reaches into this chunk (idle lanes hold kNone): all its words are zero, and
        // with a segment bitmap zero segments are not written at all
        if (smap && !__ballot(rs < c1)) continue;
        if (lane < nm) {
            while (rs < c1) {
                atomicXor(&bits[chunk_word_slot(max(rs, c0) - c0)], bit);
                if (re < c1) atomicXor(&bits[chunk_word_slot(re - c0)], bit);
                if (re > c1) break;                   // run continues into the next chunk
                ++cur;
                rs = ns; re = ne;
                if (cur + 1 < hi) { ns = run_start[cur + 1]; ne = run_end[cur + 1]; } else { ns = ne = kNone; }
            }
        }
        lds_phase_fence();                            // toggles of all lanes are in LDS
        WordT x[kPx];
#pragma unroll
        for (int j = 0; j < kPx / 4; ++j) {
            const V4 t = *reinterpret_cast<const V4 *>(bits + own[j]);
            x[4 * j] = t[0]; x[4 * j + 1] = t[1]; x[4 * j + 2] = t[2]; x[4 * j + 3] = t[3];
        }
        // a pixel whose toggle word is not zero starts a new piece (its word differs from its left neighbour's):
        // one flag per pixel, spread to the low bit of a nibble
        uint32_t f_lo = 0, f_hi = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f_lo |= nonzero_flag(x[k]) << (4 * k);
            f_hi |= nonzero_flag(x[8 + k]) << (4 * k);
        }
        // XOR of the lane's 16 toggles as a tree of three-input XORs (v_bitop3), one wave scan of the totals, then ONE chain
        // of 16 XORs from the coverage at the lane's first pixel
        WordT tot = 0;
#pragma unroll
        for (int k = 0; k < kPx; k += 4) tot ^= (x[k] ^ x[k + 1]) ^ (x[k + 2] ^ x[k + 3]);
        WordT carry = wave_xor_scan(tot) ^ tot;       // coverage just before this lane's first pixel
        asm volatile("" : "+v"(carry));
        x[0] ^= carry;
#pragma unroll
        for (int k = 1; k < kPx; ++k) x[k] ^= x[k - 1];
        // lanes whose 16 words are not all zero: a piece starts after the lane's first pixel (two different words), or
        // all 16 equal the first one and that is not zero
        const uint64_t nz = __ballot(((f_lo & ~1u) | f_hi) != 0 || x[0] != 0);
        const uint32_t occ8 = smap ? any_bit_per_byte(nz) : 0xFFu;   // scalar: segments of the chunk that are stored
        if (!occ8) continue;                          // (only reachable with a bitmap: nothing was toggled, LDS is still zero)
        // the words go back to LDS: the palette below picks single words out of them, the word form re-reads them coalesced
#pragma unroll
        for (int j = 0; j < kPx / 4; ++j)
            *reinterpret_cast<V4 *>(bits + own[j]) = V4{x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]};
        uint32_t wf8 = occ8;                          // scalar: segments written as words
        if (kLabels) {
            // Palette of the segment (= 8 lanes x 16 pixels): the words are piecewise constant along the row, so the palette
            // simply lists the PIECES in order (repeats allowed, the empty word included) and a pixel keeps the number of
            // its piece = the number of piece starts up to it, the segment's first pixel opening piece 0 whatever its toggle.
            if (sl == 0) f_lo &= ~1u;
            const uint32_t p_lo = nibble_prefix_sums(f_lo), p_hi = nibble_prefix_sums(f_hi);     // <= 8 each: no carries
            const int c_lo = (int)(p_lo >> 28), cnt = c_lo + (int)(p_hi >> 28);
            int incl = cnt;                           // inclusive prefix over the 8 lanes of the segment
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false) & m1;     // row_shr:1
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false) & m2;     // row_shr:2
            incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xA, false);          // row_shr:4 into lanes 4-7, 12-15
            const int base = incl - cnt;              // pieces started before this lane's pixels (piece 0 is open)
            const int pieces = __shfl(incl, lane | 7) + 1;
            const bool overflow = pieces > kPalMax;
            const uint32_t ovf8 = any_bit_per_byte(__ballot(overflow));
            wf8 = occ8 & ovf8;
            if (((occ8 & ~ovf8) >> seg_of_lane) & 1) {
                const int seg0 = seg_of_lane * 128;                    // first pixel of this lane's segment in the chunk
                if (c0 + seg0 < c1) {                                  // the plane is padded to whole segments
                    // piece numbers are < 16 here: adding the same base to every nibble cannot carry
                    uint32_t b_lo = (uint32_t)base, b_hi = (uint32_t)(base + c_lo);
                    b_lo |= b_lo << 4; b_lo |= b_lo << 8; b_lo |= b_lo << 16;
                    b_hi |= b_hi << 4; b_hi |= b_hi << 8; b_hi |= b_hi << 16;
                    uint8_t *seg = labels + (int64_t)v * label_stride + c0 + seg0;
                    *reinterpret_cast<uint2 *>(seg + 8 * sl) = make_uint2(p_lo + b_lo, p_hi + b_hi);
                    WordT *pal = reinterpret_cast<WordT *>(seg + 64);
                    if (sl == 0) pal[0] = x[0];
                    // the pixels that start a piece write its word (a lane rarely has more than one or two)
                    uint64_t rem = (uint64_t)f_lo | ((uint64_t)f_hi << 32);
                    int id = base;
                    lds_phase_fence();                                 // this lane's words are in LDS
                    while (rem) {                                      // two per round: one LDS latency for both
                        const int k0 = (__ffsll((unsigned long long)rem) - 1) >> 2;
                        rem &= rem - 1;
                        const bool two = rem != 0;
                        const int k1 = two ? (__ffsll((unsigned long long)rem) - 1) >> 2 : k0;
                        rem &= rem - 1;                                // 0 stays 0
                        const WordT w0 = bits[(lane * kPx + k0) ^ swz];         // chunk_word_slot: the XOR pattern is the lane's
                        const WordT w1 = bits[(lane * kPx + k1) ^ swz];
                        pal[id + 1] = w0;
                        if (two) pal[id + 2] = w1;
                        id += 2;
                    }
                }
            }
        }
        if (wf8) {
            // word form, read back coalesced: lane l takes words 4l..4l+3 of each 256-word block (= 2 segments)
            lds_phase_fence();
            WordT *img_c = img + c0;
#pragma unroll
            for (int h = 0; h < kWaveChunk / 256; ++h) {
                if (!((wf8 >> (2 * h + (lane >> 5))) & 1)) continue;
                const int q = h * 256 + lane * 4;
                const V4 t = *reinterpret_cast<const V4 *>(bits + chunk_word_slot(q));
                if (whole || c0 + q + 4 <= c1) {
                    *reinterpret_cast<V4 *>(img_c + q) = t;            // chunk starts are multiples of 1024 words
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c0 + q + k < c1) img_c[q + k] = t[k];
                }
            }
        }
        lds_phase_fence();                            // every read of the words is done
#pragma unroll
        for (int j = 0; j < kPx / 4; ++j) *reinterpret_cast<V4 *>(bits + own[j]) = V4{0, 0, 0, 0};   // ready for the next chunk
        lds_phase_fence();                            // ordered before the next chunk's toggles
        // a chunk's 8 segments are 8 consecutive bits of one bitmap word (chunks start at multiples of 1024 pixels)
        if (smap && lane == 0) {
            if (kLabels) {
                atomicOr(smap + 2 * (c0 >> 12), occ8 << ((c0 >> 7) & 31));
                if (wf8) atomicOr(smap + 2 * (c0 >> 12) + 1, wf8 << ((c0 >> 7) & 31));
            } else {
                atomicOr(smap + (c0 >> 12), occ8 << ((c0 >> 7) & 31));
            }
        }
    }
}

}  // namespace bff

using namespace bff;

// Profiling aid (bench.py): events attached to the next sweep dispatch of this host thread.
static thread_local hipEvent_t g_sweep_start = nullptr, g_sweep_stop = nullptr;

extern "C" int bff_profile_next_sweep(void *start_event, void *stop_event)
{
    g_sweep_start = reinterpret_cast<hipEvent_t>(start_event);
    g_sweep_stop = reinterpret_cast<hipEvent_t>(stop_event);
    return BFF_OK;
}

extern "C" void *bff_event_create(void)
{
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}

extern "C" int bff_event_destroy(void *event)
{
    hipError_t e = hipEventDestroy(reinterpret_cast<hipEvent_t>(event));
    return e == hipSuccess ? BFF_OK : fail((int)e, "bff_event_destroy: %s", hipGetErrorString(e));
}

extern "C" int bff_event_record(void *event, void *stream)
{
    hipError_t e = hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream));
    return e == hipSuccess ? BFF_OK : fail((int)e, "bff_event_record: %s", hipGetErrorString(e));
}

extern "C" int bff_event_synchronize(void *event)
{
    hipError_t e = hipEventSynchronize(reinterpret_cast<hipEvent_t>(event));
    return e == hipSuccess ? BFF_OK : fail((int)e, "bff_event_synchronize: %s", hipGetErrorString(e));
}

extern "C" int bff_event_elapsed_ms(void *start_event, void *stop_event, float *ms)
{
    BFF_REQUIRE(start_event && stop_event && ms, "bff_event_elapsed_ms: null pointer");
    hipError_t e = hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop_event));
    if (e == hipSuccess)
        e = hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start_event), reinterpret_cast<hipEvent_t>(stop_event));
    return e == hipSuccess ? BFF_OK : fail((int)e, "bff_event_elapsed_ms: %s", hipGetErrorString(e));
}

extern "C" int64_t bff_label_plane_stride(int64_t n_pixels) { return ceil_div(n_pixels, 128) * 128; }

static int rle_decode(const int32_t *run_start, const int32_t *run_end, const int32_t *mask_run_offs,
                      const int32_t *view_mask_offs, int32_t n_views, int64_t n_pixels,
                      int32_t word_bits, void *maskbits, uint32_t *segmap, uint8_t *labels, void *stream);

extern "C" int bff_rle_to_maskbits(const int32_t *run_start, const int32_t *run_end, const int32_t *mask_run_offs,
                                   const int32_t *view_mask_offs, int32_t n_views, int64_t n_pixels,
                                   int32_t word_bits, void *maskbits, uint32_t *segmap, void *stream)
{
    return rle_decode(run_start, run_end, mask_run_offs, view_mask_offs, n_views, n_pixels, word_bits, maskbits, segmap,
                      nullptr, stream);
}

extern "C" int bff_rle_to_labels(const int32_t *run_start, const int32_t *run_end, const int32_t *mask_run_offs,
                                 const int32_t *view_mask_offs, int32_t n_views, int64_t n_pixels,
                                 int32_t word_bits, uint8_t *labels, void *words, uint32_t *segmap, void *stream)
{
    BFF_REQUIRE(labels && (words || n_views == 0), "bff_rle_to_labels: null pointer");
    return rle_decode(run_start, run_end, mask_run_offs, view_mask_offs, n_views, n_pixels, word_bits, words, segmap,
                      labels, stream);
}

static int rle_decode(const int32_t *run_start, const int32_t *run_end, const int32_t *mask_run_offs,
                      const int32_t *view_mask_offs, int32_t n_views, int64_t n_pixels,
                      int32_t word_bits, void *maskbits, uint32_t *segmap, uint8_t *labels, void *stream)
{
    BFF_REQUIRE(n_views >= 0 && n_pixels > 0, "bff_rle_to_maskbits: bad sizes");
    BFF_REQUIRE(word_bits == 32 || word_bits == 64, "bff_rle_to_maskbits: word_bits must be 32 or 64");
    BFF_LIMIT(n_pixels < (1ll << 31), "bff_rle_to_maskbits: image larger than 2^31 pixels");
    if (n_views == 0) return BFF_OK;
    BFF_REQUIRE(mask_run_offs && view_mask_offs && maskbits, "bff_rle_to_maskbits: null pointer");   // run arrays may be empty (NULL)
    dim3 grid((unsigned)ceil_div(n_pixels, (int64_t)kWaveChunk * kWaveChunks * (kBlock / kWave)), (unsigned)n_views);
    const int64_t seg_words = ceil_div(ceil_div(n_pixels, 128), 32);
    BFF_REQUIRE(!labels || segmap, "bff_rle_to_labels: the label plane needs its segment bitmap");
    if (segmap) {
        hipError_t e = zero_async(segmap, sizeof(uint32_t) * (size_t)n_views * seg_words * (labels ? 2 : 1), as_stream(stream));
        if (e != hipSuccess) return fail((int)e, "bff_rle_to_maskbits: memset: %s", hipGetErrorString(e));
    }
    const int64_t ls = bff_label_plane_stride(n_pixels);
    hipStream_t st = as_stream(stream);
    if (word_bits == 32 && !labels)
        rle_to_maskbits_kernel<uint32_t, false><<<grid, kBlock, 0, st>>>(
            run_start, run_end, mask_run_offs, view_mask_offs, n_pixels, (uint32_t *)maskbits, segmap, seg_words, nullptr, 0);
    else if (word_bits == 32)
        rle_to_maskbits_kernel<uint32_t, true><<<grid, kBlock, 0, st>>>(
            run_start, run_end, mask_run_offs, view_mask_offs, n_pixels, (uint32_t *)maskbits, segmap, seg_words, labels, ls);
    else if (!labels)
        rle_to_maskbits_kernel<uint64_t, false><<<grid, kBlock, 0, st>>>(
            run_start, run_end, mask_run_offs, view_mask_offs, n_pixels, (uint64_t *)maskbits, segmap, seg_words, nullptr, 0);
    else
        rle_to_maskbits_kernel<uint64_t, true><<<grid, kBlock, 0, st>>>(
            run_start, run_end, mask_run_offs, view_mask_offs, n_pixels, (uint64_t *)maskbits, segmap, seg_words, labels, ls);
    return launched("bff_rle_to_maskbits");
}

namespace bff {
// uint16 frames [n][hs][ws] -> 8 x 8-texel tiles, [n][ceil(hs/8)][ceil(ws/8)][8][8] (padding texels 0): the 64 points of a
// wave project onto a compact patch of a frame, and the four taps of a point are neighbours in BOTH directions -- in
// tiles they touch a fraction of the 128-byte lines that row-major frames make them touch.  out_f32: the tiles hold
// float32 metres, `astype(float32) / 1000` of P:432-435 done here once per texel instead of four times per (point, frame).
template <typename OutT>
__global__ void depth_tile_kernel(const uint16_t *__restrict__ src, int hs, int ws, int th, int tw, OutT *__restrict__ dst, float scale)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // one destination texel
    const int64_t per_frame = (int64_t)th * tw * 64;
    if (t >= per_frame) return;
    const int f = blockIdx.y;
    const int tile = (int)(t >> 6), in = (int)(t & 63);
    const int y = (tile / tw) * 8 + (in >> 3), x = (tile % tw) * 8 + (in & 7);
    const uint16_t v = (y < hs && x < ws) ? src[((int64_t)f * hs + y) * ws + x] : (uint16_t)0;
    if constexpr (sizeof(OutT) == 4) dst[(int64_t)f * per_frame + t] = __fdiv_rn((float)v, scale);
    else dst[(int64_t)f * per_frame + t] = v;
}
}  // namespace bff

extern "C" int64_t bff_depth_tiled_texels(int32_t h_src, int32_t w_src)
{
    return (int64_t)((h_src + 7) / 8) * ((w_src + 7) / 8) * 64;
}

extern "C" int bff_depth_tile_u16(const uint16_t *src, int32_t n_frames, int32_t h_src, int32_t w_src, void *dst,
                                  int32_t out_f32, void *stream)
{
    BFF_REQUIRE(n_frames >= 0 && h_src > 0 && w_src > 0, "bff_depth_tile_u16: bad sizes");
    if (n_frames == 0) return BFF_OK;
    BFF_REQUIRE(src && dst, "bff_depth_tile_u16: null pointer");
    BFF_LIMIT(n_frames <= 65535, "bff_depth_tile_u16: too many frames");
    const int th = (h_src + 7) / 8, tw = (w_src + 7) / 8;
    dim3 grid((unsigned)ceil_div((int64_t)th * tw * 64, 256), (unsigned)n_frames);
    if (out_f32)
        depth_tile_kernel<float><<<grid, 256, 0, as_stream(stream)>>>(src, h_src, w_src, th, tw, (float *)dst, 1000.0f);
    else
        depth_tile_kernel<uint16_t><<<grid, 256, 0, as_stream(stream)>>>(src, h_src, w_src, th, tw, (uint16_t *)dst, 1000.0f);
    return launched("bff_depth_tile_u16");
}

#include <map>
#include <mutex>
#include <tuple>

// The resize's tap table for one combination of sizes and layout lives on the device for the rest of the process (a few
// tens of KB each, a handful of combinations): built on first use, synchronously, so that every later call on any stream
// finds it complete.
static int sensor_taps(int hs, int ws, int H, int W, int tiled, hipStream_t st, const uint32_t **out)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int, int>, uint32_t *> cache;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail((int)e, "bff_project_views_u16: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_tuple(dev, hs, ws, H, W, tiled);
    auto it = cache.find(key);
    if (it == cache.end()) {
        uint32_t *table = nullptr;
        e = hipMalloc(reinterpret_cast<void **>(&table), sizeof(uint32_t) * (3 * (size_t)(W + H) + 4));     // read as whole uint4
        if (e != hipSuccess) return fail((int)e, "bff_project_views_u16: tap table: %s", hipGetErrorString(e));
        const double sx = 1.0 / ((double)W / (double)ws), sy = 1.0 / ((double)H / (double)hs);     // io._axis_taps: 1 / (n_dst / n_src)
        sensor_taps_kernel<<<(unsigned)ceil_div(W + H, 256), 256, 0, st>>>(hs, ws, H, W, tiled, sx, sy, table);
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) { (void)hipFree(table); return fail((int)e, "bff_project_views_u16: tap table: %s", hipGetErrorString(e)); }
        it = cache.emplace(key, table).first;
    }
    *out = it->second;
    return BFF_OK;
}

// layout: 0 uint16 row-major frames as stored, 1 uint16 in 8 x 8 tiles, 2 float32 metres in 8 x 8 tiles
static int raw_depth_params(int32_t depth_h, int32_t depth_w, int32_t height, int32_t width, int32_t layout, hipStream_t st,
                            RawDepth *r)
{
    BFF_REQUIRE(layout >= 0 && layout <= 2 && depth_h > 0 && depth_w > 0, "bff_project_views_u16: bad depth layout / size");
    const int tiled = layout != 0;
    r->n_taps = width + height;
    r->texel_f32 = layout == 2;
    r->frame_stride = tiled ? bff_depth_tiled_texels(depth_h, depth_w) : (int64_t)depth_h * depth_w;
    r->scale = 1000.0f;                                                 // depth_scale, hard-coded at P:346
    BFF_LIMIT(r->frame_stride < (1ll << 31), "bff_project_views_u16: depth frame too large");
    return sensor_taps(depth_h, depth_w, height, width, tiled, st, &r->taps);
}

static int project_views_launch(const double *xyz, int64_t n_points, int64_t n_pad,
                                const double *inv_pose, const double *cam_intr_host, int32_t n_frames,
                                const void *depth, const RawDepth *raw, const int32_t *depth_index, int32_t height, int32_t width,
                                double depth_thresh,
                                const void *maskbits, const uint8_t *labels, const uint32_t *segmap, int32_t word_bits,
                                const int32_t *frame_mask, const int32_t *frame_rowbase, const int32_t *frame_nmask,
                                const int32_t *frame_flags,
                                uint64_t *rows, int64_t n_rows, int64_t nw, uint64_t *chunk_mask,
                                int32_t *masked_count, int32_t *viewed_count, const double *tile_bounds,
                                void *stream)
{
    BFF_REQUIRE(n_points >= 0 && n_pad >= n_points && n_frames >= 0, "bff_project_views: bad sizes");
    BFF_REQUIRE(height > 0 && width > 0, "bff_project_views: bad image size");
    BFF_LIMIT((int64_t)height * width < (1ll << 31), "bff_project_views: image larger than 2^31 pixels");
    if (n_points == 0 || n_frames == 0) return BFF_OK;
    BFF_REQUIRE(xyz && inv_pose && cam_intr_host && depth && depth_index && frame_flags, "bff_project_views: null pointer");
    BFF_REQUIRE(nw == ceil_div(n_points, 64), "bff_project_views: nw must be ceil(n_points/64)");
    size_t taps_bytes = 0;
    if (raw) {
        BFF_LIMIT(height < (1 << 15) && width < (1 << 16), "bff_project_views_u16: image too large");
        taps_bytes = sizeof(uint32_t) * ((3 * (size_t)raw->n_taps + 3) / 4 * 4);
        BFF_LIMIT(taps_bytes <= 48 * 1024, "bff_project_views_u16: the resize's tap table (12 B per image row and column) "
                  "exceeds 48 KB of LDS: resize in a separate pass (bff_depth_from_u16)");
    }
    if (maskbits) {
        BFF_REQUIRE(word_bits == 32 || word_bits == 64, "bff_project_views: word_bits must be 32 or 64");
        BFF_REQUIRE(frame_mask && frame_rowbase && frame_nmask && rows && n_rows >= 0, "bff_project_views: mask frames need row outputs");
    }
    const int mw = (int)ceil_div(ceil_div(nw, kCW), 64);
    Intrinsics K;
    for (int i = 0; i < 9; ++i) K.k[i] = cam_intr_host[i];
    const int64_t gx = ceil_div(n_points, kPtsPerBlock);
    int fpb = (int)((int64_t)n_frames * gx / 4096);      // keep >= ~4096 blocks in flight
    fpb = fpb < 1 ? 1 : (fpb > 8 ? 8 : fpb);
    dim3 grid((unsigned)gx, (unsigned)ceil_div(n_frames, fpb));
    const int64_t seg_words = ceil_div(ceil_div((int64_t)height * width, 128), 32);
    const int64_t label_stride = bff_label_plane_stride((int64_t)height * width);
    BFF_REQUIRE(!labels || (maskbits && segmap), "bff_project_views: a label plane comes with its word plane and segment bitmap");
    const hipEvent_t ev0 = g_sweep_start, ev1 = g_sweep_stop;   // attached to the dispatch itself when set
    g_sweep_start = g_sweep_stop = nullptr;
    const RawDepth rd = raw ? *raw : RawDepth{};
#define BFF_SWEEP(WORD, LAB, RAW)                                                                                        \
    hipExtLaunchKernelGGL((project_views_kernel<WORD, LAB, RAW>), grid, dim3(kBlock), (unsigned)taps_bytes, as_stream(stream), \
        ev0, ev1, 0,                                                                                                     \
        xyz, n_points, n_pad, inv_pose, K, n_frames, fpb, depth, rd, depth_index, height, width, depth_thresh,           \
        (const WORD *)maskbits, labels, label_stride, segmap, seg_words, frame_mask, frame_rowbase, frame_nmask,        \
        frame_flags, rows, nw, chunk_mask, mw, masked_count, viewed_count, tile_bounds)
#define BFF_SWEEP_LR(WORD)                                                                                               \
    do { if (labels) { if (raw) BFF_SWEEP(WORD, true, true); else BFF_SWEEP(WORD, true, false); }                        \
         else { if (raw) BFF_SWEEP(WORD, false, true); else BFF_SWEEP(WORD, false, false); } } while (0)
    if (!maskbits || word_bits == 32) BFF_SWEEP_LR(uint32_t); else BFF_SWEEP_LR(uint64_t);
#undef BFF_SWEEP_LR
#undef BFF_SWEEP
    return launched("bff_project_views");
}

extern "C" int bff_project_views(const double *xyz, int64_t n_points, int64_t n_pad,
                                 const double *inv_pose, const double *cam_intr_host, int32_t n_frames,
                                 const float *depth, const int32_t *depth_index, int32_t height, int32_t width,
                                 double depth_thresh,
                                 const void *maskbits, const uint8_t *labels, const uint32_t *segmap, int32_t word_bits,
                                 const int32_t *frame_mask, const int32_t *frame_rowbase, const int32_t *frame_nmask,
                                 const int32_t *frame_flags,
                                 uint64_t *rows, int64_t n_rows, int64_t nw, uint64_t *chunk_mask,
                                 int32_t *masked_count, int32_t *viewed_count, const double *tile_bounds,
                                 void *stream)
{
    return project_views_launch(xyz, n_points, n_pad, inv_pose, cam_intr_host, n_frames, depth, nullptr, depth_index, height,
                                width, depth_thresh, maskbits, labels, segmap, word_bits, frame_mask, frame_rowbase,
                                frame_nmask, frame_flags, rows, n_rows, nw, chunk_mask, masked_count, viewed_count,
                                tile_bounds, stream);
}

extern "C" int bff_project_views_u16(const double *xyz, int64_t n_points, int64_t n_pad,
                                     const double *inv_pose, const double *cam_intr_host, int32_t n_frames,
                                     const void *depth_raw, int32_t depth_h, int32_t depth_w, int32_t depth_layout,
                                     const int32_t *depth_index, int32_t height, int32_t width, double depth_thresh,
                                     const void *maskbits, const uint8_t *labels, const uint32_t *segmap, int32_t word_bits,
                                     const int32_t *frame_mask, const int32_t *frame_rowbase, const int32_t *frame_nmask,
                                     const int32_t *frame_flags,
                                     uint64_t *rows, int64_t n_rows, int64_t nw, uint64_t *chunk_mask,
                                     int32_t *masked_count, int32_t *viewed_count, const double *tile_bounds,
                                     void *stream)
{
    RawDepth raw;
    if (n_points == 0 || n_frames == 0) return BFF_OK;
    BFF_REQUIRE(height > 0 && width > 0, "bff_project_views_u16: bad image size");
    const int rc = raw_depth_params(depth_h, depth_w, height, width, depth_layout, as_stream(stream), &raw);
    if (rc != BFF_OK) return rc;
    return project_views_launch(xyz, n_points, n_pad, inv_pose, cam_intr_host, n_frames, depth_raw, &raw, depth_index, height,
                                width, depth_thresh, maskbits, labels, segmap, word_bits, frame_mask, frame_rowbase,
                                frame_nmask, frame_flags, rows, n_rows, nw, chunk_mask, masked_count, viewed_count,
                                tile_bounds, stream);
}

extern "C" int bff_point_tile_bounds(const double *xyz, int64_t n_points, int64_t n_pad, double *bounds, void *stream)
{
    BFF_REQUIRE(n_points >= 0 && n_pad >= n_points, "bff_point_tile_bounds: bad sizes");
    if (n_points == 0) return BFF_OK;
    BFF_REQUIRE(xyz && bounds, "bff_point_tile_bounds: null pointer");
    const int64_t n_tiles = ceil_div(n_points, (int64_t)kPPT * kWave);
    point_tile_bounds_kernel<<<(unsigned)ceil_div(n_tiles, kBlock / kWave), kBlock, 0, as_stream(stream)>>>(
        xyz, n_points, n_pad, n_tiles, bounds);
    return launched("bff_point_tile_bounds");
}

extern "C" int bff_point_tile_size(void) { return kPPT * kWave; }

// Marks the 128-byte lines one sweep touches (see sweep_lines_kernel).  depth_lines / mask_lines: uint32
// [n_frames][line_words] bitmaps, zeroed by the caller, line_words >= ceil(ceil(H*W / 16) / 32).
static int diag_sweep_lines(const double *xyz, int64_t n_points, int64_t n_pad, const double *inv_pose,
                            const double *cam_intr_host, int32_t n_frames, const void *depth, const RawDepth *raw,
                            const int32_t *depth_index, int32_t height, int32_t width, double depth_thresh,
                            const uint32_t *segmap, int32_t word_bits, const int32_t *frame_mask,
                            uint32_t *depth_lines, uint32_t *mask_lines, int64_t line_words,
                            uint32_t *label_lines, void *stream)
{
    BFF_REQUIRE(xyz && inv_pose && cam_intr_host && depth && depth_index && depth_lines && mask_lines && n_points > 0 &&
                n_frames > 0 && (word_bits == 32 || word_bits == 64), "bff_diag_sweep_lines: bad arguments");
    BFF_REQUIRE(!label_lines || segmap, "bff_diag_sweep_lines: label lines need the label segment bitmap");
    BFF_LIMIT(n_frames <= 65535, "bff_diag_sweep_lines: too many frames");
    Intrinsics K;
    for (int i = 0; i < 9; ++i) K.k[i] = cam_intr_host[i];
    const int64_t seg_words = ceil_div(ceil_div((int64_t)height * width, 128), 32);
    dim3 grid((unsigned)ceil_div(n_points, 256), (unsigned)n_frames);
    const RawDepth rd = raw ? *raw : RawDepth{};
    if (word_bits == 32)
        sweep_lines_kernel<uint32_t><<<grid, 256, 0, as_stream(stream)>>>(xyz, n_points, n_pad, inv_pose, K, n_frames, depth, rd,
            raw != nullptr, depth_index, height, width, depth_thresh, segmap, seg_words, frame_mask, depth_lines, mask_lines,
            line_words, label_lines);
    else
        sweep_lines_kernel<uint64_t><<<grid, 256, 0, as_stream(stream)>>>(xyz, n_points, n_pad, inv_pose, K, n_frames, depth, rd,
            raw != nullptr, depth_index, height, width, depth_thresh, segmap, seg_words, frame_mask, depth_lines, mask_lines,
            line_words, label_lines);
    return launched("bff_diag_sweep_lines");
}

extern "C" int bff_diag_sweep_lines(const double *xyz, int64_t n_points, int64_t n_pad, const double *inv_pose,
                                    const double *cam_intr_host, int32_t n_frames, const float *depth,
                                    const int32_t *depth_index, int32_t height, int32_t width, double depth_thresh,
                                    const uint32_t *segmap, int32_t word_bits, const int32_t *frame_mask,
                                    uint32_t *depth_lines, uint32_t *mask_lines, int64_t line_words,
                                    uint32_t *label_lines, void *stream)
{
    return diag_sweep_lines(xyz, n_points, n_pad, inv_pose, cam_intr_host, n_frames, depth, nullptr, depth_index, height, width,
                            depth_thresh, segmap, word_bits, frame_mask, depth_lines, mask_lines, line_words, label_lines, stream);
}

extern "C" int bff_diag_sweep_lines_u16(const double *xyz, int64_t n_points, int64_t n_pad, const double *inv_pose,
                                        const double *cam_intr_host, int32_t n_frames, const void *depth_raw,
                                        int32_t depth_h, int32_t depth_w, int32_t depth_layout,
                                        const int32_t *depth_index, int32_t height, int32_t width, double depth_thresh,
                                        const uint32_t *segmap, int32_t word_bits, const int32_t *frame_mask,
                                        uint32_t *depth_lines, uint32_t *mask_lines, int64_t line_words,
                                        uint32_t *label_lines, void *stream)
{
    RawDepth raw;
    BFF_REQUIRE(height > 0 && width > 0, "bff_diag_sweep_lines_u16: bad image size");
    const int rc = raw_depth_params(depth_h, depth_w, height, width, depth_layout, as_stream(stream), &raw);
    if (rc != BFF_OK) return rc;
    return diag_sweep_lines(xyz, n_points, n_pad, inv_pose, cam_intr_host, n_frames, depth_raw, &raw, depth_index, height, width,
                            depth_thresh, segmap, word_bits, frame_mask, depth_lines, mask_lines, line_words, label_lines, stream);
}
