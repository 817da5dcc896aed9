// Cosine-similarity GEMM on the matrix cores (include/bff_hip.h: a21, a24).
//
// cos[i][j] = <a_i, b_j> / (|a_i| |b_j|) for f16 rows a (na x dim) and b (nb x dim), f32 accumulate.
// One wave owns a 16x16 output tile and walks dim in steps of 32 with v_mfma_f32_16x16x32_f16; both
// operands are row-major along k, so each lane's 8-element fragment is one 16-byte load and no LDS
// staging is needed (the problem is a few GFLOP at most: 200 labels x 768 dims x <= 10^4 features).
// Row norms are accumulated from the same fragments.
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "common.h"

namespace bff {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using float4v = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(256) void cosine_gemm_f16_kernel(const _Float16 *__restrict__ a, int na,
                                                               const _Float16 *__restrict__ b, int nb, int dim,
                                                               float *__restrict__ out, int norm_b)
{
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int tile_j = blockIdx.x * 4 + wave;                 // 4 column tiles per block
    const int i0 = blockIdx.y * 16, j0 = tile_j * 16;
    if (j0 >= nb) return;                                     // wave-uniform
    const int r = lane & 15, kq = lane >> 4;                  // fragment: row/col r, k = 8*kq .. 8*kq+7
    const bool ra = i0 + r < na, rb = j0 + r < nb;
    const _Float16 *a_row = a + (int64_t)(ra ? i0 + r : 0) * dim;
    const _Float16 *b_row = b + (int64_t)(rb ? j0 + r : 0) * dim;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    float sa = 0.f, sb = 0.f;
    const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    // A trip covers 128 k: lane (r, kq) loads the 64 contiguous bytes k0 + 32 kq .. + 31 of its row (the four lanes of
    // a row read two whole 128-B lines) and feeds 8 of them to each of 4 MFMAs.  That permutes k inside the trip --
    // identically for A and B, so every product still meets its partner; only the order of the float32 additions
    // differs.  (With the natural fragment layout a lane reads 16 B of every 64: each line is fetched twice, by two
    // different k-steps, and the L1 is far too small to hold it in between -- the kernel ran at the L2's transaction
    // rate.)  Two trips' loads are issued before their MFMAs.
    constexpr int kT2 = 2;
    for (int k0 = 0; k0 < dim; k0 += 128 * kT2) {
        half8 fa[kT2][4], fb[kT2][4];
#pragma unroll
        for (int t = 0; t < kT2; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 128 * t + 32 * kq + 8 * u;      // dim is a multiple of 32: a fragment is in or out
                fa[t][u] = (ra && k < dim) ? *reinterpret_cast<const half8 *>(a_row + k) : zero;
                fb[t][u] = (rb && k < dim) ? *reinterpret_cast<const half8 *>(b_row + k) : zero;
            }
#pragma unroll
        for (int t = 0; t < kT2; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = (float)fa[t][u][e], y = (float)fb[t][u][e];
                    sa = fmaf(x, x, sa);
                    sb = fmaf(y, y, sb);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[t][u], fb[t][u], acc, 0, 0, 0);
            }
    }
    // the four k-quarters of a row live in lanes r, r+16, r+32, r+48
    sa += __shfl_xor(sa, 16); sa += __shfl_xor(sa, 32);
    sb += __shfl_xor(sb, 16); sb += __shfl_xor(sb, 32);
    const float nbj = norm_b ? sqrtf(sb) : 1.0f;              // column j0 + (lane & 15)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = kq * 4 + q;                           // C/D map: col = lane & 15, row = 4*(lane>>4) + q
        const float nai = sqrtf(__shfl(sa, row));
        const int i = i0 + row, j = j0 + r;
        if (i < na && j < nb) out[(int64_t)i * nb + j] = acc[q] / (nai * nbj);
    }
}


// Row-tile form for wide banks (nb > 64): one block per 16 rows of A, the k range split over its 4 waves, every wave
// against ALL column tiles (13 at a time: 52 accumulator registers).  A is then read exactly once (13.8 MB at 9000 x 768
// instead of once per column tile -- it does not fit one XCD's L2, so those re-reads came from beyond it), the bank
// (0.3 MB) is read by every block out of L2, and the four partial sums of a tile meet in LDS.  The products are the
// same, only the order of the float32 additions differs from the one-wave-per-tile kernel (k quarters summed last).
// acc + the sum of the squares of 8 halves: four v_dot2_f32_f16 (exact products, float32 accumulation)
__device__ __forceinline__ float sumsq8(half8 v, float acc)
{
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const half2v p = {v[e], v[e + 1]};
        acc = __builtin_amdgcn_fdot2(p, p, acc, false);
    }
    return acc;
}

constexpr int kCT = 13;                                        // column tiles per pass: 208 columns (ScanNet200: 198 + 2)
__global__ __launch_bounds__(256) void cosine_gemm_f16_rows_kernel(const _Float16 *__restrict__ a, int na,
                                                                    const _Float16 *__restrict__ b, int nb, int dim,
                                                                    float *__restrict__ out, int norm_b)
{
    __shared__ float4v red[3][kCT][64];                        // partial accumulators of waves 1..3: [wave - 1][tile][lane]
                                                               // (39 KB: three blocks per CU, all 563 of config 5 resident)
    __shared__ float red_sa[4][16], red_sb[4][kCT][16];        // partial squared norms: rows of A, columns per tile
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int i0 = blockIdx.x * 16;
    const int r = lane & 15, kq = lane >> 4;                   // fragment: row/col r, k = 8*kq .. 8*kq+7
    const int steps = dim / 32, per = (steps + 3) / 4;         // k-steps of this wave: [wave * per, min(steps, ...))
    const int s_lo = wave * per, s_hi = min(steps, s_lo + per);
    const bool ra = i0 + r < na;
    const _Float16 *a_row = a + (int64_t)(ra ? i0 + r : 0) * dim + 8 * kq;
    const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const int n_ct = (nb + 15) / 16;
    for (int c0 = 0; c0 < n_ct; c0 += kCT) {                   // block-uniform
        float4v acc[kCT];
        float sbn[kCT];
#pragma unroll
        for (int t = 0; t < kCT; ++t) { acc[t] = float4v{0.f, 0.f, 0.f, 0.f}; sbn[t] = 0.f; }
        float sa = 0.f;
        for (int st = s_lo; st < s_hi; ++st) {
            const int k0 = st * 32;
            const half8 fa = ra ? *reinterpret_cast<const half8 *>(a_row + k0) : zero;
            half8 fb[kCT];
#pragma unroll
            for (int t = 0; t < kCT; ++t) {                    // all of the step's bank fragments in flight together
                const int j = (c0 + t) * 16 + r;
                fb[t] = j < nb ? *reinterpret_cast<const half8 *>(b + (int64_t)j * dim + 8 * kq + k0) : zero;
            }
            sa = sumsq8(fa, sa);
#pragma unroll
            for (int t = 0; t < kCT; ++t) {
                if (norm_b) sbn[t] = sumsq8(fb[t], sbn[t]);    // every block recomputes the bank's norms: keep it to 4 ops
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb[t], acc[t], 0, 0, 0);
            }
        }
        // the four k-quarters of a row / column live in lanes r, r+16, r+32, r+48
        sa += __shfl_xor(sa, 16); sa += __shfl_xor(sa, 32);
#pragma unroll
        for (int t = 0; t < kCT; ++t) {
            sbn[t] += __shfl_xor(sbn[t], 16); sbn[t] += __shfl_xor(sbn[t], 32);
            if (wave) red[wave - 1][t][lane] = acc[t];
            if (lane < 16) red_sb[wave][t][lane] = sbn[t];
        }
        if (lane < 16) red_sa[wave][lane] = sa;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kCT; ++t) {                        // wave 0 adds the other waves' k-quarters to its own and stores
            if (wave != 0 || c0 + t >= n_ct) continue;
            float4v sum = acc[t];
#pragma unroll
            for (int w = 0; w < 3; ++w) { const float4v p = red[w][t][lane]; sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2]; sum[3] += p[3]; }
            const float sbj = red_sb[0][t][r] + red_sb[1][t][r] + red_sb[2][t][r] + red_sb[3][t][r];
            const float nbj = norm_b ? sqrtf(sbj) : 1.0f;      // column (c0 + t) * 16 + (lane & 15)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = kq * 4 + q;                    // C/D map: col = lane & 15, row = 4*(lane>>4) + q
                const float nai = sqrtf(red_sa[0][row] + red_sa[1][row] + red_sa[2][row] + red_sa[3][row]);
                const int i = i0 + row, j = (c0 + t) * 16 + r;
                if (i < na && j < nb) out[(int64_t)i * nb + j] = sum[q] / (nai * nbj);
            }
        }
        __syncthreads();                                       // the next pass reuses the LDS
    }
}


// Many rows against a wide bank (config 5: 9000 x 768 against 200 x 768): one block per 64 rows of A, wave w owns rows
// 16 w .. 16 w + 15 against ALL column tiles (13 per pass) over the whole k range.  The bank's k-slice of a step
// (208 columns x 32 k = 13 KB) is staged ONCE per block in LDS and read by all four waves -- double buffered, the next
// slice travelling global -> registers while the current one is multiplied, one barrier per step --, A is read once
// (each wave prefetches its next fragment), the bank 141 times out of L2 (43 MB) instead of once per 16 rows.
constexpr int kSK = 32;                                       // k per staged slice (64 = two MFMA k-steps per barrier was measured:
                                                              // 28.4 vs 26.5 us -- the time follows the bytes staged per block)
constexpr int kBPitch = kSK + 8;                              // halves per staged column (+16 B: the 16 columns of a tile
                                                              // spread over the banks for ds_read_b128)
__global__ __launch_bounds__(256) void cosine_gemm_f16_tile64_kernel(const _Float16 *__restrict__ a, int na,
                                                                      const _Float16 *__restrict__ b, int nb, int dim,
                                                                      float *__restrict__ out, int norm_b)
{
    __shared__ _Float16 sB[2][kCT * 16][kBPitch];             // 2 x 208 x 80 B = 33 KB
    __shared__ float col_sq[kCT * 16];
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;                  // fragment: row/col r, k = 8*kq .. 8*kq+7
    const int i0 = blockIdx.x * 64 + wave * 16;
    const bool ra = i0 + r < na;
    const _Float16 *a_row = a + (int64_t)(ra ? i0 + r : 0) * dim + 8 * kq;
    const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const int steps = (dim + kSK - 1) / kSK;                  // dim is a multiple of 32: the last slice may be half full
    const int n_ct = (nb + 15) / 16;
    constexpr int kParts = kSK / 8;                           // 16-byte pieces per staged column
    constexpr int kPieces = kCT * 16 * kParts;
    constexpr int kSlots = (kPieces + 255) / 256;
    for (int c0 = 0; c0 < n_ct; c0 += kCT) {                  // block-uniform
        float4v acc[kCT];
#pragma unroll
        for (int t = 0; t < kCT; ++t) acc[t] = float4v{0.f, 0.f, 0.f, 0.f};
        float sa = 0.f, sq[kSlots];
        const _Float16 *src[kSlots];                          // this thread's pieces: fixed (column, part) for all steps
#pragma unroll
        for (int q = 0; q < kSlots; ++q) {
            const int p = tid + 256 * q, col = p / kParts, j = c0 * 16 + col;
            sq[q] = 0.f;
            src[q] = (p < kPieces && j < nb) ? b + (int64_t)j * dim + 8 * (p % kParts) : nullptr;
        }
        half8 piece[kSlots];
        auto fetch = [&](int st) {
#pragma unroll
            for (int q = 0; q < kSlots; ++q) {
                const int p = tid + 256 * q;
                const bool kin = kSK * st + 8 * (p % kParts) < dim;
                piece[q] = (src[q] && kin) ? *reinterpret_cast<const half8 *>(src[q] + kSK * st) : zero;
            }
        };
        auto stage = [&](int buf) {
#pragma unroll
            for (int q = 0; q < kSlots; ++q) {
                const int p = tid + 256 * q;
                if (p < kPieces) *reinterpret_cast<half8 *>(&sB[buf][p / kParts][8 * (p % kParts)]) = piece[q];
                if (norm_b) sq[q] = sumsq8(piece[q], sq[q]);
            }
        };
        auto load_a = [&](int st, half8 (&f)[kSK / 32]) {
#pragma unroll
            for (int u = 0; u < kSK / 32; ++u)
                f[u] = (ra && kSK * st + 32 * u < dim) ? *reinterpret_cast<const half8 *>(a_row + kSK * st + 32 * u) : zero;
        };
        half8 fa[kSK / 32], fa_next[kSK / 32];
        fetch(0);
        load_a(0, fa);
        stage(0);
        __syncthreads();
        int cur = 0;
        for (int st = 0; st < steps; ++st) {
            if (st + 1 < steps) {                              // the next slice and A fragments travel while this one is multiplied
                fetch(st + 1);
                load_a(st + 1, fa_next);
            }
#pragma unroll
            for (int u = 0; u < kSK / 32; ++u) {
                sa = sumsq8(fa[u], sa);
                half8 fb[kCT];                                 // all 13 fragments in flight, then the MFMAs: read two by two
#pragma unroll                                                 // (what the compiler does with one temporary) a k-step pays the
                for (int t = 0; t < kCT; ++t)                  // LDS latency seven times
                    fb[t] = *reinterpret_cast<const half8 *>(&sB[cur][t * 16 + r][32 * u + 8 * kq]);
                __builtin_amdgcn_sched_barrier(0);             // keep the scheduler from sinking the reads between the MFMAs
#pragma unroll
                for (int t = 0; t < kCT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[u], fb[t], acc[t], 0, 0, 0);
            }
            if (st + 1 < steps) stage(cur ^ 1);                // the other buffer: its readers finished before the last barrier
            __syncthreads();
            cur ^= 1;
#pragma unroll
            for (int u = 0; u < kSK / 32; ++u) fa[u] = fa_next[u];
        }
        // squared norms: rows live in lanes r, r+16, r+32, r+48 of their own wave; a column's kParts pieces are
        // neighbouring threads of one slot
        sa += __shfl_xor(sa, 16); sa += __shfl_xor(sa, 32);
#pragma unroll
        for (int q = 0; q < kSlots; ++q) {
            float v = sq[q];
#pragma unroll
            for (int d = 1; d < kParts; d <<= 1) v += __shfl_xor(v, d);
            const int p = tid + 256 * q;
            if (p < kPieces && (p % kParts) == 0) col_sq[p / kParts] = v;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kCT; ++t) {
            if (c0 + t >= n_ct) continue;
            const float nbj = norm_b ? sqrtf(col_sq[t * 16 + r]) : 1.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = kq * 4 + q;                    // C/D map: col = lane & 15, row = 4*(lane>>4) + q
                const float nai = sqrtf(__shfl(sa, row));
                const int i = i0 + row, j = (c0 + t) * 16 + r;
                if (i < na && j < nb) out[(int64_t)i * nb + j] = acc[t][q] / (nai * nbj);
            }
        }
        __syncthreads();                                       // col_sq and the buffers are reused by the next pass
    }
}

// Bank-stationary form for config 5 (many rows, a bank of 65..256 columns, dim = 768 (CLIP ViT-L/14) or 512 (ViT-B/32)): wave t of a block keeps the
// 16-column strip t of the bank IN REGISTERS for the whole k range (dim / 32 fragments of 4 VGPRs: 96 at dim 768) and
// multiplies it with every 16-row tile of A the block owns; the block's tiles (<= 4 x 24 KB) are brought into LDS by all
// its threads in ONE round of loads issued together with the strip's, so the kernel pays one memory latency, not one
// per k-step (the LDS-staged kernel above: 24 steps x ~1 us of exposed load latency on 141 CUs).  A is read from HBM
// once, the bank 256 x out of L2, every MFMA's A fragment comes out of LDS (16 B per lane), its B fragment is already
// there.  One block per CU (13 waves at config 5), tiles dealt round-robin.
// Measured at 9000 x 200 x 768 (kernel with parts switched off, 200 launches back to back): 23.7 us in all = 6.4 us of an
// empty shell (launch of 256 x 832 threads with 99 KB of LDS, staging loop, epilogue arithmetic) + 0.5 us for A (13.8 MB
// from HBM, hidden under the strip's loads) + 7 us for the bank strips (every lane's 24 fragments are 64 B apart: a load
// instruction touches 16 half lines; the same 24 KB read as 1-KB contiguous loads cost 2.5 us -- an LDS transpose of the
// strip would buy ~4 us) + 6.4 us for 72 LDS fragment reads + MFMAs per wave (LDS-read bound: 16 B per lane and MFMA)
// + 3.7 us for the 7.2 MB of results (64-B row segments).
constexpr int kBankTiles = 4;                                 // 16-row tiles of A resident in LDS per round
constexpr int kAPitch = 768 + 8;                              // (also used at dim 512)                              // halves per staged row (+16 B: rows spread over the banks)

template <int kDim, int kStage>
__global__ __launch_bounds__(1024) void cosine_gemm_f16_bank_kernel(const _Float16 *__restrict__ a, int na,
                                                                     const _Float16 *__restrict__ b, int nb,
                                                                     float *__restrict__ out, int norm_b, int n_tiles)
{
    constexpr int dim = kDim;
    extern __shared__ _Float16 sA[];                           // [kBankTiles][16][kAPitch]
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int r = lane & 15, kq = lane >> 4;                  // fragment: row/col r, k = 8*kq .. 8*kq+7
    constexpr int steps = kDim / 32;
    const int j = wave * 16 + r;                               // this lane's bank column
    const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int pieces_per_row = kDim / 8;                  // 16-byte pieces
    constexpr int tile_pieces = 16 * pieces_per_row;
    // a round = up to kBankTiles of the block's tiles (what the LDS holds), fetched kStage pieces per thread at a time
    // (config 5: 13 waves, 3 tiles -> two batches; the registers of a batch are free again once it sits in LDS)
    const int round_tiles = kBankTiles;
    // the strip: all k-steps in flight at once, together with the first batch of A pieces below
    half8 fb[steps];
    const _Float16 *b_row = b + (int64_t)(j < nb ? j : 0) * dim + 8 * kq;
#pragma unroll
    for (int st = 0; st < steps; ++st) fb[st] = j < nb ? *reinterpret_cast<const half8 *>(b_row + 32 * st) : zero;
    for (int t0 = blockIdx.x; t0 < n_tiles; t0 += gridDim.x * round_tiles) {       // block-uniform rounds
        // this round's tiles: t0, t0 + G, ... -> registers (every piece of a thread in flight together; in the first round
        // the bank strip's fragments with them) -> LDS
        int n_here = 0;
        for (int q = 0; q < round_tiles; ++q) n_here += (t0 + q * (int)gridDim.x < n_tiles) ? 1 : 0;
        const int total = n_here * tile_pieces;
        half8 piece[kStage];
        const int reps = (total + kStage * (int)blockDim.x - 1) / (kStage * (int)blockDim.x);
        for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
            for (int q = 0; q < kStage; ++q) {
                const int p = tid + (int)blockDim.x * (q + kStage * rep);
                piece[q] = zero;
                if (p < total) {
                    const int tl = p / tile_pieces, rem = p - tl * tile_pieces;
                    const int row = rem / pieces_per_row, pc = rem - row * pieces_per_row;
                    const int i = (t0 + tl * (int)gridDim.x) * 16 + row;
                    if (i < na) piece[q] = *reinterpret_cast<const half8 *>(a + (int64_t)i * dim + 8 * pc);
                }
            }
#pragma unroll
            for (int q = 0; q < kStage; ++q) {
                const int p = tid + (int)blockDim.x * (q + kStage * rep);
                if (p < total) {
                    const int tl = p / tile_pieces, rem = p - tl * tile_pieces;
                    const int row = rem / pieces_per_row, pc = rem - row * pieces_per_row;
                    *reinterpret_cast<half8 *>(&sA[((size_t)tl * 16 + row) * kAPitch + 8 * pc]) = piece[q];
                }
            }
        }
        __syncthreads();
        float sb = 0.f;
        if (norm_b) {
#pragma unroll
            for (int st = 0; st < steps; ++st) sb = sumsq8(fb[st], sb);
            sb += __shfl_xor(sb, 16); sb += __shfl_xor(sb, 32);
        }
        const float nbj = norm_b ? sqrtf(sb) : 1.0f;
        for (int tl = 0; tl < n_here; ++tl) {
            const _Float16 *arow = &sA[((size_t)tl * 16 + r) * kAPitch + 8 * kq];
            float4v acc = {0.f, 0.f, 0.f, 0.f};                // one chain: the SIMD's other waves fill the MFMA's latency
            float sa = 0.f;
#pragma unroll
            for (int st = 0; st < steps; ++st) {
                // four A fragments in flight at a time: left alone the scheduler hoists all 24 LDS reads (96 registers
                // next to the 96 of the strip) and spills
                if (st % 4 == 0) __builtin_amdgcn_sched_barrier(0);
                const half8 fa = *reinterpret_cast<const half8 *>(arow + 32 * st);
                sa = sumsq8(fa, sa);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb[st], acc, 0, 0, 0);
            }
            sa += __shfl_xor(sa, 16); sa += __shfl_xor(sa, 32);       // the four k-quarters of a row: lanes r, r+16, r+32, r+48
            const int i0 = (t0 + tl * (int)gridDim.x) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = kq * 4 + q;                    // C/D map: col = lane & 15, row = 4*(lane>>4) + q
                const float nai = sqrtf(__shfl(sa, row));
                const int i = i0 + row;
                if (i < na && j < nb) out[(int64_t)i * nb + j] = acc[q] / (nai * nbj);
            }
        }
        __syncthreads();                                       // the next round overwrites the tiles
    }
    (void)n_waves;
}

// Cosine of every (a_i, b_j) pair IN THE DTYPE OF THE EMBEDDINGS, i.e. with the roundings of the reference's
// tensor expression  (e1 @ e2.T) / (e1.norm() * e2.norm().T)  (compute_clip_similarity R:109-114): each of the
// four tensor ops rounds its result to the embedding dtype.  The class threshold of the refinement is an order
// statistic of the SET of these values (R:321-324), so the rounding decides which values tie -- with fp16
// embeddings (CLIP on a GPU) cosines are multiples of 2^-11 and many labels collapse onto one value.
// One wave per pair, float64 accumulation (the sums are then correctly rounded; a BLAS dot in float32 differs
// from that by an ulp at most, which no implementation can pin across machines).  Tiny by construction:
// <= 200 labels x a few queries.
template <typename T>
__global__ __launch_bounds__(256) void cosine_rows_kernel(const T *__restrict__ a, int na, const T *__restrict__ b,
                                                           int nb, int dim, float *__restrict__ out)
{
    const int lane = lane_id();
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= (int64_t)na * nb) return;                     // wave-uniform
    const int i = (int)(pair / nb), j = (int)(pair % nb);
    const T *pa = a + (int64_t)i * dim, *pb = b + (int64_t)j * dim;
    double dot = 0.0, sa = 0.0, sb = 0.0;
    for (int k = lane; k < dim; k += kWave) {
        const double x = (double)(float)pa[k], y = (double)(float)pb[k];
        dot = fma(x, y, dot);
        sa = fma(x, x, sa);
        sb = fma(y, y, sb);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { dot += __shfl_xor(dot, d); sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); }
    if (lane) return;
    if constexpr (sizeof(T) == 2) {
        const _Float16 hd = (_Float16)(float)dot;                              // e1 @ e2.T          -> f16
        const _Float16 h1 = (_Float16)(float)sqrt(sa), h2 = (_Float16)(float)sqrt(sb);   // norms -> f16
        const _Float16 den = (_Float16)((float)h1 * (float)h2);                // norm * norm.T      -> f16
        out[pair] = (float)(_Float16)((float)hd / (float)den);                 // sim / den          -> f16
    } else {
        const float fd = (float)dot, f1 = (float)sqrt(sa), f2 = (float)sqrt(sb);
        out[pair] = __fdiv_rn(fd, __fmul_rn(f1, f2));
    }
}

}  // namespace bff

using namespace bff;

static bool bank_kernel_on()
{
    static const bool on = [] { const char *e = getenv("BFF_GEMM_BANK"); return !e || atoi(e) != 0; }();
    return on;
}

static int launch_cosine_gemm(const void *a, int32_t na, const void *b, int32_t nb, int32_t dim, float *cos, int norm_b,
                              void *stream, const char *what)
{
    BFF_REQUIRE(na >= 0 && nb >= 0 && dim > 0, "%s: bad sizes", what);
    BFF_REQUIRE(dim % 32 == 0, "%s: dim must be a multiple of 32", what);
    if (na == 0 || nb == 0) return BFF_OK;
    BFF_REQUIRE(a && b && cos, "%s: null pointer", what);
    if (nb > 64 && nb <= 256 && na >= 2048 && (dim == 768 || dim == 512) && bank_kernel_on()) {
        // many rows against a bank of <= 256 columns: the bank stays in registers, A streams through LDS once
        const int n_ct = (int)ceil_div(nb, 16), n_tiles = (int)ceil_div(na, 16);
        const size_t lds = sizeof(_Float16) * (size_t)kBankTiles * 16 * kAPitch;
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipSuccess;
            for (const void *fn : {reinterpret_cast<const void *>(cosine_gemm_f16_bank_kernel<768, 2>),
                                   reinterpret_cast<const void *>(cosine_gemm_f16_bank_kernel<768, 4>),
                                   reinterpret_cast<const void *>(cosine_gemm_f16_bank_kernel<512, 4>)})
                if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail((int)e, "%s: LDS attribute: %s", what, hipGetErrorString(e));
            attr_set = true;
        }
        static const int cus = [] {                            // once: the query costs microseconds, the kernel ~10
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
            return n;
        }();
        const int grid = n_tiles < cus ? n_tiles : cus;
        static const int stage = [] { const char *e = getenv("BFF_GEMM_STAGE"); return e ? atoi(e) : 2; }();
        if (dim == 768 && stage == 2)
            cosine_gemm_f16_bank_kernel<768, 2><<<grid, 64 * n_ct, lds, as_stream(stream)>>>((const _Float16 *)a, na, (const _Float16 *)b,
                                                                                          nb, cos, norm_b, n_tiles);
        else if (dim == 768)
            cosine_gemm_f16_bank_kernel<768, 4><<<grid, 64 * n_ct, lds, as_stream(stream)>>>((const _Float16 *)a, na, (const _Float16 *)b,
                                                                                          nb, cos, norm_b, n_tiles);
        else
            cosine_gemm_f16_bank_kernel<512, 4><<<grid, 64 * n_ct, lds, as_stream(stream)>>>((const _Float16 *)a, na, (const _Float16 *)b,
                                                                                          nb, cos, norm_b, n_tiles);
        return launched(what);
    }
    if (nb > 64 && na >= 2048) {                               // many rows, wide bank: the bank's k-slices through LDS
        cosine_gemm_f16_tile64_kernel<<<(unsigned)ceil_div(na, 64), 256, 0, as_stream(stream)>>>(
            (const _Float16 *)a, na, (const _Float16 *)b, nb, dim, cos, norm_b);
        return launched(what);
    }
    if (nb > 64) {                                             // wide bank: A read once, k split over the block's waves
        cosine_gemm_f16_rows_kernel<<<(unsigned)ceil_div(na, 16), 256, 0, as_stream(stream)>>>(
            (const _Float16 *)a, na, (const _Float16 *)b, nb, dim, cos, norm_b);
        return launched(what);
    }
    dim3 grid((unsigned)ceil_div(ceil_div(nb, 16), 4), (unsigned)ceil_div(na, 16));
    cosine_gemm_f16_kernel<<<grid, 256, 0, as_stream(stream)>>>((const _Float16 *)a, na, (const _Float16 *)b, nb, dim, cos, norm_b);
    return launched(what);
}

extern "C" int bff_cosine_gemm_f16(const void *a, int32_t na, const void *b, int32_t nb, int32_t dim, float *cos,
                                   void *stream)
{
    return launch_cosine_gemm(a, na, b, nb, dim, cos, 1, stream, "bff_cosine_gemm_f16");
}

extern "C" int bff_normalized_gemm_f16(const void *a, int32_t na, const void *b, int32_t nb, int32_t dim, float *sim,
                                       void *stream)
{
    return launch_cosine_gemm(a, na, b, nb, dim, sim, 0, stream, "bff_normalized_gemm_f16");
}

extern "C" int bff_cosine_rows(const void *a, int32_t na, const void *b, int32_t nb, int32_t dim, int32_t dtype,
                               float *cos, void *stream)
{
    BFF_REQUIRE(na >= 0 && nb >= 0 && dim > 0, "bff_cosine_rows: bad sizes");
    BFF_REQUIRE(dtype == 0 || dtype == 1, "bff_cosine_rows: dtype 0 (float32) or 1 (float16)");
    if (na == 0 || nb == 0) return BFF_OK;
    BFF_REQUIRE(a && b && cos, "bff_cosine_rows: null pointer");
    const unsigned grid = (unsigned)ceil_div((int64_t)na * nb, 4);
    if (dtype == 1)
        cosine_rows_kernel<_Float16><<<grid, 256, 0, as_stream(stream)>>>((const _Float16 *)a, na, (const _Float16 *)b, nb, dim, cos);
    else
        cosine_rows_kernel<float><<<grid, 256, 0, as_stream(stream)>>>((const float *)a, na, (const float *)b, nb, dim, cos);
    return launched("bff_cosine_rows");
}

namespace bff {
// one wave per class: lanes stride over dim (<= 64 * 32 = 2048), rows of the class one after the other
template <typename T>
__global__ __launch_bounds__(64) void description_means_kernel(const T *__restrict__ desc, const int32_t *__restrict__ offs,
                                                                int dim, T *__restrict__ out)
{
    constexpr int kMax = 32;                                       // dim <= 2048
    const int c = blockIdx.x, lane = threadIdx.x;
    const int lo = offs[c], hi = offs[c + 1];
    float mean[kMax];
#pragma unroll
    for (int q = 0; q < kMax; ++q) mean[q] = 0.f;
    for (int row = lo; row < hi; ++row) {
        const T *p = desc + (int64_t)row * dim;
        float v[kMax], ss = 0.f;
#pragma unroll
        for (int q = 0; q < kMax; ++q) {
            const int k = lane + 64 * q;
            v[q] = k < dim ? (float)p[k] : 0.f;
            ss = fmaf(v[q], v[q], ss);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) ss += __shfl_xor(ss, d);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);        // F.normalize: x / max(||x||, eps)
#pragma unroll
        for (int q = 0; q < kMax; ++q) mean[q] += v[q] * inv;
    }
    const float cnt = (float)(hi - lo);
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < kMax; ++q) { mean[q] = mean[q] / cnt; ss = fmaf(mean[q], mean[q], ss); }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) ss += __shfl_xor(ss, d);
    const float nrm = sqrtf(ss);
#pragma unroll
    for (int q = 0; q < kMax; ++q) {
        const int k = lane + 64 * q;
        if (k < dim) out[(int64_t)c * dim + k] = (T)(mean[q] / nrm);
    }
}
}  // namespace bff

extern "C" int bff_description_means(const void *desc, const int32_t *offs, int32_t n_classes, int32_t dim, int32_t dtype,
                                     void *out, void *stream)
{
    BFF_REQUIRE(n_classes >= 0 && dim > 0 && (dtype == 0 || dtype == 1), "bff_description_means: bad arguments");
    BFF_LIMIT(dim <= 2048, "bff_description_means: dim > 2048");
    if (n_classes == 0) return BFF_OK;
    BFF_REQUIRE(desc && offs && out, "bff_description_means: null pointer");
    if (dtype == 1)
        description_means_kernel<_Float16><<<n_classes, 64, 0, as_stream(stream)>>>((const _Float16 *)desc, offs, dim, (_Float16 *)out);
    else
        description_means_kernel<float><<<n_classes, 64, 0, as_stream(stream)>>>((const float *)desc, offs, dim, (float *)out);
    return launched("bff_description_means");
}
