/* bff_hip.h -- C ABI of libbff_hip.so: the MI355X (gfx950) kernels behind the 2D->3D mask
 * projection + multi-view fusion + refinement hot path of Beyond-Fixed-Forms.
 *
 * The reference has no FFI/operator ABI for this path: its boundary is files + Python dicts
 * (SURVEY.md section 8b).  This header is therefore the seam between this repo's Python host
 * (the beyond_fixed_forms_amd Python modules, which mirror the reference's dict/CLI interface) and its HIP
 * kernels.  Each entry point cites the reference lines whose arithmetic it replaces; paths are
 * relative to the reference checkout (tools/projection_2d_to_3d.py = P, tools/refinement.py = R,
 * tools/utils/rle_encode_decode.py = RLE, tools/segmentation_2d.py = SEG).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all memory
 *     (the host allocates through torch); no entry point allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream);
 *   - return value: 0 = launched, <0 = rejected argument (BFF_E_*), >0 = hipError_t of the launch;
 *     bff_last_error() gives a message for the calling thread;
 *   - "bit rows": a boolean row over N points is stored as ceil(N/64) uint64 words, point n is bit
 *     (n & 63) of word (n >> 6); padding bits of the last word are always 0;  `nw` = ceil(N/64);
 *   - integers are exact, float64 geometry is bit-exact w.r.t. the reference (see bff_project_views).
 */
#ifndef BFF_HIP_H
#define BFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BFF_OK 0
#define BFF_E_ARG (-1)      /* null pointer / negative size / unsupported parameter */
#define BFF_E_LIMIT (-2)    /* size beyond what a kernel supports (documented per call) */

#define BFF_ABI_VERSION 8

int bff_abi_version(void);
const char *bff_last_error(void);
/* gfx target the library was compiled for ("gfx950"). */
const char *bff_arch(void);

/* Host-side helper (plain CPU code, all pointers are HOST pointers, no stream): component ids -> the groups
 * merge_masks keeps (P:203-226) as CSR.  comp[i] in [0, n) names the component of row i (e.g. the output of
 * bff_merge_components); kept: components with >= max(min_members, 1) members except isolated rows without
 * a self loop (the reference's `[]`, counted in *n_void when min_members <= 0); order: by smallest member,
 * members ascending.  Returns K (offs[0..K], members[0..offs[K]), sizes[0..K)); -1 = id out of range. */
int bff_host_component_csr(const int32_t *comp, const uint8_t *has_self_loop, int32_t n, int32_t min_members,
                           int32_t *offs, int32_t *members, int32_t *sizes, int32_t *n_void);

/* ------------------------------------------------------------------------------------------
 * a1 -- 2-D RLE masks -> per-pixel mask words.   Replaces RLE.rle_decode_batch (RLE:35-61) +
 * decode_2d_masks (RLE:82-99) + the float conversion at P:417-421; the dense (M,1,H,W) uint8
 * tensors (11.3 GB per scene at 200k x 300 x 30) are never materialised.
 *
 * Mask-view v owns masks [view_mask_offs[v], view_mask_offs[v+1]) (at most `word_bits` of them);
 * mask g owns runs [mask_run_offs[g], mask_run_offs[g+1]); run r covers flattened row-major pixels
 * [run_start[r], run_end[r]) (0-based, end exclusive, clipped to H*W by the host).  Runs of one
 * mask must be sorted and disjoint (what rle_encode_batch RLE:10-32 emits; the host normalises
 * anything else).  Output: maskbits[v][p] has bit b set iff pixel p lies in mask view_mask_offs[v]+b.
 * word_bits = 32 -> uint32 words, 64 -> uint64 words.
 * segmap (optional): uint32 [n_views][ceil(ceil(n_pixels/128)/32)] bitmap, bit s of a view = "128-pixel
 * segment s contains a mask pixel".  When given, all-zero segments of maskbits are NOT written (their
 * content is undefined) and bff_project_views must be given the same bitmap; NULL = every word is written.
 */
int bff_rle_to_maskbits(const int32_t *run_start, const int32_t *run_end, const int32_t *mask_run_offs,
                        const int32_t *view_mask_offs, int32_t n_views, int64_t n_pixels, int32_t word_bits,
                        void *maskbits, uint32_t *segmap, void *stream);

/* The same decode, each 128-pixel segment written in the cheaper of two forms:
 *   palette form  one 128-byte block of `labels` (uint8, rows of bff_label_plane_stride(n_pixels) bytes; block s of view
 *                 v = bytes [128 s, 128 s + 128) of its row): 64 bytes of 4-bit indices, pixel p's in byte (p mod 128) / 2,
 *                 low nibble for even p; then the palette: the words of the segment's PIECES (maximal runs of pixels
 *                 with the same word, the empty word included) in order -- 16 entries of 32 bits or 8 of 64 bits.  Pixel
 *                 p's word is palette[index(p)].  Index bytes of pixels past the image edge are unspecified;
 *   word form     words[v][p] = the word of bff_rle_to_maskbits (same layout, same word_bits) -- when the segment has
 *                 more pieces than the palette holds.
 * segmap (required): uint32 [n_views][2 * ceil(ceil(n_pixels/128)/32)]: word 2k = "segment holds a mask pixel" for
 * segments 32k..32k+31 (others are not written at all), word 2k + 1 = "segment is in word form".  Only the plane a
 * segment's form names is written for it.  Mask words change only where a run of some mask starts or ends, so a
 * 128-pixel stretch of a row has a handful of pieces whether masks overlap or not: the decoder -- bound by its writes --
 * stores, and the sweep gathers from, ONE 128-byte line per segment instead of four (word_bits 32) or eight.
 * labels must be 128-byte aligned, segmap 8-byte aligned. */
int64_t bff_label_plane_stride(int64_t n_pixels);
int bff_rle_to_labels(const int32_t *run_start, const int32_t *run_end, const int32_t *mask_run_offs,
                      const int32_t *view_mask_offs, int32_t n_views, int64_t n_pixels, int32_t word_bits,
                      uint8_t *labels, void *words, uint32_t *segmap, void *stream);

/* ------------------------------------------------------------------------------------------
 * a2-a7 (+a15) -- fused per-frame: world->camera transform, projection, rounding, bounds +
 * depth test, mask-word gather, instance bit rows and the two per-point vote counters.
 * Replaces, per frame: P:424-425 (inv(pose) @ cloud), compute_projected_pts_tensor P:37-48,
 * compute_visibility_mask_tensor P:51-70 (depth_thresh is passed by the caller: 0.08 at P:438,565),
 * compute_visible_masked_pts_tensor P:73-92, the scatter-adds P:459-461 and, for the
 * detection-ratio sweep, P:548-567.
 *
 *   xyz          float64 [3][n_pad] structure-of-arrays (x row, y row, z row), n_pad >= n_points
 *   inv_pose     float64 [n_frames][16] row-major inverse camera pose (np.linalg.inv on the host)
 *   cam_intr     float64 [9] row-major K (HOST pointer; copied into kernel arguments)
 *   depth        float32 [n_depth][H*W] metres; frame f uses image depth_index[f]
 *   maskbits     mask words of bff_rle_to_maskbits, or -- with `labels` -- the word plane of bff_rle_to_labels
 *   labels       palette plane of bff_rle_to_labels (then maskbits and segmap are its companions) or NULL
 *   segmap       the decoder's segment bitmap (see bff_rle_to_maskbits) or NULL
 *   frame_mask   int32 [n_frames]: index of the frame's mask-word image in `maskbits`, or -1
 *   frame_rowbase int32 [n_frames]: first instance row of the frame (row = rowbase + bit)
 *   frame_nmask  int32 [n_frames]: number of masks (bits) of the frame, 0..word_bits
 *   frame_flags  int32 [n_frames]: bit0 = add visibility to viewed_count (P:567)
 *   rows         uint64 [n_rows][nw] instance bit rows, ZEROED BY THE CALLER (hipMemsetAsync on the same
 *                stream): row (rowbase+b) receives the points of mask b of every frame with a mask image;
 *                only 32-byte sectors that hold a point are stored                                (a6, a8)
 *   chunk_mask   optional uint64 [n_rows][bff_chunk_mask_words(nw)], zeroed by the caller too: bit c of a row =
 *                "chunk c (words 8c..8c+7, 512 points) of the row holds a point" -- exactly what bff_row_stats
 *                computes, for free here; lets bff_row_stats / bff_or_reduce_groups skip the empty 99 %
 *   masked_count int32 [n_points], += number of masks of the frame containing the visible point
 *                (P:459-461 adds 1 per mask, not per view); may be NULL
 *   viewed_count int32 [n_points], += visibility for frames with flag bit0; may be NULL
 *   tile_bounds  optional float64 table from bff_point_tile_bounds (frustum culling per wave and frame); NULL = off
 *
 * Arithmetic contract (bit-exact with NumPy/OpenBLAS float64 as used by the reference):
 *   c_i = fma chain over k = 0..3 of inv_pose[i][k] * (x, y, z, 1)[k] starting from +0.0;
 *   p_i = fma chain over k = 0..2 of K[i][k] * c[k];  u = rint(p_0 / c_2), v = rint(p_1 / c_2)
 *   (IEEE division, round half to even); in bounds iff 0 <= u < W and 0 <= v < H evaluated on the
 *   doubles (NaN/inf/out-of-int64-range fail, which equals the reference's INT64_MIN cast);
 *   visible iff depth != 0 and fabs(c_2 - (double)depth) < depth_thresh.  No z > 0 test.
 */
int bff_project_views(const double *xyz, int64_t n_points, int64_t n_pad,
                      const double *inv_pose, const double *cam_intr_host, int32_t n_frames,
                      const float *depth, const int32_t *depth_index, int32_t height, int32_t width,
                      double depth_thresh,
                      const void *maskbits, const uint8_t *labels, const uint32_t *segmap, int32_t word_bits,
                      const int32_t *frame_mask, const int32_t *frame_rowbase, const int32_t *frame_nmask,
                      const int32_t *frame_flags,
                      uint64_t *rows, int64_t n_rows, int64_t nw, uint64_t *chunk_mask,
                      int32_t *masked_count, int32_t *viewed_count, const double *tile_bounds, void *stream);

/* The same sweep with the depth frames resident as the PNGs store them (P:431-436): depth_raw uint16
 * [n_depth][depth_h][depth_w] millimetres at the sensor's resolution.  `astype(float32) / 1000` and the bilinear resize
 * to (height, width) are evaluated per point at the pixel it projects to, with exactly the float32 operations of
 * bff_depth_from_u16 (tap coefficients as io._axis_taps: float64 source coordinate cast to float32 before its floor is
 * subtracted, border columns copied, row indices clamped) -- results are bit-identical to bff_depth_from_u16 followed by
 * bff_project_views, the (height, width) float32 images (8 x the bytes) are never built.  height < 2^15, width < 2^16.
 * depth_layout: 0 uint16 frames row-major as stored; 1 uint16 in 8 x 8-texel tiles; 2 float32 METRES in 8 x 8-texel tiles
 * (bff_depth_tile_u16: `astype(float32) / 1000` done once per texel there) -- same values, fewer 128-byte lines per wave
 * (a wave's points project onto a compact patch and a point's four taps are neighbours in both directions) and, for
 * layout 2, no conversion or division left in the sweep.  The tap table of the resize (12 bytes per image row and column:
 * two texel offsets and the fraction, as io._axis_taps) is built on the device once per size combination and staged in
 * LDS by every block of the sweep; it must fit 48 KB (height + width <= 4096). */
int64_t bff_depth_tiled_texels(int32_t h_src, int32_t w_src);      /* texels of one tiled frame (padded to whole tiles) */
/* dst: uint16 [n][tiled texels] (out_f32 == 0) or float32 metres [n][tiled texels] (out_f32 != 0) */
int bff_depth_tile_u16(const uint16_t *src, int32_t n_frames, int32_t h_src, int32_t w_src, void *dst, int32_t out_f32,
                       void *stream);
int bff_project_views_u16(const double *xyz, int64_t n_points, int64_t n_pad,
                          const double *inv_pose, const double *cam_intr_host, int32_t n_frames,
                          const void *depth_raw, int32_t depth_h, int32_t depth_w, int32_t depth_layout,
                          const int32_t *depth_index, int32_t height, int32_t width, double depth_thresh,
                          const void *maskbits, const uint8_t *labels, const uint32_t *segmap, int32_t word_bits,
                          const int32_t *frame_mask, const int32_t *frame_rowbase, const int32_t *frame_nmask,
                          const int32_t *frame_flags,
                          uint64_t *rows, int64_t n_rows, int64_t nw, uint64_t *chunk_mask,
                          int32_t *masked_count, int32_t *viewed_count, const double *tile_bounds, void *stream);

/* Frustum culling for bff_project_views (optional, exact).  bounds: float64 [ceil(n_points / bff_point_tile_size())][6]
 * = (xmin, ymin, zmin, xmax, ymax, zmax) of every tile of bff_point_tile_size() consecutive points -- the points
 * one wave of the sweep owns.  Given the table, a wave skips a frame when the box of its points cannot contain a
 * point whose pixel is in bounds (a conservative half-space test of the 8 corners against the four image
 * borders, both in front of and behind the camera: the reference has no z > 0 test, P:57-67); results are
 * bit-identical with and without it.  It pays when the cloud is spatially sorted (scene.morton_order). */
int bff_point_tile_bounds(const double *xyz, int64_t n_points, int64_t n_pad, double *bounds, void *stream);
int bff_point_tile_size(void);

/* Profiling aid.  bff_profile_next_sweep(start, stop): the next bff_project_views launch of the calling host
 * thread carries the two events on its dispatch (hipExtLaunchKernelGGL), so that bff_event_elapsed_ms(start,
 * stop) is the kernel's own duration, as a profiler reports it (events recorded around a launch add the
 * latency of two barrier packets, ~50 us here).  bff_event_create returns a hipEvent_t (NULL on failure);
 * bff_event_elapsed_ms waits for `stop`. */
int bff_profile_next_sweep(void *start_event, void *stop_event);
void *bff_event_create(void);
int bff_event_destroy(void *event);
int bff_event_elapsed_ms(void *start_event, void *stop_event, float *ms);
/* hipEventRecord / hipEventSynchronize on such an event (hosts that hold a raw stream handle: a torch.cuda.Event.record()
 * looks the current stream up first, which costs more than the record). */
int bff_event_record(void *event, void *stream);
int bff_event_synchronize(void *event);

/* ------------------------------------------------------------------------------------------
 * Bit-row primitives (a8-a13, a16-a20).
 */

/* area[r] = popcount(rows[idx ? idx[r] : r])          (torch.sum(dim=1) at P:161,592,596; R:86-87) */
int bff_popcount_rows(const uint64_t *rows, const int32_t *idx, int32_t n_rows, int64_t nw,
                      int32_t *area, void *stream);

/* inter[i][j] = popcount(a[ia[i]] & b[ib[j]]), int32 [na][nb]: the {0,1} matmuls of
 * calculate_iou_between_stages R:84 and the any-overlap test of solve_overlapping P:289-292.
 * ia / ib may be NULL (identity). */
int bff_cross_popcount(const uint64_t *a, const int32_t *ia, int32_t na,
                       const uint64_t *b, const int32_t *ib, int32_t nb, int64_t nw,
                       int32_t *inter, void *stream);

/* Per-row statistics that make the Gram block-sparse: area[r] = popcount(row r); mean_word[r] = mean word
 * index of its set bits (INT32_MAX for an empty row; a sort key that groups rows covering the same part of
 * the cloud when the points are spatially sorted); chunk_mask[r] = occupancy bits over chunks of 8 words
 * (512 points), bff_chunk_mask_words(nw) uint64 words per row; hist[r] = uint32 [64] histogram of the
 * row's set bits over 64 equal word ranges (bin width ceil(nw/64) words); signature[r] = 30-bit key: the indices
 * of the (at most 6, here: first 5) bins holding >= 15 % of the row, ascending, 6 bits each, most significant
 * first, unused slots = 63 (an empty row is all 63s and sorts last): rows showing the same object
 * get the same key, so sorting by it clusters them into the same 64-row tiles.
 * chunk_mask_given != 0: chunk_mask is an INPUT (as written by bff_project_views) and only the flagged chunks
 * of every row are read; 0: chunk_mask is computed here from a full pass over the rows.
 * chunk_pop (optional, may be NULL): uint16 [n_rows][64 * bff_chunk_mask_words(nw)], points of the row in each of
 * its 512-point chunks (0 where the row has none): the bins of bff_merge_components' second-level bound. */
int bff_row_stats(const uint64_t *rows, int32_t n_rows, int64_t nw, int32_t *area, int32_t *mean_word,
                  uint64_t *chunk_mask, int32_t chunk_mask_given, uint32_t *hist, int64_t *signature,
                  uint16_t *chunk_pop, void *stream);
int bff_chunk_mask_words(int64_t nw);

/* rows[r][chunk c] = 0 for every chunk flagged in chunk_mask (as written by bff_project_views): returns a
 * zero-filled row buffer to all-zero after a scene at the cost of the ~1 % of it that was ever stored, so the
 * buffer can serve the next scene without another full zero-fill. */
int bff_clear_flagged_chunks(uint64_t *rows, int32_t n_rows, int64_t nw, const uint64_t *chunk_mask, void *stream);

/* a9-a11: merge adjacency of `aggregate` P:100-146.  For every pair (i, j):
 *   I = popcount(rows[i] & rows[j]);  iou = (float)I / ((float)area[i] + (float)area[j] - (float)I)
 *   (IEEE float32 division; 0/0 = NaN compares false, P:149-166);
 *   adjacent = label_id[i] == label_id[j]  &&  iou > iou_thres   (float32 compare, P:120-122)
 * label_id replaces the string compare of calculate_feature_similarity P:169-187.
 * `order` (int32 [n_rows], may be NULL = identity) is the order in which rows are tiled: tile t holds rows
 * order[64t .. 64t+63].  adj: uint64 [n_rows][ceil(n_rows/64)], bit q of row p <=> rows order[p] and
 * order[q] are adjacent (i.e. indexed by POSITION in `order`), fully written.
 * chunk_mask (from bff_row_stats) + tile_mask (scratch, uint64 [ceil(n_rows/64)][bff_chunk_mask_words(nw)])
 * enable chunk skipping: a 64x64 tile pair only visits chunks both tiles occupy (exact: skipped words
 * contribute 0 to every intersection); both NULL = visit every word.
 * hist (from bff_row_stats, may be NULL; used only together with chunk_mask and when inter == NULL): a tile
 * pair first bounds every intersection by UB = sum_bins min(hist_i, hist_j) >= I and evaluates the same
 * float32 IoU test on min(UB, area_i, area_j) (the test is monotone in I); tiles without a possible edge
 * skip their word loop.  Sound: the adjacency is unchanged.
 * inter (optional, may be NULL): int32 [n_rows][n_rows] Gram matrix in ROW index space, for tests. */
int bff_merge_adjacency(const uint64_t *rows, int32_t n_rows, int64_t nw, const int32_t *order,
                        const uint64_t *chunk_mask, uint64_t *tile_mask, const uint32_t *hist,
                        const int32_t *area,
                        const int32_t *label_id, float iou_thres,
                        uint64_t *adj, int32_t *inter, void *stream);

/* a9-a12 in one pass, the production path: connected components of the merge graph of `aggregate`
 * (same adjacency definition as bff_merge_adjacency) WITHOUT materialising the adjacency matrix and
 * without the transitive-closure matmuls of find_unconnected_subgraphs_tensor P:250-274.  64x64 tile pairs
 * are visited diagonal-first in `order`; a pair is examined only if the histogram bound allows an edge AND
 * its two rows are not yet in one component; edges found are merged into a disjoint-set forest
 * (`parent`, int32 [n_rows]; init_parent != 0: start from singletons, == 0: continue from the forest
 * already in `parent`, e.g. a coarser partition known to the caller) with compare-and-swap.  Exact: a pair is skipped only when it cannot
 * be an edge or when adding the edge could not change the components.
 * `order` may list only n_order <= n_rows of the rows (a sample): only pairs among the listed rows are
 * examined, which is how the caller builds a coarse forest first (every 8th row) and then refines it with the
 * full order and init_parent = 0.  comp (int32 [n_rows], out, may be NULL): comp[i] = smallest row index of
 * i's component.  Rows with an empty
 * adjacency row (area 0, or thr >= 1) form singleton components here; the host turns them into the
 * reference's empty lists (it knows area and thr).  All other arguments as for bff_merge_adjacency;
 * chunk_mask, tile_mask and hist are required; scratch: uint32 [bff_merge_scratch_words(n_rows)] (sorted histograms,
 * tile bounds, position-indexed row tables, the two tile-pair lists).
 * Inside a tile pair the block keeps disjoint sets of its 128 rows in LDS (started from the global forest): every
 * few chunks it settles the pairs whose PARTIAL intersection already passes the IoU test (the float32 expression is
 * monotone in I, so the edge exists) or whose rows have become connected meanwhile, and stops as soon as no pair is
 * open; only edges that merge two local sets are pushed into `parent`.
 * chunk_pop (optional, from bff_row_stats): second-level bound -- pairs that pass the 64-bin histogram bound are
 * bounded again by sum over the shared 512-point chunks of min(points of i, points of j) before any word is read.
 * diag (optional, NULL in production): int32 [16 + 2 * capacity], zeroed by the caller: += {tile pairs evaluated,
 * chunks visited, candidate pairs, unions, phase clocks ...} (scripts/diag_merge_phases.py). */
int64_t bff_merge_scratch_words(int32_t n_rows);
/* Whether bff_merge_components applies the chunk bound for rows of nw words (clouds of >= ~0.5 M points; the environment
 * variable BFF_CHUNK_BOUND=0/1 overrides): callers can skip computing chunk_pop otherwise. */
int32_t bff_merge_uses_chunk_bound(int64_t nw);
/* Profiling aid: the next tile-pass dispatch of bff_merge_components on this host thread carries the two events
 * (like bff_profile_next_sweep). */
int bff_profile_next_merge(void *start_event, void *stop_event);
int bff_merge_components(const uint64_t *rows, int32_t n_rows, int64_t nw, const int32_t *order, int32_t n_order,
                         const uint64_t *chunk_mask, uint64_t *tile_mask, const uint32_t *hist,
                         uint32_t *scratch, const int32_t *area, const int32_t *label_id, float iou_thres,
                         int32_t *parent, int32_t init_parent, int32_t *comp, int32_t *diag,
                         const uint16_t *chunk_pop, void *stream);

/* rows_out[r] bit o = rows_in[r] bit idx[o], o < n_out (bit gather).  Undoes the spatial point sort the
 * host applies at upload: idx[o] = position of original point o in the sorted cloud. */
int bff_permute_bits(const uint64_t *rows_in, int32_t n_rows, int64_t nw_in, const int32_t *idx,
                     int64_t n_out, int64_t nw_out, uint64_t *rows_out, void *stream);

/* a12: connected components of a symmetric bit adjacency (find_unconnected_subgraphs_tensor
 * P:250-274 computes the transitive closure by n rounds of clamp(R@A + A); for the symmetric
 * matrices produced by bff_merge_adjacency its rows are the connected components).
 * One call = one propagation round: label[i] <- min(label[i], min over neighbours j of label[j]),
 * followed by pointer jumping; *changed (int32, device) is set to 1 if any label moved.
 * The host initialises label[i] = i, zeroes *changed and iterates until it stays 0; the result is
 * label[i] = smallest member index of i's component. */
int bff_components_round(const uint64_t *adj, int32_t n_nodes, const int32_t *label_in, int32_t *label_out,
                         int32_t *changed, void *stream);

/* a13: out[g] = OR of rows[members[group_offs[g] .. group_offs[g+1])]      (merge_masks P:219-224;
 * also the `.any(dim=0)` merge of R:269).  max_group_size >= the largest group (host knows the groups);
 * it only sizes the launch.  conf / conf_mean (optional, both or neither; dtype as bff_group_conf_mean): the
 * same launch also computes bff_group_conf_mean on extra blocks, so the long sequential sums run beside the OR.
 * chunk_mask (optional, the rows' chunk flags): long rows are read through it. */
int bff_or_reduce_groups(const uint64_t *rows, int64_t nw, const int32_t *group_offs, const int32_t *members,
                         int32_t n_groups, int32_t max_group_size, uint64_t *out, const void *conf,
                         int32_t conf_dtype, void *conf_mean, const uint64_t *chunk_mask, void *stream);

/* a13: mean[g] = (((c[m0] + c[m1]) + c[m2]) ...) / len, every step rounded to the confidence dtype
 * (P:225: python `sum(conf) / len(conf)` over 0-dim tensors).  dtype: 0 = float32, 1 = float16. */
int bff_group_conf_mean(const void *conf, int32_t dtype, const int32_t *group_offs, const int32_t *members,
                        int32_t n_groups, void *mean, void *stream);

/* a16 + a14 + the two popcounts around them in one pass, k <= bff_resolve_overlaps_max_rows() (4096) rows:
 *   before[i] = popcount(rows[i])                             (before any edit, P:592)
 *   solve_overlapping P:277-301 on `rows` in place.  The reference visits the pairs (i < j) that overlap before any
 *     edit in (i, j) order; the row merged from fewer raw masks (size[], ties: row i) loses the points of the other.
 *     For a single point that walk is a champion scan over the rows holding it, so the point ends up in exactly one of
 *     them: the one with the largest size, among equals the largest index (derivation: rows.hip,
 *     resolve_priority_kernel) -- an exclusive prefix OR over the rows in that order, no intersections needed;
 *   rows[i] &= keep (P:595; keep may be NULL);  after[i] = popcount(rows[i])  (P:596).
 * The literal ordered replay stays available as bff_overlap_ops + bff_apply_row_ops (tests compare the two). */
int bff_resolve_overlaps(uint64_t *rows, int32_t k, int64_t nw, const int32_t *size, const uint64_t *keep,
                         int32_t *before, int32_t *after, void *stream);
int bff_resolve_overlaps_max_rows(void);

/* a16/a20: sequential row program applied independently to every word column.
 * ops: int32 [n_ops][3] = (opcode, dst, src) executed in order;
 *   opcode 0: rows[dst] &= ~rows[src]   (solve_overlapping P:295-299)
 *   opcode 1: rows[dst] |=  rows[src]   (stage-1 duplicate merge R:248)
 *   opcode 2: rows[dst]  =  rows[src]
 * n_ops < 0: the list lives on the device as [count, triples...] (as written by bff_overlap_ops). */
int bff_apply_row_ops(uint64_t *rows, int64_t nw, const int32_t *ops, int32_t n_ops, void *stream);

/* a16: the pair loop of solve_overlapping P:285-299 without a host round trip.  inter = K x K intersections
 * of the aggregated rows before any edit (bff_cross_popcount), size[i] = number of raw masks merged into row
 * i (P:285).  ops (int32 [1 + 3*K*(K-1)/2]) receives [count, (0, loser, winner)...] in the reference's pair
 * order: the row built from fewer masks loses the overlap, ties: the first row loses. */
int bff_overlap_ops(const int32_t *inter, const int32_t *size, int32_t k, int32_t *ops, void *stream);

/* a16: rows[r] &= keep for r < n_rows                                                   (P:595) */
int bff_and_rows(uint64_t *rows, int32_t n_rows, int64_t nw, const uint64_t *keep, void *stream);

/* out[r] = rows[idx[r]] (row gather; P:601-607 row selection, R:310). */
int bff_gather_rows(const uint64_t *rows, const int32_t *idx, int32_t n_out, int64_t nw, uint64_t *out,
                    void *stream);

/* bit rows <-> dense boolean rows (uint8 0/1, the layout of torch.bool), for the dict contract
 * {"ins": bool (K,N)} (P:630-634, R:411). */
int bff_unpack_rows(const uint64_t *rows, int32_t n_rows, int64_t nw, int64_t n_points, uint8_t *dense,
                    void *stream);
int bff_pack_rows(const uint8_t *dense, int32_t n_rows, int64_t n_points, int64_t nw, uint64_t *rows,
                  void *stream);

/* a18: 1-D RLE (Open3DIS stage-1 "ins") -> bit rows.  rle_decode R:26-39.  Same run layout as
 * bff_rle_to_maskbits: row g owns runs [row_run_offs[g], row_run_offs[g+1]), sorted, disjoint. */
int bff_rle_to_rows(const int32_t *run_start, const int32_t *run_end, const int32_t *row_run_offs,
                    int32_t n_rows, int64_t n_points, int64_t nw, uint64_t *rows, void *stream);

/* SURVEY section 8f row 1 -- bit rows -> 1-D RLE in the reference's format (rle_encode_batch RLE:10-32: 1-based
 * start, length pairs), so results can be stored like Open3DIS stage-1 files (eval_scannet200.py:123-124 reads
 * them).  bff_rle_count_runs: n_runs[r] = number of runs of row r.  The host turns that into exclusive offsets
 * run_offs (int64 [n_rows]) and a total; bff_rle_encode_rows then fills counts (int64 [2 * n_runs_total]) with
 * row r's pairs at counts[2*run_offs[r] ...].  Padding bits of the rows must be zero (they always are here). */
int bff_rle_count_runs(const uint64_t *rows, int32_t n_rows, int64_t nw, int32_t *n_runs, void *stream);
int bff_rle_encode_rows(const uint64_t *rows, int32_t n_rows, int64_t nw, const int64_t *run_offs,
                        int64_t n_runs_total, int64_t *counts, void *stream);

/* SURVEY section 8f row 4 -- the consumer right after the path: ScanNet instance evaluation intersects every
 * predicted mask with every ground-truth instance (`count_nonzero(logical_and(gts == instance_id, pred_mask))`,
 * evaluation/eval/scannetv2_inst_eval.py:334).  bff_ids_to_rows turns the per-point id vector into one bit row
 * per requested value (rows[v] bit p = ids[p] == values[v]); the intersections are then one bff_cross_popcount. */
int bff_ids_to_rows(const int64_t *ids, int64_t n_points, const int64_t *values, int32_t n_values, int64_t nw,
                    uint64_t *rows, void *stream);

/* ------------------------------------------------------------------------------------------
 * a14/a15 -- point filters (P:512-583), kept entirely on the device.
 * bff_point_values: vals[n] = (float)masked[n] / ((float)viewed[n] + 1.0f) (P:571; IEEE float32), or
 * (float)masked[n] when viewed == NULL (P:513).  The caller sorts vals ascending (bff_sort_f32), then
 * bff_select_unique_rank writes *thr = distinct_values[floor(fraction * n_distinct)]
 * -- the reference's `x.unique()[math.floor(t * x.unique().shape[0])]` (P:516-518, 574-576; float64 product;
 * NaN when the index is out of range, where python raises IndexError) -- and *n_unique.  block_scratch:
 * int32 [ceil(n/1024)].
 * bff_ratio_keep: keep bit n = masked[n] > 0 && !(value[n] < thr) with value as above (P:522,578,583);
 * the threshold is read from thr_dev (device) when non-NULL, else the immediate `thr`; use_thr == 0 -> keep =
 * masked > 0. */
int bff_point_values(const int32_t *masked, const int32_t *viewed, int64_t n_points, float *vals, void *stream);
/* The same (thr, n_unique) without sorting n_points values: the statistic is a function of the integer pair (masked,
 * viewed), of which a scene holds only ~10^3..10^5 different ones: every block collects the distinct values of its
 * 1024 points in LDS and writes them to its slice of `scratch` (no global atomics, nothing to clear); 64 blocks, one per
 * 1/64 of the hash space, merge the slices in LDS sets (together = x.unique() of P:516 / P:574); one block radix-selects
 * the rank: three launches.  scratch: uint32 [bff_point_threshold_scratch_words(n_points)]; *overflow (device, not
 * cleared by the call) is set to 1 if a partition holds more distinct values than bff_point_threshold_capacity(): use
 * the sorting path then. */
int64_t bff_point_threshold_scratch_words(int64_t n_points);
int32_t bff_point_threshold_capacity(void);
int32_t bff_point_threshold_capacity_set(int32_t cap);     /* test hook: smaller capacity (0 = default); returns the new one */
int bff_point_threshold_pairs(const int32_t *masked, const int32_t *viewed, int64_t n_points, double fraction,
                              uint32_t *scratch, float *thr, int32_t *n_unique, int32_t *overflow, void *stream);
int bff_select_unique_rank(const float *sorted, int64_t n, double fraction, int32_t *block_scratch,
                           float *thr, int32_t *n_unique, void *stream);
int bff_ratio_keep(const int32_t *masked, const int32_t *viewed, int64_t n_points, float thr,
                   const float *thr_dev, int32_t use_thr, int64_t nw, uint64_t *keep, void *stream);

/* ------------------------------------------------------------------------------------------
 * a4 / SURVEY section 8f row 2 -- depth ingestion on the device.  src: uint16 [n_frames][h_src][w_src] raw
 * depth in millimetres (the decoded 16-bit PNGs, P:432-433); dst: float32 [n_frames][height][width] =
 * resize(src.astype(f32) / depth_scale, (width, height)) with the bilinear rule of cv2.resize (INTER_LINEAR:
 * half-pixel centres, edge clamp, horizontal then vertical 2-tap passes in float32, P:436).  The host passes
 * the tap tables it computed in float64 (x0/x1/ax [width], y0/y1/ay [height]; io.bilinear_taps); with equal
 * source and target size the tables may be NULL and the call is the plain division.  Uploading 2 B/pixel at
 * sensor resolution instead of 4 B/pixel at colour resolution cuts the per-scene H2D volume ~8x. */
int bff_depth_from_u16(const uint16_t *src, int32_t n_frames, int32_t h_src, int32_t w_src,
                       const int32_t *x0, const int32_t *x1, const float *ax,
                       const int32_t *y0, const int32_t *y1, const float *ay,
                       int32_t height, int32_t width, float depth_scale, float *dst, void *stream);

/* ------------------------------------------------------------------------------------------
 * Library sorts (rocPRIM radix sort) so that the ABI is self-sufficient.  Both follow the usual two-call
 * protocol: temp == NULL -> only *temp_bytes is written (no launch); then call again with that much device
 * scratch.  bff_sort_f32: keys_out = keys_in ascending (the input of bff_select_unique_rank).
 * bff_argsort_i64: order_out = stable ascending argsort of int64 keys (ties keep index order), keys_scratch =
 * int64 [n]; key_bits = 64 sorts the full signed keys, key_bits < 64 promises 0 <= key < 2^key_bits and sorts
 * only those bits (fewer radix passes).  Used to order the Gram tiles by row signature (30 bits) and, stably on
 * top, by label id. */
int bff_sort_f32(const float *keys_in, float *keys_out, int64_t n, void *temp, size_t *temp_bytes, void *stream);
int bff_argsort_i64(const int64_t *keys, int64_t *keys_scratch, int32_t *order_out, int32_t n, int32_t key_bits,
                    void *temp, size_t *temp_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * a21/a24 -- cosine similarity GEMM on the matrix cores (MFMA f16 -> f32).
 *   cos[i][j] = <a_i, b_j> / (||a_i|| * ||b_j||), float32 accumulate and normalisation
 * (compute_clip_similarity R:93-115; bbox_filter SEG:388-393).  a: f16 [na][dim], b: f16 [nb][dim],
 * dim % 32 == 0, cos: float32 [na][nb]. */
int bff_cosine_gemm_f16(const void *a, int32_t na, const void *b, int32_t nb, int32_t dim,
                        float *cos, void *stream);
/* a24 -- the box filter's product  F.normalize(box_emb) @ text.T  (SEG:388-393): rows of a are normalised, rows of b
 * are taken as they are (the reference normalises the text mean beforehand, SEG:336).  Same kernels, same limits. */
int bff_normalized_gemm_f16(const void *a, int32_t na, const void *b, int32_t nb, int32_t dim,
                            float *sim, void *stream);
/* a24 -- compute_avg_description_encodings (SEG:324-337): per class c, rows [offs[c], offs[c+1]) of desc (float16 or
 * float32 [n][dim]) are L2-normalised (eps 1e-12 as F.normalize), averaged, and the mean is normalised again;
 * out: same dtype [n_classes][dim]; float32 arithmetic. */
int bff_description_means(const void *desc, const int32_t *offs, int32_t n_classes, int32_t dim, int32_t dtype,
                          void *out, void *stream);

/* a21 -- the per-class text cosines in the dtype of the embeddings: cos[i][j] with every tensor op of
 *   (e1 @ e2.T) / (e1.norm() * e2.norm().T)  (compute_clip_similarity R:109-114) rounded to that dtype, so that
 * the SET of similarities the class threshold is taken from (R:321-324) ties exactly where the reference's
 * does (fp16 CLIP on a GPU: multiples of 2^-11).  dtype 0: a, b float32; dtype 1: a, b float16; cos float32
 * (holding float16 values for dtype 1); float64 accumulation; any dim. */
int bff_cosine_rows(const void *a, int32_t na, const void *b, int32_t nb, int32_t dim, int32_t dtype,
                    float *cos, void *stream);


/* ------------------------------------------------------------------------------------------
 * Groups formed on the device (P:203-226 without the host round trip) and the whole scene in one call.
 */
#define BFF_GROUP_CAP 256         /* groups the device forms by itself (= rows of the fused overlap pass): the default ... */
#define BFF_GROUP_CAP_MAX 512     /* ... and the largest capacity a workspace may ask for (bff_scene_workspace.group_cap) */
#define BFF_SIGNATURE_BITS 30     /* bff_row_stats signatures are 30-bit keys */

/* Device twin of bff_host_component_csr for at most `cap` <= BFF_GROUP_CAP kept groups.  comp[i] = smallest row index
 * of i's component (bff_merge_components) -- an INPUT when parent == NULL; with parent (the disjoint-set forest
 * bff_merge_components leaves behind when called with comp == NULL) comp is an OUTPUT, flattened here on the way.
 * Outputs (device): info[4] = {K kept groups (may exceed cap), flags (bit 0:
 * K > cap, bit 1: min_members <= 0 and empty components exist -- both: use the host path), largest group, number of
 * 32-member slices}; sizes[cap], first[cap] (= smallest member = where the group's label comes from), offs[cap+1],
 * members[n_rows] (ascending inside a group), slices[3 * bff_group_slice_cap(n_rows, cap)] (work items of
 * bff_or_reduce_grouped); count: scratch int32 [n_rows] (count_is_zero != 0: the caller has cleared it). */
int32_t bff_group_slice_cap(int32_t n_rows, int32_t cap);
int bff_group_components(int32_t *comp, int32_t *parent, const int32_t *area, int32_t n_rows, float iou_thres,
                         int32_t min_members, int32_t cap, int32_t *count, int32_t count_is_zero, int32_t *info,
                         int32_t *sizes, int32_t *first, int32_t *offs, int32_t *members, int32_t *slices, void *stream);
/* bff_or_reduce_groups for those groups: out [cap][nw] (zeroed here; rows >= K stay zero), conf_mean [cap] in the
 * confidence dtype.  chunk_mask (optional, the rows' chunk flags): long rows are read through it. */
int bff_or_reduce_grouped(const uint64_t *rows, int64_t nw, int32_t n_rows, const int32_t *info, int32_t cap,
                          const int32_t *offs, const int32_t *members, const int32_t *slices, uint64_t *out,
                          const void *conf, int32_t conf_dtype, void *conf_mean, const uint64_t *chunk_mask, void *stream);
/* bff_resolve_overlaps with the row count on the device (*k_dev <= k_cap, else nothing is touched). */
int bff_resolve_overlaps_dev(uint64_t *rows, int32_t k_cap, int64_t nw, const int32_t *size, const uint64_t *keep,
                             int32_t *before, int32_t *after, const int32_t *k_dev, void *stream);
/* out[r] bit perm[s] = in[r] bit s, set bits only (undoes the spatial point sort by scatter: aggregated rows are
 * sparse); out must be zero; rows >= *k_dev (when given) are skipped. */
int bff_scatter_bits(const uint64_t *rows_in, int32_t n_rows, int64_t nw_in, const int32_t *perm, int64_t n,
                     int64_t nw_out, uint64_t *rows_out, const int32_t *k_dev, void *stream);
/* bff_cross_popcount when only the first *k_dev rows of b's leading `lead` rows (and of a, with limit_a) are
 * non-zero: tiles inside the zero part are skipped; inter is zeroed first. */
int bff_cross_popcount_dev(const uint64_t *a, int32_t na, const uint64_t *b, int32_t nb, int64_t nw, int32_t *inter,
                           const int32_t *k_dev, int32_t limit_a, int32_t lead, void *stream);
/* bff_clear_flagged_chunks unless *veto != 0 (device flag). */
int bff_clear_flagged_chunks_unless(uint64_t *rows, int32_t n_rows, int64_t nw, const uint64_t *chunk_mask,
                                    const int32_t *veto, void *stream);

/* Device-resident inputs of one scene (what scene.prepare_scene uploads; all pointers are device pointers). */
typedef struct bff_scene {
    int64_t n_points, n_pad, nw;
    const double *xyz;              /* [3][n_pad] */
    const double *tile_bounds;      /* bff_point_tile_bounds table or NULL */
    const double *inv_pose;         /* [n_frames][16] */
    double cam_intr[9];
    const float *depth;             /* [n_depth][height * width] metres, or NULL when depth_raw is given */
    const void *depth_raw;          /* sensor-resolution frames in layout `depth_tiled` (bff_project_views_u16), or NULL */
    const int32_t *depth_index, *frame_mask, *frame_rowbase, *frame_nmask, *frame_flags;   /* [n_frames] */
    const int32_t *run_start, *run_end, *mask_run_offs, *view_mask_offs;                     /* 2-D RLE run tables */
    const void *conf;               /* [n_rows] float16 / float32 */
    const int32_t *label_id;        /* [n_rows] */
    const int32_t *unsort;          /* [n_points] position of original point o in the sorted cloud, or NULL */
    const int32_t *perm;            /* [n_points] original index of sorted position s (inverse of unsort), or NULL */
    const int32_t *s1_run_start, *s1_run_end, *s1_row_run_offs;     /* stage-1 run tables or NULL */
    int32_t height, width, n_frames, n_mviews, word_bits, n_rows, conf_f16, n_label_ids, s1_rows, depth_h, depth_w,
            depth_tiled;            /* depth_layout of bff_project_views_u16: 0 uint16 rows, 1 uint16 tiles, 2 float32 tiles */
} bff_scene;

typedef struct bff_scene_params {
    double depth_thresh;            /* 0.08 at P:438,565 */
    double filter_fraction;         /* occurance_threshold / detected_ratio_threshold */
    float iou_thres;
    int32_t min_members;            /* cfg.min_aggragated_masks */
    int32_t filter_mode;            /* 0 none, 1 occurrence (P:512-522), 2 detection ratio (P:524-578) */
    int32_t filter_sort;            /* != 0: threshold by sorting all values (after header word BFF_HDR_OVERFLOW was set) */
} bff_scene_params;

/* Scratch of bff_scene_project, allocated by the caller for the scene's sizes (beyond_fixed_forms_amd/pipeline.py).
 * `rows` must be all zero on entry; it is all zero again when the call's work has run on the fast path.
 * Every buffer the call's steps expect zeroed -- masked, viewed, count, chunk_mask, segmap, hdr, agg, merge_scratch
 * and (clouds that use the chunk bound) chunk_pop -- must lie inside ONE allocation of `zero_bytes` bytes
 * starting at `masked`: the call clears it with a single fill (checked; beyond_fixed_forms_amd/pipeline.py lays it out). */
typedef struct bff_scene_workspace {
    void *maskbits; uint32_t *segmap;  /* word plane and two-word segment bitmap of bff_rle_to_labels */
    uint8_t *labels;                /* palette blocks, [n_mviews][bff_label_plane_stride(H*W)] */
    uint64_t *rows, *chunk_mask, *keep, *tile_mask, *agg, *both;
    int32_t *masked, *viewed, *sel_scratch, *area, *mean_word, *order, *parent, *comp, *count;
    int32_t *gmembers, *goffs, *slices;
    uint32_t *pair_scratch;         /* bff_point_threshold_scratch_words(n_points) */
    float *vals, *vals_sorted;
    uint32_t *hist, *merge_scratch;
    uint16_t *chunk_pop;            /* [n_rows][64 * chunk-mask words] */
    int64_t *sig, *sig_keys, *sig_sorted;
    void *sort_temp; size_t sort_temp_bytes;
    size_t zero_bytes;              /* size of the block that starts at `masked` (see above) */
    int32_t *hdr;                   /* device, bff_scene_header_words(s1_rows, group_cap) int32 */
    int32_t *hdr_host;              /* pinned host mirror of the same size */
    int32_t group_cap;              /* BFF_GROUP_CAP or BFF_GROUP_CAP_MAX: kept groups formed on the device; agg, both and the
                                       header are sized by it.  More groups than that: header flag, the host continues */
    int32_t pad_;
    void *heavy_stream;             /* optional hipStream_t for the chip-filling kernels (decode, sweep, tile pass): callers that keep
                                       several scenes in flight give the SAME stream to a few of them, so that at most as many
                                       chip-filling kernels run side by side as there are heavy streams while the scenes' chains of
                                       small kernels overlap freely.  NULL: everything on the call's stream */
    void *events[4];                /* four hipEvent_t of this workspace for the hand-overs between the two streams (heavy_stream != NULL) */
    void *aux_stream;               /* optional second hipStream_t of this workspace: the point filter's threshold chain (four small
                                       launches that only the overlap resolution at the end waits for) runs there, beside the row
                                       statistics / tile order / components chain.  NULL: everything in one chain */
    void *aux_events[2];            /* two hipEvent_t for the fork after the sweep and the join before the overlap resolution */
} bff_scene_workspace;

/* Header layout (int32 words). */
#define BFF_HDR_K 0                 /* info[4] of bff_group_components: K, flags, largest group, slices */
#define BFF_HDR_NUNIQUE 4           /* distinct filter values (0: the reference would raise IndexError) */
#define BFF_HDR_THR 5               /* float32 threshold */
#define BFF_HDR_OVERFLOW 6          /* != 0: more distinct filter values than bff_point_threshold_pairs holds: run the
                                       scene again with params.filter_sort = 1 (everything after the sweep is void) */
/* cap = the workspace's group_cap (BFF_GROUP_CAP or BFF_GROUP_CAP_MAX) */
#define BFF_HDR_SIZES(cap) 16                         /* [cap] members per group */
#define BFF_HDR_FIRST(cap) (16 + (cap))               /* [cap] smallest member of the group */
#define BFF_HDR_BEFORE(cap) (16 + 2 * (cap))          /* [cap] popcount before overlap resolution (P:592) */
#define BFF_HDR_AFTER(cap) (16 + 3 * (cap))           /* [cap] popcount after overlaps + point filter (P:596) */
#define BFF_HDR_CONF(cap) (16 + 4 * (cap))            /* [cap] confidence means, in the confidence dtype, packed */
#define BFF_HDR_CROSS(cap) (16 + 5 * (cap))           /* [s1_rows][cap + s1_rows] stage-1 x (stage-2 groups | stage-1) */
int32_t bff_scene_header_words(int32_t s1_rows, int32_t cap);
int32_t bff_scene_struct_bytes(int32_t which);     /* 0 bff_scene, 1 bff_scene_params, 2 bff_scene_workspace */

/* The whole device side of one scene on `stream`, ending with an asynchronous copy of the header into
 * ws->hdr_host: nothing in it waits for the host.  After the stream has reached that copy the host reads K,
 * flags, sizes, first members, before/after counts, confidence means and the refinement's intersections from
 * the header; ws->both holds the K aggregated, overlap-resolved, filtered rows in the caller's point order followed
 * by the decoded stage-1 rows.  flags != 0: the group tables are incomplete -- continue from ws->comp / ws->area on
 * the host (projection._projection_back), ws->rows is then still intact. */
int bff_scene_project(const bff_scene *scene, const bff_scene_params *params, const bff_scene_workspace *ws,
                      void *stream);

/* ------------------------------------------------------------------------------------------
 * Ingestion: cloud layout on the device.  pts: float64 [n][stride] exactly as <scene>.npy
 * holds it (P:387: stride 6, xyz first).  Writes the structure-of-arrays cloud the sweep reads (soa [3][n_pad], pad
 * zeroed) and, with sort != 0, orders the points along a 30-bit Morton curve over their bounding box first (same
 * codes and a stable sort as scene.morton_order on the host -> the same permutation): unsort[o] = position of
 * original point o, perm = its inverse.  codes: uint32 [2 n], box: float64 [6], temp: sort scratch (temp == NULL:
 * size query into *temp_bytes). */
int bff_cloud_layout(const double *pts, int64_t n, int64_t stride, int64_t n_pad, int32_t sort, double *soa,
                     int32_t *unsort, int32_t *perm, uint32_t *codes, double *box, void *temp, size_t *temp_bytes,
                     void *stream);

/* Measurement aid: the 128-byte lines of the depth and mask-word images that one sweep touches, as bitmaps (uint32
 * [n_frames][line_words], zeroed by the caller; line index = pixel / (pixels per 128 B)).  128 B per marked line is
 * the sweep's compulsory HBM traffic for these images (bench.py: roofline.compulsory).  label_lines != NULL: segmap is
 * bff_rle_to_labels' two-word bitmap; segments in palette form mark label_lines (128 pixels per line) instead. */
int bff_diag_sweep_lines(const double *xyz, int64_t n_points, int64_t n_pad, const double *inv_pose,
                         const double *cam_intr_host, int32_t n_frames, const float *depth, const int32_t *depth_index,
                         int32_t height, int32_t width, double depth_thresh, const uint32_t *segmap, int32_t word_bits,
                         const int32_t *frame_mask, uint32_t *depth_lines, uint32_t *mask_lines, int64_t line_words,
                         uint32_t *label_lines, void *stream);
/* The same for bff_project_views_u16: depth_lines counts the 128-byte lines of the uint16 source frames (64 texels
 * per line; line_words >= ceil(ceil(depth_h * depth_w / 64) / 32) as well). */
int bff_diag_sweep_lines_u16(const double *xyz, int64_t n_points, int64_t n_pad, const double *inv_pose,
                             const double *cam_intr_host, int32_t n_frames, const void *depth_raw, int32_t depth_h,
                             int32_t depth_w, int32_t depth_layout, const int32_t *depth_index, int32_t height, int32_t width,
                             double depth_thresh, const uint32_t *segmap, int32_t word_bits, const int32_t *frame_mask,
                             uint32_t *depth_lines, uint32_t *mask_lines, int64_t line_words, uint32_t *label_lines,
                             void *stream);
/* Measurement aid: n_lanes lanes each read one float at element lane * stride of src (every element once per launch)
 * -- a gather with a known number of distinct cache lines, to calibrate the FETCH_SIZE counter (scripts/diag_membw.py). */
int bff_diag_gather(const float *src, int64_t n_lanes, int64_t stride, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BFF_HIP_H */
