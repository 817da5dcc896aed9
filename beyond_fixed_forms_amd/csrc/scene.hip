// One scene, one call: every device step of tools/projection_2d_to_3d.py:402-634 (+ the first device pass of
// tools/refinement.py:186-217) issued on one stream without a single host dependency, results in ONE pinned
// read-back.  The host thread spends one native call per scene instead of ~45 Python-issued launches and two
// stream synchronisations; everything here is a sequence of the library's own entry points (include/bff_hip.h),
// so each step keeps its contract and its tests.
//
//   decode -> sweep -> point filter threshold -> keep bits -> row statistics -> tile order -> components ->
//   groups (device) -> OR of members + confidence means -> (rows recycled) -> overlaps + filter + counts ->
//   caller point order -> stage-1 decode -> stage-1 x (stage-2 | stage-1) intersections -> header to the host
#include "common.h"

using namespace bff;

namespace bff {
int merge_components_streams(const uint64_t *rows, int32_t n_rows, int64_t nw, const int32_t *order,
                             int32_t n_order, const uint64_t *chunk_mask, uint64_t *tile_mask,
                             const uint32_t *hist, uint32_t *scratch, const int32_t *area,
                             const int32_t *label_id, float iou_thres, int32_t *parent, int32_t init_parent,
                             int32_t *comp, int32_t *diag, const uint16_t *chunk_pop, void *stream, void *heavy_stream,
                             void *before_heavy, void *after_heavy);
}

namespace {

__global__ void order_keys_kernel(const int64_t *__restrict__ sig, const int32_t *__restrict__ label_id, int n,
                                  int sig_bits, int64_t *__restrict__ keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ((int64_t)label_id[i] << sig_bits) | sig[i];     // stable sort by (label, signature)
}

int bits_for(int n)
{
    int b = 1;
    while ((1 << b) < n && b < 31) ++b;
    return b;
}

}  // namespace

#define BFF_TRY(call) do { const int rc_ = (call); if (rc_ != BFF_OK) return rc_; } while (0)

extern "C" int32_t bff_scene_header_words(int32_t s1_rows, int32_t cap)
{
    return BFF_HDR_CROSS(cap) + s1_rows * (cap + s1_rows);
}

extern "C" int32_t bff_scene_struct_bytes(int32_t which)
{
    return which == 0 ? (int32_t)sizeof(bff_scene) : which == 1 ? (int32_t)sizeof(bff_scene_params)
                                                                 : (int32_t)sizeof(bff_scene_workspace);
}

extern "C" int bff_scene_project(const bff_scene *sc, const bff_scene_params *pr, const bff_scene_workspace *ws,
                                 void *stream)
{
    BFF_REQUIRE(sc && pr && ws, "bff_scene_project: null struct");
    BFF_REQUIRE(sc->n_points > 0 && sc->n_rows > 0 && sc->n_mviews > 0 && sc->n_frames > 0,
                "bff_scene_project: empty scenes take the general path");
    BFF_REQUIRE(ws->hdr && ws->hdr_host, "bff_scene_project: no header buffers");
    hipStream_t st = as_stream(stream);
    const int64_t n = sc->n_points, nw = sc->nw, hw = (int64_t)sc->height * sc->width;
    const int n_rows = sc->n_rows, cap = ws->group_cap;
    BFF_REQUIRE(cap == BFF_GROUP_CAP || cap == BFF_GROUP_CAP_MAX, "bff_scene_project: group_cap must be %d or %d", BFF_GROUP_CAP, BFF_GROUP_CAP_MAX);
    const int mw = bff_chunk_mask_words(nw);
    int32_t *hdr = ws->hdr;
    hipError_t e;
#define BFF_ZERO(ptr, bytes) do { e = hipMemsetAsync((ptr), 0, (bytes), st); \
        if (e != hipSuccess) return fail((int)e, "bff_scene_project: memset: %s", hipGetErrorString(e)); } while (0)

    // ONE fill clears every scratch buffer of the call: counters, chunk flags, segment bitmap, header, merge lists and
    // split slots, the OR targets (bff_scene_workspace lays them out in one block); the steps'
    // own clears are switched off for the duration of the call
    BFF_REQUIRE(ws->zero_bytes > 0 && ws->masked, "bff_scene_project: no zero block");
    {
        const char *z0 = reinterpret_cast<const char *>(ws->masked), *z1 = z0 + ws->zero_bytes;
        auto inside = [&](const void *p, size_t bytes) {
            const char *c = reinterpret_cast<const char *>(p);
            return p != nullptr && c >= z0 && c + bytes <= z1;
        };
        const size_t seg_bytes = sizeof(uint32_t) * 2 * (size_t)sc->n_mviews * (size_t)ceil_div(ceil_div(hw, 128), 32);
        const bool use_cpop = bff_merge_uses_chunk_bound(nw) != 0;
        BFF_REQUIRE(inside(ws->masked, sizeof(int32_t) * (size_t)n) && inside(ws->viewed, sizeof(int32_t) * (size_t)n) &&
                    inside(ws->count, sizeof(int32_t) * (size_t)n_rows) &&
                    inside(ws->chunk_mask, sizeof(uint64_t) * (size_t)n_rows * mw) && inside(ws->segmap, seg_bytes) &&
                    inside(hdr, sizeof(int32_t) * (size_t)bff_scene_header_words(sc->s1_rows, cap)) &&
                    inside(ws->agg, sizeof(uint64_t) * (size_t)cap * nw) &&
                    inside(ws->merge_scratch, sizeof(uint32_t) * (size_t)bff_merge_scratch_words(n_rows)) &&
                    (!use_cpop || inside(ws->chunk_pop, sizeof(uint16_t) * (size_t)n_rows * mw * 64)),
                    "bff_scene_project: a scratch buffer lies outside the zero block (workspace layout)");
    }
    BFF_ZERO(ws->masked, ws->zero_bytes);
    struct Prezeroed {
        Prezeroed() { scratch_prezeroed() = 1; }
        ~Prezeroed() { scratch_prezeroed() = 0; }
    } prezeroed_scope;
    // The chip-filling kernels (decode, sweep, tile pass) go to the workspace's heavy stream when it has one: `hv`.
    // Hand-overs by events: the call's stream -> heavy (after the fill) -> back (after the sweep) -> heavy (tile pass,
    // inside merge_components_streams) -> back.
    void *hv = (ws->heavy_stream && ws->heavy_stream != stream) ? ws->heavy_stream : stream;
    const bool two = hv != stream;
    auto hand_over = [&](void *event, void *from, void *to) -> int {
        hipError_t ee = hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(from));
        if (ee == hipSuccess) ee = hipStreamWaitEvent(as_stream(to), reinterpret_cast<hipEvent_t>(event), 0);
        return ee == hipSuccess ? BFF_OK : fail((int)ee, "bff_scene_project: stream hand-over: %s", hipGetErrorString(ee));
    };
    if (two) {
        BFF_REQUIRE(ws->events[0] && ws->events[1] && ws->events[2] && ws->events[3], "bff_scene_project: a heavy stream needs the workspace's four events");
        BFF_TRY(hand_over(ws->events[0], stream, hv));
    }
    // a1: 2-D RLE -> label plane (+ words where masks overlap, + segment bitmap)
    BFF_TRY(bff_rle_to_labels(sc->run_start, sc->run_end, sc->mask_run_offs, sc->view_mask_offs, sc->n_mviews, hw,
                              sc->word_bits, ws->labels, ws->maskbits, ws->segmap, hv));
    // a2-a8 (+a15): the fused sweep.  ws->rows is all zero on entry (and again on exit, see below)
    const bool ratio = pr->filter_mode == 2;
    if (sc->depth_raw)          // depth as the PNGs store it: /1000 + bilinear resize per point inside the sweep
        BFF_TRY(bff_project_views_u16(sc->xyz, n, sc->n_pad, sc->inv_pose, sc->cam_intr, sc->n_frames, sc->depth_raw,
                                      sc->depth_h, sc->depth_w, sc->depth_tiled, sc->depth_index, sc->height, sc->width, pr->depth_thresh,
                                      ws->maskbits, ws->labels, ws->segmap, sc->word_bits, sc->frame_mask, sc->frame_rowbase,
                                      sc->frame_nmask, sc->frame_flags, ws->rows, n_rows, nw, ws->chunk_mask, ws->masked,
                                      ratio ? ws->viewed : nullptr, sc->tile_bounds, hv));
    else
        BFF_TRY(bff_project_views(sc->xyz, n, sc->n_pad, sc->inv_pose, sc->cam_intr, sc->n_frames, sc->depth, sc->depth_index,
                                  sc->height, sc->width, pr->depth_thresh, ws->maskbits, ws->labels, ws->segmap, sc->word_bits,
                                  sc->frame_mask, sc->frame_rowbase, sc->frame_nmask, sc->frame_flags, ws->rows, n_rows, nw,
                                  ws->chunk_mask, ws->masked, ratio ? ws->viewed : nullptr, sc->tile_bounds, hv));
    if (two) BFF_TRY(hand_over(ws->events[1], hv, stream));
    // a14 / a15: point filter, threshold stays on the device (header words 2, 3 = n_unique, thr).  Nothing before the
    // overlap resolution reads `keep`: with an aux stream the chain forks here and joins there.
    const bool fork = ws->aux_stream && ws->aux_stream != stream && pr->filter_mode != 0 && !pr->filter_sort;
    void *const main_stream = stream;
    if (fork) {
        BFF_REQUIRE(ws->aux_events[0] && ws->aux_events[1], "bff_scene_project: an aux stream needs the workspace's two aux events");
        BFF_TRY(hand_over(ws->aux_events[0], main_stream, ws->aux_stream));
        stream = ws->aux_stream;
    }
    if (pr->filter_mode != 0) {
        if (pr->filter_sort) {          // the general formulation: sort all n values (more distinct ones than the set holds)
            BFF_TRY(bff_point_values(ws->masked, ratio ? ws->viewed : nullptr, n, ws->vals, stream));
            size_t tb = ws->sort_temp_bytes;
            BFF_TRY(bff_sort_f32(ws->vals, ws->vals_sorted, n, ws->sort_temp, &tb, stream));
            BFF_TRY(bff_select_unique_rank(ws->vals_sorted, n, pr->filter_fraction, ws->sel_scratch,
                                           reinterpret_cast<float *>(hdr + BFF_HDR_THR), hdr + BFF_HDR_NUNIQUE, stream));
        } else {                        // the statistic is a function of (masked, viewed): distinct values, no sort
            BFF_TRY(bff_point_threshold_pairs(ws->masked, ratio ? ws->viewed : nullptr, n, pr->filter_fraction,
                                              ws->pair_scratch, reinterpret_cast<float *>(hdr + BFF_HDR_THR),
                                              hdr + BFF_HDR_NUNIQUE, hdr + BFF_HDR_OVERFLOW, stream));
        }
        BFF_TRY(bff_ratio_keep(ws->masked, ratio ? ws->viewed : nullptr, n, 0.0f,
                               reinterpret_cast<const float *>(hdr + BFF_HDR_THR), 1, nw, ws->keep, stream));
    } else {
        BFF_TRY(bff_ratio_keep(ws->masked, nullptr, n, 0.0f, nullptr, 0, nw, ws->keep, stream));
    }
    if (fork) {
        hipError_t ee = hipEventRecord(reinterpret_cast<hipEvent_t>(ws->aux_events[1]), as_stream(stream));
        if (ee != hipSuccess) return fail((int)ee, "bff_scene_project: aux stream: %s", hipGetErrorString(ee));
        stream = main_stream;
    }
    // a9-a12: statistics through the chunk flags, tile order (label, signature), components
    uint16_t *cpop = bff_merge_uses_chunk_bound(nw) ? ws->chunk_pop : nullptr;
    BFF_TRY(bff_row_stats(ws->rows, n_rows, nw, ws->area, ws->mean_word, ws->chunk_mask, 1, ws->hist, ws->sig, cpop, stream));
    {
        const int64_t *keys = ws->sig;
        int key_bits = BFF_SIGNATURE_BITS;
        if (sc->n_label_ids > 1) {
            key_bits += bits_for(sc->n_label_ids);
            order_keys_kernel<<<(unsigned)ceil_div(n_rows, 256), 256, 0, st>>>(ws->sig, sc->label_id, n_rows,
                                                                              BFF_SIGNATURE_BITS, ws->sig_keys);
            keys = ws->sig_keys;
        }
        size_t tb = ws->sort_temp_bytes;
        BFF_TRY(bff_argsort_i64(keys, ws->sig_sorted, ws->order, n_rows, key_bits, ws->sort_temp, &tb, stream));
    }
    // the forest is initialised by the tile pre-pass and flattened by the grouping step (comp == NULL here)
    BFF_TRY(merge_components_streams(ws->rows, n_rows, nw, ws->order, n_rows, ws->chunk_mask, ws->tile_mask, ws->hist,
                                     ws->merge_scratch, ws->area, sc->label_id, pr->iou_thres, ws->parent, 1, nullptr, nullptr,
                                     cpop, stream, hv, ws->events[2], ws->events[3]));
    // P:203-226 on the device: groups, OR of the members, sequential confidence means
    int32_t *info = hdr + BFF_HDR_K;
    BFF_TRY(bff_group_components(ws->comp, ws->parent, ws->area, n_rows, pr->iou_thres, pr->min_members, cap, ws->count, 1, info,
                                 hdr + BFF_HDR_SIZES(cap), hdr + BFF_HDR_FIRST(cap), ws->goffs, ws->gmembers, ws->slices, stream));
    BFF_TRY(bff_or_reduce_grouped(ws->rows, nw, n_rows, info, cap, ws->goffs, ws->gmembers, ws->slices, ws->agg, sc->conf,
                                  sc->conf_f16, hdr + BFF_HDR_CONF(cap), ws->chunk_mask, stream));
    // last reader of the raw rows is done: give the arena its zeros back -- unless the host has to take the general
    // path (more groups than the device forms), which reads the rows again and clears them itself
    BFF_TRY(bff_clear_flagged_chunks_unless(ws->rows, n_rows, nw, ws->chunk_mask, info + 1, stream));
    if (fork) {         // join: `keep` and the header's threshold words are complete
        hipError_t ee = hipStreamWaitEvent(st, reinterpret_cast<hipEvent_t>(ws->aux_events[1]), 0);
        if (ee != hipSuccess) return fail((int)ee, "bff_scene_project: aux stream: %s", hipGetErrorString(ee));
    }
    // a16 + P:592-596: overlap decisions (one prefix OR in priority order), &= keep, both popcounts
    BFF_TRY(bff_resolve_overlaps_dev(ws->agg, cap, nw, hdr + BFF_HDR_SIZES(cap), ws->keep, hdr + BFF_HDR_BEFORE(cap),
                                     hdr + BFF_HDR_AFTER(cap), info, stream));
    // caller's point order (scatter of the set bits); the refinement's first device pass (R:186-217) rides along when
    // stage 1 is resident.  `both` is the caller's buffer (it outlives the workspace's reuse): its clear is the one
    // fill besides the block's
    if (sc->perm) {
        BFF_ZERO(ws->both, sizeof(uint64_t) * (size_t)cap * nw);
        BFF_TRY(bff_scatter_bits(ws->agg, cap, nw, sc->perm, n, nw, ws->both, info, stream));
    } else {
        e = hipMemcpyAsync(ws->both, ws->agg, sizeof(uint64_t) * (size_t)cap * nw, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return fail((int)e, "bff_scene_project: copy: %s", hipGetErrorString(e));
    }
    if (sc->s1_rows > 0) {
        uint64_t *s1 = ws->both + (size_t)cap * nw;
        BFF_TRY(bff_rle_to_rows(sc->s1_run_start, sc->s1_run_end, sc->s1_row_run_offs, sc->s1_rows, n, nw, s1, stream));
        BFF_TRY(bff_cross_popcount_dev(s1, sc->s1_rows, ws->both, cap + sc->s1_rows, nw, hdr + BFF_HDR_CROSS(cap), info, 0, cap,
                                       stream));
    }
    e = hipMemcpyAsync(ws->hdr_host, hdr, sizeof(int32_t) * (size_t)bff_scene_header_words(sc->s1_rows, cap),
                       hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return fail((int)e, "bff_scene_project: header copy: %s", hipGetErrorString(e));
    return BFF_OK;
#undef BFF_ZERO
}
