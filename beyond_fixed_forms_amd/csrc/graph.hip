// Connected components of the merge adjacency (include/bff_hip.h: a12).
//
// The reference computes the transitive closure with n dense matmuls
// (projection_2d_to_3d.py:258-260, O(n^4)); components of a symmetric 0/1 matrix are found here by
// min-label propagation over the bit adjacency + pointer jumping: O(n^2 / 64) word reads per round,
// O(log n) rounds.
#include "common.h"

namespace bff {

// one wave per node: lanes stride over the adjacency words of row i
__global__ __launch_bounds__(256) void cc_hook_kernel(const uint64_t *__restrict__ adj, int n, int aw,
                                                       const int32_t *__restrict__ lin, int32_t *__restrict__ lout,
                                                       int32_t *__restrict__ changed)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = lane_id();
    const int mine = lin[i];
    int best = mine;
    for (int w = lane; w < aw; w += kWave) {
        uint64_t bits = adj[(int64_t)i * aw + w];
        while (bits) {
            const int j = w * 64 + __ffsll((unsigned long long)bits) - 1;
            bits &= bits - 1;
            best = min(best, lin[j]);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) best = min(best, __shfl_xor(best, d));
    if (lane == 0) {
        lout[i] = best;
        if (best != mine) *changed = 1;
    }
}

__global__ void cc_jump_kernel(int32_t *__restrict__ label, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // labels only ever decrease towards the component minimum and label[k] <= k, so chasing the
    // chain terminates; concurrent updates by other threads can only shorten it
    int l = label[i];
    for (int hop = 0; hop < 64; ++hop) {
        const int p = label[l];
        if (p == l) break;
        l = p;
    }
    label[i] = l;
}

}  // namespace bff

using namespace bff;

extern "C" int bff_components_round(const uint64_t *adj, int32_t n_nodes, const int32_t *label_in,
                                    int32_t *label_out, int32_t *changed, void *stream)
{
    BFF_REQUIRE(n_nodes >= 0, "bff_components_round: bad size");
    if (n_nodes == 0) return BFF_OK;
    BFF_REQUIRE(adj && label_in && label_out && changed && label_in != label_out, "bff_components_round: bad pointers");
    const int aw = (int)ceil_div(n_nodes, 64);
    cc_hook_kernel<<<(unsigned)ceil_div(n_nodes, 4), 256, 0, as_stream(stream)>>>(adj, n_nodes, aw, label_in, label_out, changed);
    cc_jump_kernel<<<(unsigned)ceil_div(n_nodes, 256), 256, 0, as_stream(stream)>>>(label_out, n_nodes);
    return launched("bff_components_round");
}
