// Library-level entry points of libbff_hip.so.
#include <vector>

#include "common.h"

namespace bff {
char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int &scratch_prezeroed() {
    static thread_local int flag = 0;
    return flag;
}
}  // namespace bff

extern "C" int bff_abi_version(void) { return BFF_ABI_VERSION; }
extern "C" const char *bff_last_error(void) { return bff::err_buf(); }
extern "C" const char *bff_arch(void) { return "gfx950"; }

// ---- host-side helper (no GPU involved) -------------------------------------------------------------
// Component ids -> the groups merge_masks keeps (P:203-226) in CSR form: the native twin of
// projection.component_csr (same contract; one counting sort instead of NumPy argsorts).
//   comp[i] in [0, n): any integer naming the connected component of row i
//   kept groups: components with >= max(min_members, 1) members, except isolated rows without a self loop
//   (the reference's empty lists; counted in *n_void when min_members <= 0, where they survive the filter);
//   order: by smallest member; members ascending.
// Returns K and fills offs[0..K], members[0..offs[K]), sizes[0..K); -1 if an id is out of range.
extern "C" int bff_host_component_csr(const int32_t *comp, const uint8_t *has_self_loop, int32_t n, int32_t min_members,
                                      int32_t *offs, int32_t *members, int32_t *sizes, int32_t *n_void)
{
    if (n < 0 || !comp || !has_self_loop || !offs || !members || !sizes || !n_void) return -1;
    std::vector<int32_t> count(n + 1, 0), first(n, -1), ids;
    ids.reserve(64);
    for (int i = 0; i < n; ++i) {
        const int c = comp[i];
        if (c < 0 || c >= n) return -1;
        if (first[c] < 0) { first[c] = i; ids.push_back(c); }   // ids in order of their smallest member
        ++count[c];
    }
    const int need = min_members > 1 ? min_members : 1;
    std::vector<int32_t> start(n, -1);
    int k = 0, pos = 0, voids = 0;
    for (int c : ids) {
        const int sz = count[c];
        if (sz == 1 && !has_self_loop[first[c]]) { if (min_members <= 0) ++voids; continue; }
        if (sz < need) continue;
        start[c] = pos;
        offs[k] = pos;
        sizes[k] = sz;
        pos += sz;
        ++k;
    }
    offs[k] = pos;
    for (int i = 0; i < n; ++i) {                               // ascending i -> members ascending within a group
        const int c = comp[i];
        if (start[c] >= 0) members[start[c]++] = i;
    }
    *n_void = voids;
    return k;
}
