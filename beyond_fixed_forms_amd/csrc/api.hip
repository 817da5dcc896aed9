// Library-level entry points of libbff_hip.so.
#include "common.h"

namespace bff {
char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace bff

extern "C" int bff_abi_version(void) { return BFF_ABI_VERSION; }
extern "C" const char *bff_last_error(void) { return bff::err_buf(); }
extern "C" const char *bff_arch(void) { return "gfx950"; }
