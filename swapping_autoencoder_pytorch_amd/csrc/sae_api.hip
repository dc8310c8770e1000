// ABI version + thread-local error message (include/sae_hip.h).
#include "sae_common.h"

#include <atomic>
#include <cstdlib>
#include <cstring>

namespace sae {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

namespace {
// the ONE piece of process-wide mutable state of the library (include/sae_hip.h): an atomic so that a switch
// racing with convolutions on other threads is at worst observed late, never torn
std::atomic<int>& conv_math_ref() {
    static std::atomic<int> mode([] {
        const char* e = getenv("SAE_CONV_MATH");
        return (e && (!strcmp(e, "bf16x6") || !strcmp(e, "1"))) ? 1 : 0;
    }());
    return mode;
}
}  // namespace

int conv_math() { return conv_math_ref().load(std::memory_order_relaxed); }

}  // namespace sae

extern "C" int sae_set_conv_math(int32_t mode) {
    if (mode != SAE_CONV_MATH_F32 && mode != SAE_CONV_MATH_BF16X6)
        return sae::fail(SAE_EINVAL, "sae_set_conv_math: unknown mode %d", (int)mode);
    sae::conv_math_ref().store(mode, std::memory_order_relaxed);
    return SAE_OK;
}
extern "C" int sae_get_conv_math(void) { return sae::conv_math(); }

extern "C" int sae_abi_version(void) { return SAE_ABI_VERSION; }
extern "C" const char* sae_last_error(void) { return sae::err_buf(); }
