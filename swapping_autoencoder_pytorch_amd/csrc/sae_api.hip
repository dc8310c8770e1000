// ABI version + thread-local error message (include/sae_hip.h).
#include "sae_common.h"

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

}  // namespace sae

extern "C" int sae_abi_version(void) { return SAE_ABI_VERSION; }
extern "C" const char* sae_last_error(void) { return sae::err_buf(); }
