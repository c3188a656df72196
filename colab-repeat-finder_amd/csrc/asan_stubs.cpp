// asan_stubs.cpp -- what the host-only sanitizer build (make asan) needs from api.cpp: the error string.
#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/prf.h"
#include "prf_plan.h"

static thread_local std::string g_err;

int prf_set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

extern "C" {
const char *prf_last_error(void) { return g_err.c_str(); }
int prf_abi_version(void) { return PRF_ABI_VERSION; }
int prf_plan_describe(uint32_t kmin, uint32_t kmax, uint32_t min_repeats, uint32_t min_span, char *buf, uint64_t buf_len) {
    if (!buf || buf_len == 0) return prf_set_error(PRF_EINVAL, "prf_plan_describe: no buffer");
    const int n = prf_plan_json(kmin, kmax, min_repeats, min_span, buf, buf_len);
    if (n < 0) return prf_set_error(PRF_EINVAL, "prf_plan_describe: buffer too small");
    return n;
}
}
