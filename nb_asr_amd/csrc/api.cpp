// Host-only pieces of the C ABI: version, thread-local error text, padding / length rules.
#include "common.h"

#include <cstring>

namespace nbasr {

static thread_local char g_error[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

void clear_error() { g_error[0] = '\0'; }

}  // namespace nbasr

extern "C" int nbasr_version(void) { return NBASR_ABI_VERSION; }

#ifndef NBASR_BUILD_ID
#define NBASR_BUILD_ID "unknown"
#endif
extern "C" const char* nbasr_build_id(void) { return NBASR_BUILD_ID; }

extern "C" const char* nbasr_last_error(void) { return nbasr::g_error; }

extern "C" int nbasr_pad_amounts(int kernel, int dilation, int stride, int* left, int* right) {
    nbasr::clear_error();
    NBASR_REQUIRE(left && right, NBASR_ENULL, "nbasr_pad_amounts: NULL output pointer");
    NBASR_REQUIRE(kernel >= 1 && dilation >= 1 && stride >= 1, NBASR_EINVAL,
                  "nbasr_pad_amounts: kernel=%d dilation=%d stride=%d must all be >= 1", kernel, dilation, stride);
    *left = nbasr::pad_left(kernel, dilation, stride);
    *right = nbasr::pad_right(kernel, dilation, stride);
    return NBASR_OK;
}

extern "C" int nbasr_output_frames(int frames) {
    if (frames <= 0) return 0;
    const int half = (frames + 1) / 2;
    return (half + 1) / 2;
}
