// C-ABI plumbing shared by every entry point of libunivid_hip.so: error text, version, init.
// Every uv_* function returns 0 on success, non-zero on failure (with uv_last_error() set); nothing
// here or in the other translation units takes or returns a torch type.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void uv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* uv_last_error(void) { return g_err; }

extern "C" int uv_version(void) { return 200; }  // 0.2.0

// Digest of the kernel sources this library was built from (univid_amd/build.py: source_id()); the loader compares it with
// the tree so that a stale .so never runs behind newer bindings.
#ifndef UV_BUILD_ID
#define UV_BUILD_ID "unstamped"
#endif
extern "C" const char* uv_build_id(void) { return UV_BUILD_ID; }

const float* uv_zero_page();

// Allocates the library's few persistent device objects (outside any stream capture).
extern "C" int uv_init(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        uv_set_error("uv_init: no HIP device visible");
        return -2;
    }
    if (!uv_zero_page()) {
        uv_set_error("uv_init: zero page allocation failed");
        return -3;
    }
    return 0;
}

// Device arch string of the current device, for the loader's "is this gfx950" check.
extern "C" int uv_device_arch(char* buf, int len) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        uv_set_error("uv_device_arch: no device");
        return -2;
    }
    snprintf(buf, len, "%s", prop.gcnArchName);
    return 0;
}
