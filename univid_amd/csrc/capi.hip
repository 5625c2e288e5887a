// C-ABI plumbing shared by every entry point of libunivid_hip.so: error text, version, init.
// Every uv_* function returns 0 on success, non-zero on failure (with uv_last_error() set); nothing
// here or in the other translation units takes or returns a torch type.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <atomic>

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

// ---- developer options: explicit, process-wide, lock-free (see common.h) ----
static std::atomic<int> g_opt[UV_OPT_COUNT] = {{-1}, {0}, {0}};
static const int kOptDefault[UV_OPT_COUNT] = {-1, 0, 0};
int uv_option(int key) { return g_opt[key].load(std::memory_order_relaxed); }

extern "C" int uv_set_option(int key, int value) {
    UV_CHECK_ARG(key >= 0 && key < UV_OPT_COUNT, "uv_set_option: unknown key %d", key);
    if (key == UV_OPT_CONV_HALO) UV_CHECK_ARG(value >= -1 && value <= 1, "uv_set_option(UV_OPT_CONV_HALO): value %d not in {-1, 0, 1}", value);
    if (key == UV_OPT_GEMM_GM) UV_CHECK_ARG(value >= 0 && value <= 64, "uv_set_option(UV_OPT_GEMM_GM): value %d not in 0..64", value);
    if (key == UV_OPT_ATTN_CUT) UV_CHECK_ARG(value >= 0 && value <= 4096, "uv_set_option(UV_OPT_ATTN_CUT): value %d not in 0..4096", value);
    g_opt[key].store(value, std::memory_order_relaxed);
    return 0;
}

extern "C" int uv_get_option(int key, int* value) {
    UV_CHECK_ARG(key >= 0 && key < UV_OPT_COUNT && value, "uv_get_option: unknown key %d or null result pointer", key);
    *value = uv_option(key);
    return 0;
}

extern "C" int uv_reset_options(void) {
    for (int i = 0; i < UV_OPT_COUNT; ++i) g_opt[i].store(kOptDefault[i], std::memory_order_relaxed);
    return 0;
}

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

// Host scheduling policy of the CURRENT device: hipDeviceScheduleBlockingSync (the host thread sleeps on an interrupt inside
// hipStreamSynchronize / hipDeviceSynchronize / hipEventSynchronize instead of spinning on the completion signal) or the runtime's default.
// Process-wide policy, so never set implicitly: the application calls it once per device, before its first synchronize - bench.py's ranks
// and univid_amd.parallel's workers do, because eight ranks spinning on one host is the scaling risk SURVEY 8(e) names.
extern "C" int uv_host_blocking_sync(int on) {
    const hipError_t e = hipSetDeviceFlags(on ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        uv_set_error("uv_host_blocking_sync: hipSetDeviceFlags failed: %s", hipGetErrorString(e));
        return -1;
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
