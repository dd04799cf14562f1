// Library-level entry points: version, thread-local error / kernel-name strings.
#include <atomic>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "qnn_common.h"

namespace {
thread_local char g_error[512] = "";
thread_local char g_kernel[64] = "";
std::atomic<int> g_conv_impl{0};
}  // namespace

int qnn_conv_impl_pref() { return g_conv_impl.load(std::memory_order_relaxed); }

extern "C" int qnn_set_conv_impl(int impl) {
    if (impl < 0 || impl > 2) {
        qnn_set_error("qnn_set_conv_impl: impl=%d (0 auto, 1 valu, 2 mfma)", impl);
        return QNN_EINVAL;
    }
    g_conv_impl.store(impl, std::memory_order_relaxed);
    return QNN_OK;
}

void qnn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

void qnn_set_kernel_name(const char* name) {
    snprintf(g_kernel, sizeof(g_kernel), "%s", name);
}

extern "C" int qnn_version(void) { return QNN_ABI_VERSION; }
extern "C" const char* qnn_last_error(void) { return g_error; }
extern "C" const char* qnn_last_kernel(void) { return g_kernel; }
