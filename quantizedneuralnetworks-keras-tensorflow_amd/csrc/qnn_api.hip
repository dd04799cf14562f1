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
std::atomic<int> g_option[QNN_OPT_COUNT] = {{1}, {-1}, {0}, {0}, {1}};    // strip kernel on, Cin 64: auto, halo kernel on
}  // namespace

int qnn_option(int which) { return g_option[which].load(std::memory_order_relaxed); }

// domain declared for the float32 input of the conv call in flight on this thread: 0 none, 1 image bytes / 255, 2 [0, 1]
namespace { thread_local int g_first_mode = 0; }
void qnn_set_call_first_mode(int mode) { g_first_mode = mode; }
int qnn_call_first_mode() { return g_first_mode; }

extern "C" int qnn_set_option(const char* key, int value) {
    if (key && strcmp(key, "strip") == 0) {
        g_option[QNN_OPT_STRIP].store(value ? 1 : 0, std::memory_order_relaxed);
        return QNN_OK;
    }
    if (key && strcmp(key, "strip64") == 0) {
        g_option[QNN_OPT_STRIP64].store(value < 0 ? -1 : value ? 1 : 0, std::memory_order_relaxed);
        return QNN_OK;
    }
    if (key && strcmp(key, "first_fixed") == 0) {
        g_option[QNN_OPT_FIRST_FIXED].store(value ? 1 : 0, std::memory_order_relaxed);
        return QNN_OK;
    }
    if (key && strcmp(key, "halo") == 0) {
        g_option[QNN_OPT_HALO].store(value ? 1 : 0, std::memory_order_relaxed);
        return QNN_OK;
    }
    if (key && strcmp(key, "first_image") == 0) {
        g_option[QNN_OPT_FIRST_IMAGE].store(value ? 1 : 0, std::memory_order_relaxed);
        return QNN_OK;
    }
    qnn_set_error("qnn_set_option: unknown key '%s'", key ? key : "(null)");
    return QNN_EINVAL;
}

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
