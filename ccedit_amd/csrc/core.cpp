// Error reporting / device info for libccedit_hip.so.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ccedit_hip.h"

static thread_local char g_err[512] = "";

void cc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local char g_kernel[96] = "";

// Name of the kernel template a GEMM / attention entry point dispatched to (for per-kernel roofline accounting in bench.py)
void cc_note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}

extern "C" const char* ccedit_last_kernel(void) { return g_kernel; }

extern "C" int ccedit_abi_version(void) { return CCEDIT_ABI_VERSION; }

extern "C" const char* ccedit_last_error(void) { return g_err; }

extern "C" int ccedit_device_info(char* name, int name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        cc_set_error("hipGetDevice: %s", hipGetErrorString(e));
        return -(int)e;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) {
        cc_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
        return -(int)e;
    }
    if (name && name_len > 0) {
        strncpy(name, prop.gcnArchName, (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    return prop.multiProcessorCount;
}
