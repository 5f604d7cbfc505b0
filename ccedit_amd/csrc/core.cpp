// Error reporting / device info for libccedit_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

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

// ---- dispatch policy (common.h: CcPolicy) ----
static CcPolicy g_policy;
const CcPolicy& cc_policy() { return g_policy; }

namespace {
struct PolicyEntry {
    const char* name;
    int CcPolicy::*field;
};
const PolicyEntry kPolicy[] = {
    {"conv_halo", &CcPolicy::conv_halo},     {"g8", &CcPolicy::g8},           {"g8_conv", &CcPolicy::g8_conv},
    {"g8_temporal", &CcPolicy::g8_temporal}, {"g8_split", &CcPolicy::g8_split}, {"lin320", &CcPolicy::lin320},
    {"lin320s", &CcPolicy::lin320s},         {"lin640", &CcPolicy::lin640},   {"temp320", &CcPolicy::temp320},
    {"attn_short", &CcPolicy::attn_short},   {"attn_text", &CcPolicy::attn_text}, {"attn_spatial", &CcPolicy::attn_spatial},
    {"attn_pv16", &CcPolicy::attn_pv16},     {"attn_opt", &CcPolicy::attn_opt},       {"gn_flat", &CcPolicy::gn_flat}, {"gn_apply_flat", &CcPolicy::gn_apply_flat},
    {"f32_split", &CcPolicy::f32_split},
};
}  // namespace

extern "C" int ccedit_policy_set(const char* name, int32_t value) {
    CC_CHECK_ARG(name != nullptr, "ccedit_policy_set: null name");
    for (const PolicyEntry& e : kPolicy)
        if (!strcmp(e.name, name)) {
            g_policy.*(e.field) = value;
            return CCEDIT_OK;
        }
    cc_set_error("ccedit_policy_set: unknown switch '%s'", name);
    return CCEDIT_EINVAL;
}

extern "C" int ccedit_policy_get(const char* name, int32_t* value) {
    CC_CHECK_ARG(name != nullptr && value != nullptr, "ccedit_policy_get: null argument");
    for (const PolicyEntry& e : kPolicy)
        if (!strcmp(e.name, name)) {
            *value = g_policy.*(e.field);
            return CCEDIT_OK;
        }
    cc_set_error("ccedit_policy_get: unknown switch '%s'", name);
    return CCEDIT_EINVAL;
}

// Comma-separated names of the table, in declaration order
extern "C" const char* ccedit_policy_names(void) {
    static char buf[512] = "";
    if (!buf[0]) {
        size_t n = 0;
        for (const PolicyEntry& e : kPolicy) n += (size_t)snprintf(buf + n, sizeof(buf) - n, "%s%s", n ? "," : "", e.name);
    }
    return buf;
}
