// Version / diagnostics entry points of the C ABI, and the only host-side state the library keeps:
// per-device facts that never change once read (CU count, "this kernel's dynamic-LDS limit has been
// raised on this device").  Nothing here is tunable state: results do not depend on it.
#include "coda_abi_internal.h"
#include <stdio.h>
#include <mutex>
#include <vector>

static thread_local char g_last_error[256] = "";

void coda_set_last_error(const char* what, hipError_t e) {
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
}

void coda_set_last_error_msg(const char* msg) { snprintf(g_last_error, sizeof(g_last_error), "%s", msg); }

extern "C" int coda_version(void) { return CODA_ABI_VERSION; }
extern "C" const char* coda_last_error(void) { return g_last_error; }

namespace {
constexpr int MAX_DEV = 64;
struct DevState {
    int cus = 0;                                  // 0 = not read yet
    std::vector<std::pair<const void*, int>> lds; // (kernel, bytes already granted)
};
std::mutex g_mu;
DevState g_dev[MAX_DEV];
}  // namespace

int coda_device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return 256;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_dev[dev].cus == 0) {
        int cus = 0;
        g_dev[dev].cus = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus : 256;
    }
    return g_dev[dev].cus;
}

int coda_ensure_dyn_lds(const void* func, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return CODA_ERR_LAUNCH;
    if (dev >= 0 && dev < MAX_DEV) {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& e : g_dev[dev].lds)
            if (e.first == func && e.second >= bytes) return CODA_OK;
    }
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        coda_set_last_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e);
        return CODA_ERR_LAUNCH;
    }
    if (dev >= 0 && dev < MAX_DEV) {
        std::lock_guard<std::mutex> lk(g_mu);
        g_dev[dev].lds.emplace_back(func, bytes);
    }
    return CODA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// coda_div_by's licence (coda_common.h): the sequence is run here exactly as the device runs it (fp32 multiply, two fused
// multiply-adds) for every bf16 x in its fast-path domain and compared with the IEEE quotient.  A read-once fact per
// temperature, like the CU count: results never depend on whether it is cached.
#include <math.h>
#include <string.h>
namespace {
std::mutex g_rcp_mu;
std::vector<std::pair<float, float>> g_rcp;      // (t, rt or 0)
}

float coda_reciprocal_if_exact(float t) {
    if (!(t > 0.f) || !(t >= 1e-3f && t <= 1e3f)) return 0.f;
    {
        std::lock_guard<std::mutex> lk(g_rcp_mu);
        for (auto& e : g_rcp)
            if (e.first == t) return e.second;
    }
    const float rt = (float)(1.0 / (double)t);
    bool exact = true;
    for (unsigned int b = 0; b < 65536u && exact; ++b) {
        const unsigned int bits = b << 16;
        float x;
        memcpy(&x, &bits, 4);
        if (!coda_div_fast_domain(bits)) continue;                      // the device divides there
        volatile float q0 = x * rt;                                       // volatile: one fp32 rounding per operation
        volatile float r = fmaf(-q0, t, x);
        volatile float q1 = fmaf(r, rt, q0);
        volatile float q = x / t;
        if (!(q1 == q)) exact = false;
    }
    const float out = exact ? rt : 0.f;
    std::lock_guard<std::mutex> lk(g_rcp_mu);
    if (g_rcp.size() < 4096) g_rcp.emplace_back(t, out);
    return out;
}

extern "C" int coda_temperature_fast_division(float temperature) { return coda_reciprocal_if_exact(temperature) != 0.f ? 1 : 0; }
