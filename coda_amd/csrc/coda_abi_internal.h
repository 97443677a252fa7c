// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include "coda_common.h"

void coda_set_last_error(const char* what, hipError_t e);
void coda_set_last_error_msg(const char* msg);   // argument errors that deserve words (read back by coda_last_error)

// Per-device facts, read once per device (abi_misc.hip): the CU count of the CURRENT device, and the
// hipFuncAttributeMaxDynamicSharedMemorySize grant for (current device, kernel).  A process may hold engines on
// several GPUs: every launcher asks for the device that is current at the call.
int coda_device_cus();
int coda_ensure_dyn_lds(const void* func, int bytes);   // CODA_OK or CODA_ERR_LAUNCH

// RN(1 / t) if coda_div_by's three-operation quotient equals the IEEE quotient x / t for EVERY bf16 x of its fast-path
// domain (checked on the host, once per distinct t, ~0.3 ms; remembered), else 0 = "divide".  t <= 0: 0.
float coda_reciprocal_if_exact(float t);

#define CODA_CHECK_LAUNCH(what)                                  \
    do {                                                         \
        hipError_t e_ = hipGetLastError();                       \
        if (e_ != hipSuccess) {                                  \
            coda_set_last_error(what, e_);                       \
            return CODA_ERR_LAUNCH;                              \
        }                                                        \
    } while (0)
