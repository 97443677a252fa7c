// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include "coda_common.h"

void coda_set_last_error(const char* what, hipError_t e);

#define CODA_CHECK_LAUNCH(what)                                  \
    do {                                                         \
        hipError_t e_ = hipGetLastError();                       \
        if (e_ != hipSuccess) {                                  \
            coda_set_last_error(what, e_);                       \
            return CODA_ERR_LAUNCH;                              \
        }                                                        \
    } while (0)
