// Version / diagnostics entry points of the C ABI.
#include "coda_abi_internal.h"
#include <stdio.h>

static thread_local char g_last_error[256] = "";

void coda_set_last_error(const char* what, hipError_t e) {
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
}

extern "C" int coda_version(void) { return CODA_ABI_VERSION; }
extern "C" const char* coda_last_error(void) { return g_last_error; }
