// Error plumbing of the C ABI: int status + thread-local message (include/scail_hip.h).
#include "common.h"

static thread_local std::string g_last_error;

void scail_set_error(const std::string& msg) { g_last_error = msg; }

int scail_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_error = std::string(what) + ": launch failed: " + hipGetErrorString(e);
        return 2;
    }
    return 0;
}

extern "C" const char* scail_last_error(void) { return g_last_error.c_str(); }
extern "C" int scail_abi_version(void) { return 5; }

// Device-side caches of the library that outlive a call: the GEMM's tile-order tables (one small hipMalloc per tile grid and device).
int scail_gemm4_release_tables();   // gemm.hip
extern "C" int scail_release_caches(void) { return scail_gemm4_release_tables(); }
