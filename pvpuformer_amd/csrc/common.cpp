// Error reporting shared by the C-ABI entry points (thread-local last-error string).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../include/vpu_hip.h"

static thread_local char g_err[512] = "";

void vpu_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

int vpu_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return -3;
    }
    return 0;
}

extern "C" const char* vpu_last_error(void) { return g_err; }
extern "C" int vpu_abi_version(void) { return 2; }   // 2: vpu_gemm_desc.cs_tn / cs_t0 / cs_ld
