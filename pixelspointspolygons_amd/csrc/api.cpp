// p3hip — C-ABI housekeeping (error string, version)
#include <string.h>

#include "../../include/p3hip.h"

static thread_local char g_err[256] = "";

void p3_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

// which kernel did the last p3_gemm of this thread launch (bench.py's roofline object labels its HIP-event timings with the name rocprofv3 shows)
static thread_local char g_kernel[96] = "";
static int g_trace = 0;
int p3_tracing(void) { return g_trace; }
void p3_note_kernel(const char* name) {
    strncpy(g_kernel, name ? name : "", sizeof(g_kernel) - 1);
    g_kernel[sizeof(g_kernel) - 1] = 0;
}
extern "C" void p3_trace_kernels(int on) { g_trace = on; g_kernel[0] = 0; }
extern "C" const char* p3_last_kernel(void) { return g_kernel; }

extern "C" int p3_version(void) { return 100; }
extern "C" const char* p3_last_error_string(void) { return g_err; }
