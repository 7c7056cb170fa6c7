// p3hip — C-ABI housekeeping (error string, version)
#include <string.h>

#include "../../include/p3hip.h"

static thread_local char g_err[256] = "";

void p3_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

extern "C" int p3_version(void) { return 100; }
extern "C" const char* p3_last_error_string(void) { return g_err; }
