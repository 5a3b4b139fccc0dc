// thread-local last-error string + version for the tdr C ABI
#include <stdarg.h>
#include <stdio.h>
#include "../../include/tdr.h"
static thread_local char g_err[512] = "";
void tdr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* tdr_last_error(void) { return g_err; }
extern "C" int tdr_version(void) { return 100; }
