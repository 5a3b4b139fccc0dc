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
// 100: rounds 1-2.  101: round 3 changed tdr_patchify / tdr_vit_assemble / tdr_attention_fwd_math (a `flat` argument in the middle of the
// list) without bumping -- a stale library would bind shifted arguments.  102: round 4 (P16 entry points).  _lib.load() checks it.
extern "C" int tdr_version(void) { return 107; }
