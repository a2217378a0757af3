// C ABI housekeeping: version, workspace size, thread-local error text.
#include <stdarg.h>
#include <stdio.h>
#include "osq_host.h"

namespace osq {
static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
}  // namespace osq

extern "C" const char* osq_last_error(void) { return osq::g_error; }
extern "C" int osq_abi_version(void) { return 2; }
extern "C" size_t osq_workspace_bytes(void) { return osq::kWsHeaderBytes + osq::kWsScratchBytes + osq::kWsWideBytes + osq::kWsMeetBytes; }
