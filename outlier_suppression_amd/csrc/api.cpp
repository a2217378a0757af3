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
static thread_local int g_time_which = 0;
static thread_local hipEvent_t g_time_start = nullptr, g_time_stop = nullptr;

TimingHook take_timing_hook(int which) {
    if (which != g_time_which) return {nullptr, nullptr};
    const TimingHook h{g_time_start, g_time_stop};
    g_time_which = 0;
    g_time_start = g_time_stop = nullptr;
    return h;
}
}  // namespace osq

// ---- measurement aid: HIP events attached to a dispatch packet (hipExtLaunchKernelGGL) time the kernel's
// own execution -- what rocprofv3 --kernel-trace reports -- instead of the stream-order interval between
// two recorded events, which also contains the dispatch latency of the kernel boundary (~2-3 us).
extern "C" int osq_timing_events_create(void** start, void** stop) {
    OSQ_REQUIRE(start && stop, "timing_events_create: null pointer");
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        osq::set_error("timing_events_create: hipEventCreate failed");
        return OSQ_ERR_HIP;
    }
    *start = a;
    *stop = b;
    return OSQ_OK;
}

extern "C" int osq_timing_events_destroy(void* start, void* stop) {
    if (start) (void)hipEventDestroy(static_cast<hipEvent_t>(start));
    if (stop) (void)hipEventDestroy(static_cast<hipEvent_t>(stop));
    return OSQ_OK;
}

extern "C" int osq_time_next_launch(int which, void* start, void* stop) {
    OSQ_REQUIRE(which >= OSQ_TIME_NONE && which <= OSQ_TIME_OBSERVE_TOKENS, "time_next_launch: unknown kernel family");
    OSQ_REQUIRE((start == nullptr) == (stop == nullptr), "time_next_launch: give both events or neither");
    osq::g_time_which = start ? which : 0;
    osq::g_time_start = static_cast<hipEvent_t>(start);
    osq::g_time_stop = static_cast<hipEvent_t>(stop);
    return OSQ_OK;
}

extern "C" int osq_timing_elapsed_us(void* start, void* stop, float* us) {
    OSQ_REQUIRE(start && stop && us, "timing_elapsed_us: null pointer");
    float ms = 0.f;
    if (hipEventSynchronize(static_cast<hipEvent_t>(stop)) != hipSuccess ||
        hipEventElapsedTime(&ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)) != hipSuccess) {
        (void)hipGetLastError();      // the runtime's last-error word is sticky: the next launch check must not report THIS failure
        osq::set_error("timing_elapsed_us: events not recorded");
        return OSQ_ERR_HIP;
    }
    *us = ms * 1000.0f;
    return OSQ_OK;
}

extern "C" const char* osq_last_error(void) { return osq::g_error; }
extern "C" int osq_abi_version(void) { return OSQ_ABI_VERSION; }
extern "C" int osq_build_flags(void) {
    int f = 0;
#ifdef OSQ_TUNABLE
    f |= OSQ_BUILD_TUNABLE;
#endif
#ifdef OSQ_FINAL_TIMING
    f |= OSQ_BUILD_FINAL_TIMING;
#endif
    return f;
}
extern "C" size_t osq_workspace_bytes(void) { return osq::kWsHeaderBytes + osq::kWsScratchBytes + osq::kWsWideBytes + osq::kWsMeetBytes + osq::kWsFusedBytes + osq::kWsResidentBytes; }
