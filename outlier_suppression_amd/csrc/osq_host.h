// Host-side helpers of the C ABI: argument checks, error text, launch checks,
// workspace carving.  No torch types anywhere below the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/osq_hip.h"

namespace osq {

constexpr int kMaxBlocks = 2048;          // 256 CUs x 8 workgroups of 256 threads
constexpr size_t kWsHeaderBytes = 4096;   // ticket counters, one 64-byte line each (33 used)
constexpr size_t kWsScratchBytes = 64 * 1024;
constexpr size_t kWsWideBytes = 2 * 2048 * 4 + 64;   // WideState of the multi-workgroup token finaliser
constexpr size_t kWsMeetBytes = 64 * 1024;           // rendezvous words of token_select_kernel (8 B per problem)
constexpr size_t kWsFusedBytes = 1024;               // FusedState of the one-launch observe + fake-quant (fused_step.h)

void set_error(const char* fmt, ...);
bool set_observer_tuning(const char* key, int value);   // observer.hip: knobs reached through osq_set_tuning

// Measurement aid (osq_time_next_launch): events that the next launch of kernel family `which` on this thread
// carries on its dispatch packet (hipExtLaunchKernelGGL); {nullptr, nullptr} = plain launch.
struct TimingHook {
    hipEvent_t start, stop;
};
TimingHook take_timing_hook(int which);

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define OSQ_REQUIRE(cond, msg)                       \
    do {                                             \
        if (!(cond)) {                               \
            ::osq::set_error("%s", msg);             \
            return OSQ_ERR_INVALID_ARGUMENT;         \
        }                                            \
    } while (0)

static inline int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return OSQ_ERR_HIP;
    }
    return OSQ_OK;
}

// Caller-owned scratch: [4 KiB of ticket counters][64 KiB scratch][16 KiB + 64 B wide-finaliser state][64 KiB rendezvous words]
// [1 KiB state of the fused observe + fake-quant launch].  Counters are zero between
// launches (each kernel's last workgroup resets the one it used).
struct Workspace {
    char* base;
    explicit Workspace(void* p) : base(static_cast<char*>(p)) {}
    unsigned int* counter(int) const { return reinterpret_cast<unsigned int*>(base); }
    double* doubles() const { return reinterpret_cast<double*>(base + kWsHeaderBytes); }
    float* floats() const { return reinterpret_cast<float*>(base + kWsHeaderBytes); }
    void* wide() const { return base + kWsHeaderBytes + kWsScratchBytes; }
    unsigned long long* meet() const {
        return reinterpret_cast<unsigned long long*>(base + kWsHeaderBytes + kWsScratchBytes + kWsWideBytes);
    }
    void* fused() const { return base + kWsHeaderBytes + kWsScratchBytes + kWsWideBytes + kWsMeetBytes; }
};

}  // namespace osq
