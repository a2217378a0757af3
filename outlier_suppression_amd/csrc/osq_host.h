// Host-side helpers of the C ABI: argument checks, error text, launch checks,
// workspace carving.  No torch types anywhere below the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/osq_hip.h"

namespace osq {

constexpr int kMaxBlocks = 2048;          // 256 CUs x 8 workgroups of 256 threads
// Every kernel family that finishes in its last workgroup owns a block of ticket counters and a block of partials:
// two such kernels of DIFFERENT families may overlap on one stream (graph branches) without sharing either.
enum WsFamily { kFamLsqBackward = 0, kFamObserveFlat = 1, kFamWideFinal = 2, kFamMseFlat = 3, kFamMseTokens = 4,
                kFamMoments = 5, kFamHistogram = 6, kWsFamilies = 8 };
constexpr size_t kWsCounterBlock = 4096;                      // 33 counters used, one 64-byte line each
constexpr size_t kWsHeaderBytes = kWsFamilies * kWsCounterBlock;
constexpr size_t kWsScratchPerFamily = 64 * 1024;             // 2048 partials of 8 B + extras, or 256 x 32 partials (MSE grid)
constexpr size_t kWsScratchBytes = kWsFamilies * kWsScratchPerFamily;
constexpr size_t kWsWideBytes = 2 * 2048 * 4 + 64;   // WideState of the multi-workgroup token finaliser
constexpr size_t kWsMeetBytes = 64 * 1024;           // rendezvous words of token_select_kernel (8 B per problem)
constexpr size_t kWsFusedBytes = 2048;               // FusedState of the one-launch observe + fake-quant (fused_step.h)
constexpr int kResidentMaxBlocks = 512;              // workgroups of the resident MSEFast search (msefast.hip)
constexpr int kResidentMaxSites = 16;                // searches one multi-site launch can hold (one polling wave each)
constexpr size_t kWsResidentBytes = 128 + 2 * static_cast<size_t>(kResidentMaxSites) * kResidentMaxBlocks * 2 * 8;   // its epoch / status words + two buffers of partial-sum granules per site


// Process-wide state behind osq_set_tuning comes in two kinds:
//   OSQ_SWITCH   what SHIPS: switches that select a summation order, a path (persistent launch or not, which finaliser) or
//                the bound of a cross-workgroup wait.  std::atomic, relaxed: a thread that flips one while another launches
//                is a race on a word, not undefined behaviour; every launch reads each switch once.
//   OSQ_AB_KNOB  performance A/B knobs (grid caps, unroll factors, cache hints, pipeline depths).  The measured winners are
//                compile-time constants in the release library; only a -DOSQ_TUNABLE build (`make dbg`, libosq_hip_dbg.so)
//                keeps them as variables and accepts their keys -- osq_build_flags() tells the two apart.
#define OSQ_SWITCH(type, name, value) static std::atomic<type> name{value}
#ifdef OSQ_TUNABLE
#define OSQ_AB_KNOB(type, name, value) static type name = value
#else
#define OSQ_AB_KNOB(type, name, value) static constexpr type name = value
#endif

void set_error(const char* fmt, ...);
bool set_observer_tuning(const char* key, int value);   // observer.hip: knobs reached through osq_set_tuning
bool set_msefast_tuning(const char* key, int value);    // msefast.hip
bool set_layernorm_tuning(const char* key, int value);  // layernorm.hip
bool set_extra_tuning(const char* key, int value);      // observers_extra.hip
bool stream_write_through();                            // fake_quant.hip: osq_set_tuning("stream_wt", 0|1)

// Measurement aid (osq_time_next_launch): events that the next launch of kernel family `which` on this thread
// carries on its dispatch packet (hipExtLaunchKernelGGL); {nullptr, nullptr} = plain launch.
struct TimingHook {
    hipEvent_t start, stop;
};
TimingHook take_timing_hook(int which);

// Persistent launches (workgroups that spin on each other: the fused observe + fake-quant step, the resident MSEFast
// search).  persistent_grid_for: workgroups of `threads` threads that are resident together (one per CU), 0 = do not
// launch.  persistent_serialize: two such grids must never be in flight together on one device -- when the stream
// changes, the new stream first waits for what the previous one was given (observer.hip).
int persistent_grid_for(const void* kernel, int threads);
bool persistent_serialize(hipStream_t st);

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

// Caller-owned scratch: [8 x 4 KiB of ticket counters][8 x 64 KiB of partials][16 KiB + 64 B wide-finaliser state]
// [64 KiB rendezvous words][2 KiB state of the fused observe + fake-quant launch][256.1 KiB state of the resident MSEFast search].  Counters are zero between
// launches (each kernel's last workgroup resets the one it used).
struct Workspace {
    char* base;
    explicit Workspace(void* p) : base(static_cast<char*>(p)) {}
    unsigned int* counter(int family) const { return reinterpret_cast<unsigned int*>(base + family * kWsCounterBlock); }
    double* doubles(int family) const { return reinterpret_cast<double*>(base + kWsHeaderBytes + family * kWsScratchPerFamily); }
    float* floats(int family) const { return reinterpret_cast<float*>(base + kWsHeaderBytes + family * kWsScratchPerFamily); }
    void* wide() const { return base + kWsHeaderBytes + kWsScratchBytes; }
    unsigned long long* meet() const {
        return reinterpret_cast<unsigned long long*>(base + kWsHeaderBytes + kWsScratchBytes + kWsWideBytes);
    }
    void* fused() const { return base + kWsHeaderBytes + kWsScratchBytes + kWsWideBytes + kWsMeetBytes; }
    void* resident() const { return base + kWsHeaderBytes + kWsScratchBytes + kWsWideBytes + kWsMeetBytes + kWsFusedBytes; }
};

}  // namespace osq
