// One launch for a whole quantizer call in the calibrate-and-quantize state (both flags on, fake_quant.py:107-126 /
// 178-208) on a masked [B, T, H] activation: per-token extrema -> token-wise clipping (observer.py:50-70) ->
// running statistic -> calculate_qparams -> fake-quant.  Included by observer.hip after token_select.h.
//
// Why one launch: the scale depends on a reduction over the whole tensor, so a launch-per-stage version reads x
// twice (12 B/elem).  An MI355X holds 128 MiB in its vector register files; a [256,128,768] fp32 tensor is 96 MiB.
// This kernel is a persistent grid of one 1024-thread workgroup per CU in which every wave KEEPS the tokens it
// reduced in registers (24 float4 per lane = 96 VGPRs) until the scale is known, then quantises them from the
// registers: x is read from HBM once, y is written once (8 B/elem of HBM traffic for 12 B/elem of algorithmic
// work), and there are no kernel boundaries inside the call.
//
//   workgroups 0, 1   "selectors": wait until every streaming workgroup has published its per-token extrema, run
//                     select_side (token_select.h) on one side each, meet, and the second one applies the clip rule,
//                     the running statistic and calculate_qparams, writes the module's buffers and PUBLISHES
//                     (scale, zero_point) as two tagged 8-byte granules;
//   workgroups 2..    "streaming": wave w owns tokens j = w, w + W, w + 2W, ... of an enumeration that lists the valid
//                     tokens first (sample-major) and the padded ones after them.  Phase A1 loads its valid tokens
//                     and reduces them; the extrema go to the compact arrays (slot j) with write-through stores, the
//                     workgroup takes an arrival ticket.  Phase A2 loads its padded tokens (they are quantised too:
//                     the reference fake-quantises the whole tensor) while the selectors work.  Phase C: poll the
//                     granules, quantise the held registers, stream y out.  Tokens beyond the register capacity
//                     (24 float4 per lane) are streamed instead: reduced in A1 without being kept, re-read in C.
//
// Hand-offs follow cdna_hip_programming.md G16: payload written with agent-scope (sc1, write-through) stores and
// drained with s_waitcnt vmcnt(0) before the ticket; consumers read it with sc1 loads; every polled word is an
// agent-scope atomic.  No per-launch salt argument (it would be frozen under graph replay): the tag is epoch + 1
// with the epoch counter living in the workspace, read by every workgroup at its start and advanced by the
// finishing selector -- the only launch-to-launch state, and the kernel boundary orders it.  All workgroups must
// be resident together (they spin on each other): the host sizes the grid to min(CUs, occupancy) and every spin
// is bounded -- on a timeout the workgroup raises `status`, writes NaN and leaves.
#pragma once

namespace osq {

constexpr int kFusedThreads = 1024;
constexpr int kFusedWaves = kFusedThreads / OSQ_WAVE;
constexpr int kFusedHoldRegs = 18;        // float4 of activation data a lane keeps in registers across the wait (72 VGPRs)
constexpr int kFusedHoldLds = 9;          // ... and in LDS (9 x 16 B x 1024 threads = 144 KiB of the CU's 160)
constexpr int kFusedMaxBatch = 1024;      // prefix sums of the lengths live in LDS, one entry per thread
constexpr int kFusedShards = 8;           // arrival counters, one per XCD (workgroup b runs on XCD b % 8)
constexpr unsigned int kFusedSpinLimit = 1u << 21;

struct FusedState {                       // lives in the caller's workspace; ALL-ZERO before the first launch (or after a reset)
    // Arrival counters, one 64-byte line each, in TWO sets: launch number e (the epoch word) counts in set e & 1 and
    // zeroes set (e + 1) & 1 at its end.  Nobody of launch e touches the other set, and every workgroup of launch
    // e - 1 has left by the kernel boundary -- also the ones that arrived after a time-out -- so a late arrival can
    // never be added to a counter a later launch compares with `==` (round 2 zeroed the set in use, which a workgroup
    // arriving after its selectors' time-out would then have left at 1 for every launch after it).
    unsigned int arrive[2][kFusedShards][16];
    unsigned int epoch, pad0[15];             // launches completed on this workspace
    unsigned long long side[16];              // side[0], side[1]: one granule per selector (tag30 << 34 | empty << 33 | bad << 32 | value bits), adjacent: one 16-byte poll reads both
    unsigned int status, pad3[15];            // sticky: 1 = a selector timed out, 2 = a streaming workgroup timed out
    unsigned int go[2][16];                   // = tag once selector `side` has pulled the per-token extrema into its registers
};

struct FusedArgs {
    const float* x;
    float* y;
    int64_t B, T;                 // rows r = b*T + t at x + r*H, H = 256*NV floats
    const int64_t* lengths;       // NULL: every token is valid
    float* tok_min;               // compact per-token extrema, capacity B*T rounded up to 4 floats each
    float* tok_max;
    int prune;
    float q;
    int shortcut;
    FusedState* st;
    const float* scale_p;         // the module's parameters (also fin.scale_out / fin.zp_out)
    const void* zp_p;
    int zp_type, mode;
    float g, qmin, qmax;
    int gate;                     // 1: padded tokens are loaded only after every workgroup has arrived
    int deal;                     // how a slot's tokens are dealt to the waves (see the kernel): 0 = 16 consecutive tokens per workgroup,
                                  // 2 = round the workgroups token by token, 1 = only the slot with the last valid tokens
    unsigned int spin_limit;      // bound of every cross-workgroup wait (kFusedSpinLimit; osq_set_tuning("fused_spin_limit") for tests)
};

__device__ __forceinline__ unsigned long long peek64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int peek32(const unsigned int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned long long side_granule(unsigned int tag, const SideResult& r) {
    return (static_cast<unsigned long long>(tag & 0x3fffffffu) << 34) | (r.empty ? (1ull << 33) : 0ull) | (r.bad ? (1ull << 32) : 0ull) |
           __float_as_uint(r.value);
}

// one selection per call site keeps the code size (and the build time) of the four kernel instantiations down
template <int R4>
__device__ __attribute__((noinline)) SideResult select_side_compact(const float* src, int side, int64_t cap, unsigned int n_valid,
                                                                   int prune, float q, int shortcut, SelShared& S, long long* stamps,
                                                                   unsigned int* loaded_flag, unsigned int loaded_tag) {
    return select_side<R4, true>(src, side, 1, cap, nullptr, n_valid, prune, q, shortcut, S, stamps, loaded_flag, loaded_tag);
}

// streaming workgroups whose arrival lands on shard s (blockIdx % 8 == s, blockIdx >= 2)
__device__ __forceinline__ unsigned int fused_members(unsigned int nblocks, unsigned int s) {
    const unsigned int all = nblocks > s ? (nblocks - 1u - s) / kFusedShards + 1u : 0u;
    return all - (s < 2u && nblocks > s ? 1u : 0u);
}

__device__ __forceinline__ float4 as_float4(const v4u32& w) {
    return make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
}
__device__ __forceinline__ v4u32 as_v4u32(const float4& f) {
    v4u32 w;
    w.x = __float_as_uint(f.x); w.y = __float_as_uint(f.y); w.z = __float_as_uint(f.z); w.w = __float_as_uint(f.w);
    return w;
}

// 16-byte streaming store at (wave-uniform row offset + lane offset).  The offset goes into the VGPR operand, NOT the
// SGPR soffset field: with a register soffset hipcc (ROCm 7.2) pads no wait states between a buffer_store_dwordx4 and
// a following VALU write of its data registers (LLVM's hazard table exempts that form), and on gfx950 the store then
// picks up the NEXT element's v_div_scale result in its first data register for the last four lanes of every row --
// measured here as y.x == scale in ~5 % of the rows, different ones on every run.
__device__ __forceinline__ void store_row16(const v4u32& data, __amdgpu_buffer_rsrc_t rsrc, unsigned int lane_off, unsigned int row_off) {
    // sc1 = write-through: nothing of the 96 MiB of y stays dirty in the L2s for the end of the launch to write back
    // (nt stores: 40.9 us per step, plain 41.5, sc1 or sc0 sc1 38.9, nt sc1 40.5; tools/ab_step.py on one box)
    __builtin_amdgcn_raw_buffer_store_b128(data, rsrc, lane_off + row_off, 0, 16 /* sc1 */);
}

// development aid (-DOSQ_FINAL_TIMING, `make dbg`): thread 0 of every workgroup stamps the 100 MHz wall clock
#ifdef OSQ_FINAL_TIMING
#define OSQ_FSTAMP(k) do { if (threadIdx.x == 0 && g_osq_dbg) g_osq_dbg[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define OSQ_FSTAMP(k) do { } while (0)
#endif

template <int NV>
__global__ __launch_bounds__(kFusedThreads) void observe_fq_fused_kernel(FusedArgs a, Finish fin) {
    constexpr int SR = kFusedHoldRegs / NV;       // tokens a wave keeps in registers
    constexpr int SL = kFusedHoldLds / NV;        // ... and in LDS
    constexpr int S = SR + SL;
    constexpr int H4 = NV * OSQ_WAVE;             // float4 per token
    constexpr int CH = NV % 4 == 0 ? 4 : 3;       // float4 per lane and trip of the streamed (not held) tokens
    // the selectors' scratch and the streaming workgroups' token slots share the same LDS
    __shared__ union FusedLds {
        SelShared sel;
        v4u32 keep[kFusedWaves * kFusedHoldLds * OSQ_WAVE];
    } lds;
    SelShared& sel_lds = lds.sel;
    v4u32* const keep = lds.keep;
    __shared__ unsigned int pre[kFusedMaxBatch + 1];
    __shared__ unsigned int s_wtot[kFusedWaves];
    __shared__ unsigned int s_word[8];
    __shared__ unsigned int rows[kFusedWaves * 24];

    const int tid = threadIdx.x, lane = tid & (OSQ_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / OSQ_WAVE);      // wave-uniform, and the compiler knows it
    const unsigned int Bu = static_cast<unsigned int>(a.B), Tu = static_cast<unsigned int>(a.T);
    FusedState* st = a.st;

    OSQ_FSTAMP(0);
    // ---- everybody: epoch of this launch, prefix sums of the clamped lengths
    if (tid == 0) s_word[0] = peek32(&st->epoch);
    {
        unsigned int len = 0u;
        if (static_cast<unsigned int>(tid) < Bu) {
            int64_t l = a.lengths ? a.lengths[tid] : a.T;
            l = l < 0 ? 0 : (l > a.T ? a.T : l);
            len = static_cast<unsigned int>(l);
        }
        const unsigned int incl = wave_inclusive_scan_u32(len);
        if (lane == OSQ_WAVE - 1) s_wtot[wv] = incl;
        __syncthreads();
        unsigned int base = 0u;
#pragma unroll
        for (int k = 0; k < kFusedWaves; ++k) base += (k < wv) ? s_wtot[k] : 0u;
        if (tid == 0) pre[0] = 0u;
        if (static_cast<unsigned int>(tid) < Bu) pre[tid + 1] = base + incl;
        __syncthreads();
    }
    OSQ_FSTAMP(7);
    const unsigned int tag = s_word[0] + 1u;
    const unsigned int V = pre[Bu];                       // valid tokens
    const unsigned int total = Bu * Tu;

    if (blockIdx.x < 2u) {
        // =========================================================== selector (one side)
        const int side = blockIdx.x;
        if (wv == 0) {                                     // lanes 0..7 watch one arrival counter each
            const unsigned int s = lane < kFusedShards ? lane : 0;
            const unsigned int want = fused_members(gridDim.x, s);
            bool ok = false;
            for (unsigned int spins = 0; spins < a.spin_limit; ++spins) {
                ok = peek32(&st->arrive[(tag - 1u) & 1u][s][0]) == want;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (lane == 0) s_word[1] = __all(ok) ? 1u : 0u;
        }
        __syncthreads();
        OSQ_FSTAMP(1);
        const bool arrived = s_word[1] != 0u;
        SideResult r{__builtin_nanf(""), true, false};     // timed out: poison the call instead of hanging
#ifdef OSQ_FINAL_TIMING
        long long* sstamps = g_osq_dbg ? g_osq_dbg + 8 * 256 + 16 * side : nullptr;
#else
        long long* sstamps = nullptr;
#endif
        if (arrived) {
            const float* src = side ? a.tok_min : a.tok_max;
            const int64_t cap = (static_cast<int64_t>(total) + 3) & ~int64_t(3);
            const unsigned int g4 = (V + 3u) >> 2;         // 16-byte groups that hold valid slots
            unsigned int* const go = &st->go[side][0];
#define OSQ_FUSED_SELECT(R4) r = select_side_compact<R4>(src, side, cap, V, a.prune, a.q, a.shortcut, sel_lds, sstamps, go, tag)
            if (g4 <= 1u * kSelThreads) OSQ_FUSED_SELECT(1);
            else if (g4 <= 2u * kSelThreads) OSQ_FUSED_SELECT(2);
            else if (g4 <= 3u * kSelThreads) OSQ_FUSED_SELECT(3);
            else if (g4 <= 4u * kSelThreads) OSQ_FUSED_SELECT(4);
            else if (g4 <= 5u * kSelThreads) OSQ_FUSED_SELECT(5);
            else if (g4 <= 6u * kSelThreads) OSQ_FUSED_SELECT(6);
            else OSQ_FUSED_SELECT(8);
#undef OSQ_FUSED_SELECT
        } else if (tid == 0) {
            __hip_atomic_fetch_or(&st->status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&st->go[side][0], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        OSQ_FSTAMP(2);
        if (tid == 0)
            __hip_atomic_store(&st->side[side], side_granule(tag, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        OSQ_FSTAMP(3);
        return;
    }

    // =============================================================== streaming workgroup
    const unsigned int nwv = (gridDim.x - 2u) * kFusedWaves;                        // streaming waves
    // what the finishing arithmetic needs, fetched now by the thread that will do it
    float st_min = 0.f, st_max = 0.f, old_s = 0.f, old_z = 0.f;
    const bool have_state = fin.rule != OSQ_UPDATE_NONE && fin.min_val && fin.max_val;
    if (tid == 0) {
        if (have_state) { st_min = fin.min_val[0]; st_max = fin.max_val[0]; }
        old_s = a.scale_p[0];
        old_z = load_zp(a.zp_p, a.zp_type);
    }
    const unsigned int bp = blockIdx.x - 2u;                                         // this workgroup's number among the streaming ones
    const unsigned int gw = bp * kFusedWaves + static_cast<unsigned int>(wv);
    // How the W tokens of a slot are dealt to the waves (a.deal, osq_set_tuning("fused_deal")).
    //   0: workgroup bp takes 16 consecutive tokens.  The same wave of neighbouring workgroups -- they issue their loads and
    //      stores at about the same time -- is then 16 rows apart, a multiple of 16 KiB at every feature count the kernel
    //      takes: simultaneous requests crowd a few memory channels.  And V is rarely a multiple of W: the slot that holds
    //      the last valid tokens gives the first (V % W) / 16 workgroups one valid token per wave more than the others
    //      (5 against 4 at the bench lengths), and the selection waits for those.
    //   2 (default): wave w of workgroup bp takes token w * G + bp (G streaming workgroups): simultaneous requests are one
    //      row apart, every workgroup has the same share of a slot's valid tokens.  Measured on one box, graph replay of
    //      the bench step: 44.3 -> 41.5 us ([256,128,768], bench lengths), 55.0 -> 52.8 (all valid), 35.4 -> 33.0
    //      ([32,128,3072]), 21.7 -> 20.0 ([64,128,1024]), 43.6 -> 41.1 ([32,128,4096]).  Two, four tokens per workgroup or
    //      the workgroups of one XCD on consecutive tokens measure the same as one; numbering the workgroups in the order
    //      they start (a ticket: XCDs receive a launch up to 2 us apart) costs the ticket's round trip: +1.5 us.
    //   1: only the slot that holds the last valid tokens is dealt like 2 (the balance without the channel spreading: -0.3 us).
    const unsigned int nwg = gridDim.x - 2u;
    const unsigned int gwi = static_cast<unsigned int>(wv) * nwg + bp;
    const unsigned int kdeal = a.deal >= 2 ? 0xffffffffu : (a.deal == 1 ? V / nwv : 0xfffffffeu);   // slot(s) dealt token by token
    // 16 KiB rows (4096 features): workgroup bp starts a token at piece bp % 16, otherwise the workgroups' simultaneous
    // requests are again whole multiples of 16 KiB apart (-1 us of 41 on [32,128,4096]; no effect at 768 / 1024 / 3072)
    const unsigned int prot = NV == 16 ? bp % 16u : 0u;
#define OSQ_FUSED_PIECE(u) ((((static_cast<unsigned int>(u) + prot) >= static_cast<unsigned int>(NV)) ? static_cast<unsigned int>(u) + prot - NV : static_cast<unsigned int>(u) + prot) * 1024u)
#define OSQ_FUSED_TOKEN(k) (static_cast<unsigned int>(k) * nwv + ((kdeal == 0xffffffffu || kdeal == static_cast<unsigned int>(k)) ? gwi : gw))
    // Rows are addressed as buffer base (SGPR descriptor) + wave-uniform row offset (SGPR) + lane * 16 (one VGPR for
    // every access): no 64-bit address pair per token in flight -- the lanes' registers are for data.
    const unsigned int tensor_bytes = total * static_cast<unsigned int>(H4 * 16);      // < 4 GiB, checked by the launcher
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, static_cast<int>(tensor_bytes), 0x00020000);
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, static_cast<int>(tensor_bytes), 0x00020000);
    const unsigned int lane_off = static_cast<unsigned int>(lane) * 16u;
    constexpr unsigned int kRowBytes = H4 * 16u;
    constexpr int kNt = 2;                                // aux: non-temporal (streamed once).  Plain, sc0 or sc1 loads instead: 50 us per step against 38.5
    // this wave's LDS slots: float4 (slot s, step u, lane) at keep[((wv * SL + s) * NV + u) * 64 + lane]
    v4u32* const keep_w = keep + static_cast<unsigned int>(wv) * (SL * NV * OSQ_WAVE) + lane;

    // token j of the enumeration -> row b*T + t.  Valid tokens of sample b are j in [pre[b], pre[b+1]); padded ones
    // follow: j - V in [b*T - pre[b], (b+1)*T - pre[b+1]).  b = number of prefix entries (i = 1..B) at or below j:
    // thread (k, w) of the first 16*S threads finds it for slot k of wave w by bisection and leaves the row in LDS.
    // (All sixteen waves counting with ballots, 64 entries at a time, took 3 us: the CU's issue slots, not latency.)
    if (tid < kFusedWaves * S) {
        const unsigned int k = static_cast<unsigned int>(tid) / kFusedWaves, w = static_cast<unsigned int>(tid) % kFusedWaves;
        const unsigned int j = k * nwv + ((kdeal == 0xffffffffu || kdeal == k) ? w * nwg + bp : bp * kFusedWaves + w);
        unsigned int r = 0u;
        if (j < total) {
            const bool valid = j < V;
            const unsigned int key = valid ? j : j - V;
            unsigned int lo = 0u, hi = Bu;                 // invariant f(lo) <= key < f(hi), f(i) = tokens of this kind before sample i
            while (lo + 1u < hi) {
                const unsigned int mid = (lo + hi) >> 1;   // f(mid) <= key  ->  b >= mid
                const unsigned int f = valid ? pre[mid] : mid * Tu - pre[mid];
                if (f <= key) lo = mid; else hi = mid;
            }
            const unsigned int b = lo;                     // f(b) <= key < f(b + 1); f is monotone, so b is the count
            r = valid ? b * Tu + (key - pre[b]) : b * Tu + (pre[b + 1u] - pre[b]) + (key - (b * Tu - pre[b]));
        }
        rows[tid] = r;
    }
    __syncthreads();
    unsigned int row[S];
#pragma unroll
    for (int k = 0; k < S; ++k) row[k] = uniform(rows[k * kFusedWaves + wv]);
    OSQ_FSTAMP(1);

    // ---- phase A1: valid tokens -> registers (slots 0..SR-1) / LDS (slots SR..S-1), per-token extrema -> compact arrays
    v4u32 hold[SR > 0 ? SR : 1][NV];
    {
        v4u32 tmp[SL > 0 ? SL : 1][NV];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const unsigned int j = OSQ_FUSED_TOKEN(k);
            if (j < V) {
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const v4u32 w = __builtin_amdgcn_raw_buffer_load_b128(xrs, lane_off, row[k] * kRowBytes + OSQ_FUSED_PIECE(u), kNt);
                    if (k < SR) hold[k][u] = w; else tmp[k - SR][u] = w;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const unsigned int j = OSQ_FUSED_TOKEN(k);
            if (j < V) {
                MinMax acc;
                acc.init();
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const v4u32 w = k < SR ? hold[k][u] : tmp[k - SR][u];
                    acc.add4(as_float4(w));
                    if (k >= SR) keep_w[((k - SR) * NV + u) * OSQ_WAVE] = w;
                }
                acc.wave_reduce();
                acc.poison();
                if (lane == 0) {
                    publish_f32(&a.tok_min[j], acc.mn);
                    publish_f32(&a.tok_max[j], acc.mx);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                // one token's temporaries at a time: the registers hold data
        }
    }
    // valid tokens beyond the capacity: reduced now, read again in phase C
    for (unsigned int j = gw + static_cast<unsigned int>(S) * nwv; j < V; j += nwv) {
        unsigned int b = 0u;
        for (unsigned int c = 0; c < Bu; c += OSQ_WAVE) {
            const unsigned int i = c + static_cast<unsigned int>(lane) + 1u;
            b += static_cast<unsigned int>(__builtin_popcountll(__ballot(i <= Bu && pre[i <= Bu ? i : Bu] <= j)));
        }
        const unsigned int soff = uniform(b * Tu + (j - pre[b])) * kRowBytes;
        MinMax acc;
        acc.init();
#pragma unroll
        for (int u0 = 0; u0 < NV; u0 += CH) {             // CH float4 in flight per lane: the held tokens keep their registers
            v4u32 v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(xrs, lane_off, soff + (u0 + u) * 1024u, kNt);
#pragma unroll
            for (int u = 0; u < CH; ++u) acc.add4(as_float4(v[u]));
        }
        acc.wave_reduce();
        acc.poison();
        if (lane == 0) {
            publish_f32(&a.tok_min[j], acc.mn);
            publish_f32(&a.tok_max[j], acc.mx);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every publishing wave: its extrema have left this CU
    OSQ_FSTAMP(2);
    __syncthreads();
    if (tid == 0)
        __hip_atomic_fetch_add(&st->arrive[(tag - 1u) & 1u][blockIdx.x % kFusedShards][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    OSQ_FSTAMP(3);

    // ---- phase A2: this wave's padded tokens -> registers / LDS, while the selectors work.  Not before every workgroup
    // has arrived and the selectors hold the extrema (they raise `go`): padded loads issued earlier take bandwidth from the workgroups still in
    // A1 and from the selectors' own loads, and both are what the scale waits for.
    // a.gate: 0 = nothing waits, 1 = every padded load waits, 2 = the register slots' padded tokens are loaded at once
    // (they fill the memory system while the last workgroups finish A1), the LDS slots' after `go`
    if (a.gate != 1) {
#pragma unroll
        for (int k = 0; k < SR; ++k) {
            const unsigned int j = OSQ_FUSED_TOKEN(k);
            if (j >= V && j < total) {
#pragma unroll
                for (int u = 0; u < NV; ++u)
                    hold[k][u] = __builtin_amdgcn_raw_buffer_load_b128(xrs, lane_off, row[k] * kRowBytes + OSQ_FUSED_PIECE(u), kNt);
            }
        }
        asm volatile("" ::: "memory");
    }
    if (a.gate) {
        if (tid == 0) {
            for (unsigned int spins = 0; spins < a.spin_limit; ++spins) {
                if (peek32(&st->go[0][0]) == tag && peek32(&st->go[1][0]) == tag) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    }
    if (a.gate == 1) {
#pragma unroll
        for (int k = 0; k < SR; ++k) {
            const unsigned int j = OSQ_FUSED_TOKEN(k);
            if (j >= V && j < total) {
#pragma unroll
                for (int u = 0; u < NV; ++u)
                    hold[k][u] = __builtin_amdgcn_raw_buffer_load_b128(xrs, lane_off, row[k] * kRowBytes + OSQ_FUSED_PIECE(u), kNt);
            }
        }
    }
#pragma unroll
    for (int k = SR; k < S; ++k) {
        const unsigned int j = OSQ_FUSED_TOKEN(k);
        if (j >= V && j < total) {
            v4u32 w[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) w[u] = __builtin_amdgcn_raw_buffer_load_b128(xrs, lane_off, row[k] * kRowBytes + OSQ_FUSED_PIECE(u), kNt);
#pragma unroll
            for (int u = 0; u < NV; ++u) keep_w[((k - SR) * NV + u) * OSQ_WAVE] = w[u];
        }
    }
    asm volatile("" ::: "memory");

    // ---- wait for the two sides; every workgroup finishes the statistic itself (clip rule, running statistic,
    // calculate_qparams: a few dozen scalar operations) instead of waiting for one finisher to do it and publish again
    if (tid == 0) {
        unsigned long long g0 = 0ull, g1 = 0ull;
        bool ok = false;
        const unsigned long long want = static_cast<unsigned long long>(tag & 0x3fffffffu);
        const auto srs = __builtin_amdgcn_make_buffer_rsrc(st->side, 0, 16, 0x00020000);
        for (unsigned int spins = 0; spins < a.spin_limit; ++spins) {
            const v4u32 w = __builtin_amdgcn_raw_buffer_load_b128(srs, 0, 0, 16);      // sc1: both granules, one request
            g0 = (static_cast<unsigned long long>(w.y) << 32) | w.x;
            g1 = (static_cast<unsigned long long>(w.w) << 32) | w.z;
            ok = (g0 >> 34) == want && (g1 >> 34) == want;
            if (ok) break;
            __builtin_amdgcn_s_sleep(1);
        }
        float s = __builtin_nanf(""), z = s;
        if (!ok) {
            __hip_atomic_fetch_or(&st->status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if ((g0 >> 33) & 1ull) {                    // nothing observed (both sides agree): parameters as they are
            s = old_s;
            z = old_z;
        } else {
            const float up = __uint_as_float(static_cast<unsigned int>(g0)), lo = -__uint_as_float(static_cast<unsigned int>(g1));
            float cur_min = (lo > up) ? up : lo;           // aminmax(clip(value, lo, up)), observer.py:68,227
            float cur_max = up;
            if (((g0 | g1) >> 32) & 1ull) { cur_min = __builtin_nanf(""); cur_max = cur_min; }
            float mn = cur_min, mx = cur_max;
            if (have_state) {
                mn = st_min;
                mx = st_max;
                apply_update(fin.rule, fin.cnt, cur_min, cur_max, &mn, &mx);
            }
            qparams_from_range(mn, mx, fin.quant_min, fin.quant_max, fin.symmetric, &s, &z);
            if (blockIdx.x == 2u) {                        // ONE workgroup writes the module's buffers
                if (have_state) { fin.min_val[0] = mn; fin.max_val[0] = mx; }
                fin.scale_out[0] = s;
                store_zp(fin.zp_out, fin.zp_type, 0, z);
            }
            if (fin.zp_type != OSQ_ZP_FLOAT32) z = static_cast<float>(static_cast<int32_t>(z));   // what a reader of the int32 buffer sees
        }
        if (blockIdx.x == 2u) {                            // ... and closes the launch's bookkeeping: the NEXT launch's counter set
            for (int k = 0; k < kFusedShards; ++k)         // (untouched by this launch) is zeroed, the epoch moves on -- every workgroup
                __hip_atomic_store(&st->arrive[tag & 1u][k][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read it at its start
            __hip_atomic_store(&st->epoch, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_word[2] = __float_as_uint(s);
        s_word[3] = __float_as_uint(z);
        OSQ_FSTAMP(4);
    }
    __syncthreads();
    OSQ_FSTAMP(5);
    const QParams p = effective_params(__uint_as_float(s_word[2]), __uint_as_float(s_word[3]), a.mode & OSQ_PARAM_MODE_MASK, a.g);
    const float qs = p.scale, qz = p.zp;

    // ---- phase C: quantise from the registers / LDS
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const unsigned int j = OSQ_FUSED_TOKEN(k);
        if (j < total) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const v4u32 w = k < SR ? hold[k < SR ? k : 0][u] : keep_w[((k - SR) * NV + u) * OSQ_WAVE];
                float4 o, q;
                fq4_plain(as_float4(w), o, q, qs, qz, a.qmin, a.qmax);
                store_row16(as_v4u32(o), yrs, lane_off, row[k] * kRowBytes + OSQ_FUSED_PIECE(u));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // tokens beyond the capacity: stream
    for (unsigned int j = gw + static_cast<unsigned int>(S) * nwv; j < total; j += nwv) {
        unsigned int b = 0u, r;
        if (j < V) {
            for (unsigned int c = 0; c < Bu; c += OSQ_WAVE) {
                const unsigned int i = c + static_cast<unsigned int>(lane) + 1u;
                b += static_cast<unsigned int>(__builtin_popcountll(__ballot(i <= Bu && pre[i <= Bu ? i : Bu] <= j)));
            }
            r = b * Tu + (j - pre[b]);
        } else {
            for (unsigned int c = 0; c < Bu; c += OSQ_WAVE) {
                const unsigned int i = c + static_cast<unsigned int>(lane) + 1u;
                b += static_cast<unsigned int>(__builtin_popcountll(__ballot(i <= Bu && i * Tu - pre[i <= Bu ? i : Bu] <= j - V)));
            }
            r = b * Tu + (pre[b + 1u] - pre[b]) + ((j - V) - (b * Tu - pre[b]));
        }
        const unsigned int soff = uniform(r) * kRowBytes;
#pragma unroll
        for (int u0 = 0; u0 < NV; u0 += CH) {
            v4u32 v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(xrs, lane_off, soff + (u0 + u) * 1024u, kNt);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                float4 o, q;
                fq4_plain(as_float4(v[u]), o, q, qs, qz, a.qmin, a.qmax);
                store_row16(as_v4u32(o), yrs, lane_off, soff + (u0 + u) * 1024u);
            }
        }
    }
    OSQ_FSTAMP(6);
#undef OSQ_FUSED_TOKEN
#undef OSQ_FUSED_PIECE
}

}  // namespace osq
