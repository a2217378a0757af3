// Per-tensor MSEFast searches in the REFERENCE's summation order with the sites RESIDENT on the chip (round 5).
// Included by msefast.hip after ordered_evaluation (uses TensorSearch, sq_err_f64_lean / sq_err_f64, lean_level_exact,
// rcp_division_exact, loss_qparams, aten_order.h).
//
// The rounds of msefast_tensor_ordered_multi_kernel re-stream every unfinished site from HBM for every loss evaluation
// (571 MB x ~600 rounds per BERT-base batch, 146 us per round: 58.7 % of configs[3]'s GPU time; every way of streaming
// faster was measured in round 5 and moved it by <= 6 %, LABNOTES "Round 5").  But the evaluations of a search read the
// SAME tensor ~600 times, and an MI355X holds 128 MiB in its register files.  Here a persistent grid (one 512-thread
// workgroup per CU, 256 VGPRs per lane) keeps KI chunk groups per workgroup in registers for the whole search:
//
//   item      one step of the cascade's stage 1 (aten_order.h): G = 2 level-1 chunks of a float64 sum = 8192 elements, 16
//             rows per lane; or a site's OPEN unit (the tail behind its full chunks: read from memory, a few KB);
//   site      a search.  Its items are dealt round the workgroups.  Per evaluation ("round r of the site") every item's
//             workgroup computes the item's chunk sums from its registers -- the same additions in the same order as
//             cascade_chunks_pipelined -- publishes them to the site's scratch (write-through), drains, and takes ONE
//             ticket on the site's monotonic arrival counter.  The workgroup that completes round r (ticket ==
//             (r + 1) * items) is the round's MASTER: cascade_finish (upper levels, in order), the Brent step
//             (Search::tell), the next candidate's parameters, a 16-byte record + the site's round word (payload ->
//             vmcnt(0) -> flag).
//   sweep     a workgroup polls the round words of its items' sites with one vector load, fetches the records of those that
//             moved, evaluates every ready item, then arrives.  Sites advance independently: no grid-wide barrier, and a
//             workgroup that waits for one site works on its other items.
//
// One launch runs its searches to completion: no kernel boundary and no HBM stream per evaluation, only the chain
// "chunk sums -> master -> record -> poll" (two hand-offs through memory, ~10 us) per round of a site, all sites of
// the launch in parallel.  What does not fit (fp32 calls, sites whose cascade step is 32, more items than the grid holds)
// stays on the streaming rounds.  Same terms, same additions, same order as ordered_evaluation: same bits
// (tests/test_gpu_strict_order.py).
#pragma once
#include <type_traits>
#include <utility>

namespace osq {

constexpr int kRoThreads = 512;
constexpr int kRoMaxSites = 128;
constexpr unsigned int kRoSpinLimit = 1u << 22;
constexpr unsigned int kRoNoSite = 0xffffu;

struct RoSite {                    // device table entry, 64 bytes
    const float* x;                // flat fp32 input, n elements (a masked site: gathered, remove_padding order)
    TensorSearch* ts;
    double* part;                  // cascade scratch, cascade_scratch_bytes(n, W / 2, 1, 8)
    int64_t n;
    unsigned int items;            // arrivals per round: the site's chunk groups (its open unit is the master's)
    unsigned int pad[7];
};
static_assert(sizeof(RoSite) == 64, "RoSite is a 64-byte table entry");

struct alignas(64) RoSync {        // per site; ALL-ZERO at launch
    unsigned int arrive;           // monotonic ticket counter: (r + 1) * items arrivals complete round r
    unsigned int pad0[15];
    unsigned int round;            // rounds completed = index of the pending candidate (candidate 0 is what _begin left in ts)
    unsigned int done;             // the pending record says: converged, nothing to evaluate
    float zp;
    unsigned int pad1;
    double scale_d;                // parameters of the pending candidate (rounds >= 1; round 0 reads ts)
    unsigned int pad2[10];
};
static_assert(sizeof(RoSync) == 128, "RoSync is two 64-byte lines: tickets apart from the record");

struct RoItem {                    // 8 bytes
    unsigned short site;           // kRoNoSite: empty slot
    unsigned short kind;           // 0: chunk group q (resident)
    unsigned int q;
};

struct RoArgs {
    const RoSite* sites;
    RoSync* sync;
    const RoItem* items;           // [gridDim.x][KI]
    unsigned int* status;          // sticky: 1 = a workgroup gave up waiting
    int n_sites, W;                // W: fp32 SIMD lanes of the reference host (8 | 16); float64 sums use W / 2
    int lean_ok, n_masters;        // workgroups 0 .. n_masters - 1 are masters (no items), the others workers
    unsigned int spin_limit;
};

__device__ __forceinline__ unsigned int ro_peek(const unsigned int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int ro_uniform(unsigned int v) {
    return static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
}
__device__ __forceinline__ void ro_put(unsigned int* p, unsigned int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): an unrolled loop whose index is a compile-time constant
// whatever its size (hipcc gives up on "#pragma unroll" when the body is large, and a register array indexed by a run-time
// index goes to scratch memory)
template <typename F, int... Is>
__device__ __forceinline__ void ro_static_for(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}

#ifdef OSQ_RO_TIMING
#define OSQ_RO_T(var) const long long var = wall_clock64()
#define OSQ_RO_ACC(slot, d) atomicAdd(reinterpret_cast<unsigned long long*>(a.status) + 1 + (slot), static_cast<unsigned long long>(d))
#else
#define OSQ_RO_T(var) do { } while (0)
#define OSQ_RO_ACC(slot, d) do { } while (0)
#endif

template <int KI>
__global__ __launch_bounds__(kRoThreads) void msefast_ordered_resident_kernel(RoArgs a) {
    constexpr int P = 4, S = 16, R = 16;
    __shared__ double lds_raw[kOrdLdsBytes / 8];
    __shared__ RoItem s_item[KI];
    __shared__ unsigned int s_my[KI], s_ready[KI], s_done[KI];
    __shared__ float s_zp[KI];
    __shared__ double s_scale[KI];
    __shared__ unsigned int s_flag;
    // per-item constants, fetched ONCE (a sweep must not walk site table -> search state -> fields for every item: three
    // dependent loads per item and sweep were most of a round's latency)
    __shared__ double* s_part[KI];
    __shared__ long long s_chunks[KI];
    __shared__ float s_qmin[KI], s_qmax[KI];
    __shared__ unsigned int s_finite[KI];
    const int tid = threadIdx.x;
    const int W = a.W / 2, NC = 4 * W;                     // float64 sums: half the lanes
    const int nc_shift = __builtin_ctz(static_cast<unsigned int>(NC));

    if (static_cast<int>(blockIdx.x) < a.n_masters) {
        // =========================================================== MASTER workgroup: sites blockIdx.x, + n_masters, ...
        // No resident data, all its registers for cascade_finish and the Brent step.  It watches its sites' arrival counters;
        // a site whose round is complete gets its upper levels added in order, its Brent step, its next candidate published.
        // its sites' constants and the pending candidate live in LDS: a poll is ONE load (the arrival counter)
        constexpr int kMine = (kRoMaxSites + 15) / 16;
        __shared__ RoSite m_site[kMine];
        __shared__ unsigned int m_rounds[kMine], m_done[kMine], m_fin[kMine];
        __shared__ float m_qmin[kMine], m_qmax[kMine], m_z[kMine];
        __shared__ double m_sd[kMine];
        int mine = 0;
        for (int sidx = blockIdx.x; sidx < a.n_sites && mine < kMine; sidx += a.n_masters) ++mine;
        if (tid < mine) {
            const RoSite st = a.sites[blockIdx.x + tid * a.n_masters];
            m_site[tid] = st;
            m_rounds[tid] = 0u;
            m_done[tid] = st.ts->S.done ? 1u : 0u;
            m_qmin[tid] = static_cast<float>(st.ts->S.quant_min);
            m_qmax[tid] = static_cast<float>(st.ts->S.quant_max);
            m_fin[tid] = (fabs(st.ts->S.x_min) <= 3.5e38 && fabs(st.ts->S.x_max) <= 3.5e38) ? 1u : 0u;
            m_sd[tid] = st.ts->scale_d;
            m_z[tid] = st.ts->zp;
        }
        __syncthreads();
        unsigned int spins = 0u;
        for (;;) {
            unsigned int open_sites = 0u, progressed = 0u;
            for (int k = 0; k < mine; ++k) {
                if (ro_uniform(m_done[k])) continue;
                ++open_sites;
                const int sidx = blockIdx.x + k * a.n_masters;
                RoSync* sy = a.sync + sidx;
                if (tid == 0) s_flag = ro_peek(&sy->arrive) == (m_rounds[k] + 1u) * m_site[k].items ? 1u : 0u;
                __syncthreads();
                const unsigned int f = ro_uniform(s_flag);
                __syncthreads();
                if (f == 0u) continue;
                ++progressed;
                OSQ_RO_T(t_seen);
                const RoSite st = m_site[k];
                TensorSearch* ts = st.ts;
                const double sd = m_sd[k];
                const float z = m_z[k];
                const CascadeGeom g = cascade_geom(st.n, W);
                const float qmin = m_qmin[k], qmax = m_qmax[k];
                const double rcp = 1.0 / sd;
                const bool lean = a.lean_ok && m_fin[k] && rcp_division_exact(sd, 0.0, 0.0) && lean_level_exact(z, qmin, qmax);
                const float rcp32 = static_cast<float>(rcp), lo32 = qmin - z, hi32 = qmax - z;
                const float* x = st.x;
                auto term = [=](int64_t e, double (&t)[1]) {
                    t[0] = lean ? sq_err_f64_lean(x[e], sd, rcp, rcp32, lo32, hi32, z, qmin, qmax) : sq_err_f64_outofline(x[e], sd, z, qmin, qmax);
                };
                // the site's OPEN unit (what lies behind its full chunks: less than a chunk group, from memory) is the master's:
                // the workers' registers hold full groups only
                cascade_units<double, 1, kRoThreads>(g, st.part, lds_raw, term, 0u, 1u, g.chunks);      // drains and synchronises at its end
                OSQ_RO_T(t_open);
                double sum[1];
                cascade_finish<double, 1, kRoThreads>(g, st.part, lds_raw, kOrdLdsBytes / 8, term, sum);
                OSQ_RO_T(t_fin);
                if (tid == 0) {
                    ts->S.tell(sum[0] / static_cast<double>(st.n));
                    OSQ_RO_T(t_tell);
                    if (!ts->S.done)
                        loss_qparams(ts->S.cand_min, ts->S.cand_max, ts->S.quant_min, ts->S.quant_max, ts->S.symmetric, &ts->scale, &ts->zp,
                                     &ts->scale_d);
                    m_done[k] = ts->S.done ? 1u : 0u;
                    m_sd[k] = ts->scale_d;
                    m_z[k] = ts->zp;
                    m_rounds[k] += 1u;
                    ro_put(&sy->done, ts->S.done ? 1u : 0u);
                    ro_put(reinterpret_cast<unsigned int*>(&sy->zp), __float_as_uint(ts->zp));
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&sy->scale_d),
                                       static_cast<unsigned long long>(__double_as_longlong(ts->scale_d)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the record has left before its flag
                    ro_put(&sy->round, m_rounds[k]);
#ifdef OSQ_RO_TIMING
                    const long long t_pub = wall_clock64();
                    if (sidx == 0) {
                        OSQ_RO_ACC(0, t_open - t_seen); OSQ_RO_ACC(1, t_fin - t_open); OSQ_RO_ACC(2, t_tell - t_fin); OSQ_RO_ACC(3, t_pub - t_tell);
                        OSQ_RO_ACC(4, 1);
                        static __shared__ long long t_last_pub;
                        if (m_rounds[k] > 1u) OSQ_RO_ACC(5, t_seen - t_last_pub);       // publish -> all arrivals seen: the workers' part of the round
                        t_last_pub = t_pub;
                    }
#endif
                }
                __syncthreads();
            }
            if (open_sites == 0u) break;
            if (progressed) spins = 0u;
            else {
                if (++spins > a.spin_limit) { if (tid == 0) __hip_atomic_fetch_or(a.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        return;
    }

    // =============================================================== WORKER workgroup
    const int tpc = S << nc_shift;                          // (block, column) pairs of a chunk
    const int G = kRoThreads / tpc;                         // chunks per item
    const int j = tid / tpc, rem = tid - j * tpc, blk = rem >> nc_shift, c = rem & (NC - 1);
    const unsigned int wg = blockIdx.x - static_cast<unsigned int>(a.n_masters);
    if (tid < KI) {
        const RoItem it = a.items[static_cast<size_t>(wg) * KI + tid];
        s_item[tid] = it;
        s_my[tid] = 0u;
        if (it.site != kRoNoSite) {
            const RoSite st = a.sites[it.site];
            s_part[tid] = st.part;
            s_chunks[tid] = cascade_geom(st.n, W).chunks;
            s_qmin[tid] = static_cast<float>(st.ts->S.quant_min);
            s_qmax[tid] = static_cast<float>(st.ts->S.quant_max);
            const double lo = st.ts->S.x_min, hi = st.ts->S.x_max;
            s_finite[tid] = (fabs(lo) <= 3.5e38 && fabs(hi) <= 3.5e38) ? 1u : 0u;       // rcp_division_exact's condition on the data
        }
    }
    __syncthreads();

    // ---- the resident values: item i, row k of this lane's (block, column)
    float r[KI][R];
    unsigned int fin = 0u;                                  // items with nothing (more) to do; uniform
    ro_static_for([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const RoItem it = s_item[i];
        if (it.site == kRoNoSite) { fin |= 1u << i; return; }
        {
            const RoSite st = a.sites[it.site];
            const CascadeGeom g = cascade_geom(st.n, W);
            const int64_t m = static_cast<int64_t>(it.q) * G + j;
            const bool have = m < g.chunks;
            const int64_t row0 = (((m << P) + blk) << P);
#pragma unroll
            for (int k = 0; k < R; ++k) r[i][k] = have ? st.x[((row0 + k) << nc_shift) + c] : 0.0f;
        }
    }, std::make_integer_sequence<int, KI>{});
    fin = ro_uniform(fin);

    unsigned int spins = 0u, ntile = 0u;
    bool gave_up = false;
    while (fin != (1u << KI) - 1u) {
        // ---- poll: lane i looks at item i's site; a site whose round word equals the item's count has a candidate pending
        if (tid < KI) {
            unsigned int ready = 0u;
            if (!((fin >> tid) & 1u)) {
                const unsigned int sidx = s_item[tid].site;
                const RoSync* sy = a.sync + sidx;
                const unsigned int rd = ro_peek(&sy->round);
                if (rd == s_my[tid]) {
                    ready = 1u;
                    if (rd == 0u) {                          // the first candidate is what osq_msefast_tensor_begin left in the state
                        const TensorSearch* ts = a.sites[sidx].ts;
                        s_scale[tid] = ts->scale_d;
                        s_zp[tid] = ts->zp;
                        s_done[tid] = ts->S.done ? 1u : 0u;
                    } else {
                        s_done[tid] = ro_peek(&sy->done);
                        s_zp[tid] = __uint_as_float(ro_peek(reinterpret_cast<const unsigned int*>(&sy->zp)));
                        s_scale[tid] = __longlong_as_double(static_cast<long long>(
                            __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&sy->scale_d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
                    }
                }
            }
            s_ready[tid] = ready;
        }
        __syncthreads();
        unsigned int evaluated = 0u;
        // ---- resident chunk groups: the item index is a compile-time constant (its values live in registers)
        ro_static_for([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if ((fin >> i) & 1u) return;
            if (ro_uniform(s_ready[i]) == 0u) return;
            if (ro_uniform(s_done[i]) != 0u) { fin |= 1u << i; return; }
            const RoItem it = s_item[i];
            const int64_t chunks = s_chunks[i];
            double* const part = s_part[i];
            const double sd = s_scale[i];
            const float z = s_zp[i];
            const float qmin = s_qmin[i], qmax = s_qmax[i];
            const double rcp = 1.0 / sd;
            const bool lean = a.lean_ok && s_finite[i] && rcp_division_exact(sd, 0.0, 0.0) && lean_level_exact(z, qmin, qmax);
            const float rcp32 = static_cast<float>(rcp), lo32 = qmin - z, hi32 = qmax - z;
            const int64_t m = static_cast<int64_t>(it.q) * G + j;
            double acc = 0.0;
            if (m < chunks) {
#pragma unroll
                for (int k0 = 0; k0 < R; k0 += 8) {
                    double t[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        t[k] = lean ? sq_err_f64_lean(r[i][k0 + k], sd, rcp, rcp32, lo32, hi32, z, qmin, qmax)
                                    : sq_err_f64_outofline(r[i][k0 + k], sd, z, qmin, qmax);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc = acc + t[k];
                }
            }
            // consecutive EVALUATED items alternate between two tiles: the barrier of the next one protects this one's readers
            double* const tile = lds_raw + (ntile & 1u) * kRoThreads;
            ++ntile;
            tile[(((j << P) + blk) << nc_shift) + c] = acc;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tid < (G << nc_shift)) {
                const int jj = tid >> nc_shift, cc = tid & (NC - 1);
                const int64_t mm = static_cast<int64_t>(it.q) * G + jj;
                if (mm < chunks) {
                    double v[S];
#pragma unroll
                    for (int b = 0; b < S; ++b) v[b] = tile[(((jj << P) + b) << nc_shift) + cc];
                    double s2 = 0.0;
#pragma unroll
                    for (int b = 0; b < S; ++b) s2 = s2 + v[b];
                    cascade_publish<double>(&part[(mm << nc_shift) + cc], s2);
                }
            }
            // only wave 0 publishes (G * NC = 32 lanes): it drains ITS stores and arrives for the item -- no workgroup barrier,
            // no end-of-sweep drain between this item's sums and the site's master
            if (tid < OSQ_WAVE) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tid == 0) {
                    __hip_atomic_fetch_add(&a.sync[it.site].arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_my[i] += 1u;
                }
            }
            evaluated |= 1u << i;
            __builtin_amdgcn_sched_barrier(0);               // one item's temporaries at a time: the registers hold data
        }, std::make_integer_sequence<int, KI>{});
        if (evaluated) {
            spins = 0u;
        } else {
            if (++spins > a.spin_limit) { gave_up = true; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
    }
    if (gave_up && tid == 0) __hip_atomic_fetch_or(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace osq
