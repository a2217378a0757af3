// Token range finaliser, fast path (included by observer.hip; needs MinMax / Finish / FinalBatch /
// SelState / level_shift / abs_key / uniform from there).
//
// prune_token + cac_thres + quantile_range + the clip/aminmax that follows (observer.py:50-70,
// 221-227) for ONE batch whose per-token extrema are already in memory, then the running statistic
// and calculate_qparams (observer.py:194-202, 101-119).
//
// The two halves of the statistic are the same computation on one array each:
//     side 0:  v =  token_max   up =  max(v[v <= quantile(|v|, p)])
//     side 1:  v = -token_min   lo = -max(v[v <= quantile(|v|, p)])
// so every problem gets TWO workgroups (blockIdx.x = side), each on its own CU: a CU pulls its
// array with 16-byte loads that are all issued before anything else (the single-workgroup
// predecessor spent 12 of its 28 us waiting for 64 scalar loads per thread), keeps it in registers
// (<= 32 values per thread, 32768 token slots), and runs the selection on half the LDS atomics.
// The sides meet through ONE 8-byte atomic exchange on a zero-idle rendezvous word: the first
// arriver leaves its result, the second reads it, finishes (clip rule, running statistic, qparams)
// and puts the word back to zero.  No tickets, no fences: the record IS the payload.
//
// Invalid slots (padding, t >= lengths[b]) are replaced by NaN right after the load: NaN fails every
// ordered comparison below, its key 0x7fc00000 lies above every finite/inf key range, and fmaxf
// ignores it, so no validity mask is carried through the passes.  A real NaN among the valid tokens
// is detected before the replacement and poisons the result like torch's max / quantile do.
#pragma once

namespace osq {

constexpr int kSelThreads = 1024;
constexpr int kSelWaves = kSelThreads / OSQ_WAVE;

struct SelectArgs {
    const float* tok_min;
    const float* tok_max;
    int64_t B, T;
    const int64_t* lengths;
    int prune;
    float q;
    unsigned long long* meet;     // one rendezvous word per problem, zero when idle
    int shortcut;                 // 0: always run the threshold pass over the registers (tests)
};

#ifdef OSQ_FINAL_TIMING
// stamps go to LDS (a global store in front of a barrier would be what gets measured) and leave at the end (OSQ_SDUMP)
#define OSQ_SSTAMP(k) do { if (threadIdx.x == 0) S.t_stamp[k] = wall_clock64(); } while (0)
#define OSQ_SDUMP() do { if (threadIdx.x == 0 && stamps) for (int k_ = 0; k_ < 16; ++k_) stamps[k_] = S.t_stamp[k_]; } while (0)
#else
#define OSQ_SSTAMP(k) do { } while (0)
#define OSQ_SDUMP() do { } while (0)
#endif

// One CU handles all of a side's values (<= 32768) at 64 lane-operations per clock, so every VALU
// instruction per value costs ~0.2 us: the passes below are written to a handful of instructions each
// (validity as one compare against a per-group count, NaN poisoning as one select, range through
// float min/max of |v| with source modifiers, the NaN flag as a scalar lane-mask OR).
//
// Two stages: a GATHERING pass that brings a side's values into registers (select_side below: a token array in
// memory; fused_step.h: chunk by chunk while the streaming workgroups of the same launch publish them) and
// select_from_registers, the selection proper.
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));

struct alignas(16) SelShared {            // LDS of one selecting workgroup (the list and the partials are read 16 bytes at a time)
    unsigned int hist[kSelBins];
    unsigned int list[kListCap];
    // per-wave partials of pass 0: plain stores, nothing to initialise, no atomics; every thread folds the sixteen entries
    unsigned int w_n[kSelWaves], w_bad[kSelWaves], w_kmin[kSelWaves], w_kmax[kSelWaves], w_plain[kSelWaves], w_below[kSelWaves];
    alignas(16) unsigned int pick[4];                 // what the owner of the wanted rank leaves: bin, keys below it, keys in it
    alignas(16) unsigned int s_found[2];              // keys at ranks floor / ceil ...
    unsigned int s_next, s_pos;                       // ... the smallest key above the bin, sign facts: read as ONE 16-byte word
    unsigned int s_fill, s_sel, s_late, s_pad;
    alignas(16) unsigned int s_wtot[kSelWaves];
#ifdef OSQ_FINAL_TIMING
    long long t_stamp[16];
#endif
};

struct SideResult {
    float value;      // max(v[v <= quantile(|v|, p)]) of this side's v (or its plain maximum)
    bool bad;         // a NaN among the valid tokens
    bool empty;       // no valid token at all
};

// Barrier for exchanges through LDS only: __syncthreads() is a workgroup-scope fence around s_barrier, and the fence
// also waits for every global store / load the wave has in flight (vmcnt(0)) -- a polling load, a published flag, a
// debug stamp: a memory round trip (~1 us while the chip streams) in front of every barrier of the selection.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct SelWindow {        // key range keys lo .. lo + wd - 1 in bins of 2^sh keys; on = false: no window
    bool on;
    unsigned int lo, wd, sh;
};

// HINT (round 3).  An averaging observer already holds a prediction of this batch's threshold: its running mean of
// the previous batches' clip values (observer.py:194-202), which IS a mean of their order statistics.  A selector that
// has to wait for its values anyway (fused_step.h: the streaming workgroups publish them while it idles) can build
// the first histogram level over the window hint * [1/2, 3/2] WHILE the values arrive -- keys below the window
// are counted, keys inside it histogrammed -- and only the scan is left once the last value is in.  Exactness does not
// depend on the hint: the window serves as level 0 only if the wanted rank falls inside it; otherwise (first batch,
// a distribution that moved, another percentile) the full-range level runs as before.
__device__ __forceinline__ SelWindow hint_window(const float hint, const int prune) {
    SelWindow w{false, 0u, 0u, 0u};
    if (prune && hint > 1e-30f && hint < 1e30f) {          // false for NaN, inf, zero and denormal-sized hints
        w.on = true;
        w.lo = __float_as_uint(hint * 0.5f);
        w.wd = __float_as_uint(hint * 1.5f) - w.lo + 1u;     // 1.5 octaves of keys: 1536 bins of 8192 keys
        w.sh = level_shift(w.wd);
    }
    return w;
}

struct SelPass0 {         // what the gathering pass leaves (all uniform)
    unsigned int N;           // valid values
    bool have_range;          // false: kmin / kmax / plain_o are computed from the registers if a path needs them
    unsigned int kmin, kmax;  // range of their keys
    unsigned int plain_o;     // ordered bits of their plain maximum
    bool any_bad;             // a NaN among them
    SelWindow win;            // win.on: S.hist holds the window's histogram, n_below keys lie below it
    unsigned int n_below;
};

// Per-wave partials of the gathering pass -> LDS -> (one barrier) -> every thread folds the sixteen entries.
__device__ __forceinline__ SelPass0 fold_pass0(SelShared& S, unsigned int n, const bool bad, float amin, float amax, float plain,
                                               unsigned int below, const SelWindow& win) {
    const int lane = threadIdx.x & (OSQ_WAVE - 1), wv = threadIdx.x / OSQ_WAVE;
    amin = wave_min(amin);
    amax = wave_max(amax);
    plain = wave_max(plain);
    const bool wbad = wave_any(bad);
    n = wave_inclusive_scan_u32(n);
    if (lane == OSQ_WAVE - 1) S.w_n[wv] = n;
    if (win.on) {
        below = wave_inclusive_scan_u32(below);
        if (lane == OSQ_WAVE - 1) S.w_below[wv] = below;
    }
    if (lane == 0) {
        S.w_bad[wv] = wbad ? 1u : 0u;
        S.w_kmin[wv] = __float_as_uint(amin);          // non-negative floats order like their bit patterns
        S.w_kmax[wv] = __float_as_uint(amax);
        S.w_plain[wv] = ordered_bits(plain);
    }
    lds_barrier();                           // also: the LDS set-up of the caller is complete
    SelPass0 p{0u, true, 0xffffffffu, 0u, 0u, false, win, 0u};
    unsigned int any_bad_u = 0u;
    if (win.on) {
#pragma unroll
        for (int k = 0; k < kSelWaves; k += 4) {
            const uint4 f = *reinterpret_cast<const uint4*>(&S.w_below[k]);
            p.n_below += f.x + f.y + f.z + f.w;
        }
        p.n_below = uniform(p.n_below);
    }
#pragma unroll
    for (int k = 0; k < kSelWaves; k += 4) {
        const uint4 a = *reinterpret_cast<const uint4*>(&S.w_n[k]), b = *reinterpret_cast<const uint4*>(&S.w_bad[k]);
        const uint4 c = *reinterpret_cast<const uint4*>(&S.w_kmin[k]), d = *reinterpret_cast<const uint4*>(&S.w_kmax[k]);
        const uint4 e = *reinterpret_cast<const uint4*>(&S.w_plain[k]);
        p.N += a.x + a.y + a.z + a.w;
        any_bad_u |= b.x | b.y | b.z | b.w;
        p.kmin = min(min(p.kmin, min(c.x, c.y)), min(c.z, c.w));
        p.kmax = max(max(p.kmax, max(d.x, d.y)), max(d.z, d.w));
        p.plain_o = max(max(p.plain_o, max(e.x, e.y)), max(e.z, e.w));
    }
    p.N = uniform(p.N);
    p.kmin = uniform(p.kmin);
    p.kmax = uniform(p.kmax);
    p.plain_o = uniform(p.plain_o);
    p.any_bad = uniform(any_bad_u) != 0u;
    return p;
}

// does register i of a thread hold anything? (compile-time i: a uniform test, or nothing at all)
template <int PERIOD, int MAINS>
__device__ __forceinline__ bool sel_used(const int i, const unsigned int n_tail) {
    if (PERIOD == 0) return true;
    const int r = i % (PERIOD ? PERIOD : 1);
    return r < MAINS || static_cast<unsigned int>(r - MAINS) < n_tail;
}

// Range of the keys and plain maximum of the values in registers (a gathering pass that left them out: only the paths
// without a usable hinted window need them).  One barrier; S.w_kmin / w_kmax / w_plain must not be in use.
template <int R, int PERIOD, int MAINS>
__device__ __forceinline__ void range_from_registers(const float (&v)[R], const unsigned int n_tail, SelShared& S,
                                                     unsigned int* kmin, unsigned int* kmax, unsigned int* plain_o) {
    const int lane = threadIdx.x & (OSQ_WAVE - 1), wv = threadIdx.x / OSQ_WAVE;
    float amin = __builtin_inff(), amax = 0.0f, plain = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (sel_used<PERIOD, MAINS>(i, n_tail)) {
            amin = fminf(amin, __builtin_fabsf(v[i]));     // poisoned slots are NaN: fminf / fmaxf ignore them
            amax = fmaxf(amax, __builtin_fabsf(v[i]));
            plain = fmaxf(plain, v[i]);
        }
    }
    amin = wave_min(amin);
    amax = wave_max(amax);
    plain = wave_max(plain);
    if (lane == 0) {
        S.w_kmin[wv] = __float_as_uint(amin);
        S.w_kmax[wv] = __float_as_uint(amax);
        S.w_plain[wv] = ordered_bits(plain);
    }
    lds_barrier();
    unsigned int lo = 0xffffffffu, hi = 0u, pl = 0u;
#pragma unroll
    for (int k = 0; k < kSelWaves; k += 4) {
        const uint4 c = *reinterpret_cast<const uint4*>(&S.w_kmin[k]), d = *reinterpret_cast<const uint4*>(&S.w_kmax[k]);
        const uint4 e = *reinterpret_cast<const uint4*>(&S.w_plain[k]);
        lo = min(min(lo, min(c.x, c.y)), min(c.z, c.w));
        hi = max(max(hi, max(d.x, d.y)), max(d.z, d.w));
        pl = max(max(pl, max(e.x, e.y)), max(e.z, e.w));
    }
    *kmin = uniform(lo);
    *kmax = uniform(hi);
    *plain_o = uniform(pl);
}

// The selection proper on R values per thread held in registers (invalid slots poisoned to NaN), after the gathering
// pass.  S.hist is zero -- or holds the hinted window's histogram (p0.win.on) --, S.s_* are initialised.
// PERIOD > 0: the registers come in groups of PERIOD, of which the first MAINS and the n_tail (uniform) after them hold
// something (fused_step.h); PERIOD = 0: all R do.
template <int R, int PERIOD = 0, int MAINS = 0>
__device__ __forceinline__ SideResult select_from_registers(const float (&v)[R], const SelPass0& p0, const int prune, const float aq,
                                                            const int use_shortcut, SelShared& S, long long* stamps,
                                                            const unsigned int n_tail = 0u) {
    const int tid = threadIdx.x, lane = tid & (OSQ_WAVE - 1), wv = tid / OSQ_WAVE;
    const unsigned int N = p0.N;
    if (N == 0u) return SideResult{0.0f, false, true};     // both sides agree: nothing observed
    const bool any_bad = p0.any_bad;
    unsigned int kmin = p0.kmin, kmax = p0.kmax, plain_o = p0.plain_o;
    bool have_range = p0.have_range;                       // uniform
    if (!have_range && !(prune && !any_bad)) {             // the plain maximum is the result
        range_from_registers<R, PERIOD, MAINS>(v, n_tail, S, &kmin, &kmax, &plain_o);
        have_range = true;
    }
    float result = have_range ? from_ordered_bits(plain_o) : 0.0f;

    if (prune && !any_bad) {
        const float rank = aq * static_cast<float>(N - 1u);
        const float rlo = floorf(rank);
        const unsigned int k_lo = static_cast<unsigned int>(rlo);
        const unsigned int k_hi = static_cast<unsigned int>(ceilf(rank));
        const float w = rank - rlo;
        // a window histogrammed during the gathering pass serves as level 0 if the wanted rank lies inside it
        bool prehist = p0.win.on && p0.n_below <= k_lo;
        if (!prehist && !have_range) {
            range_from_registers<R, PERIOD, MAINS>(v, n_tail, S, &kmin, &kmax, &plain_o);
            have_range = true;
        }
        // The state of the search lives in REGISTERS of every thread (all uniform, all derived from uniform inputs): no
        // set-up through LDS, no barrier for it.  A level costs two barriers -- the wave totals of the scan and the three
        // words the thread that owns the wanted rank leaves (bin, keys below it, keys in it) -- plus one for its
        // histogram when that was not built during the gathering pass.
        unsigned int sel_lo, sel_width, sel_rank, sel_shift, sel_le, sel_count = N;
        bool sel_done = false;
        if (prehist) {
            sel_lo = p0.win.lo; sel_width = p0.win.wd; sel_rank = k_lo - p0.n_below; sel_shift = p0.win.sh; sel_le = p0.n_below;
        } else {
            sel_lo = kmin; sel_width = kmax - kmin + 1u; sel_rank = k_lo; sel_shift = level_shift(sel_width); sel_le = 0u;
            if (p0.win.on) {                       // the window lies above the rank: its counts are of no use
                for (int k = tid; k < kSelBins; k += kSelThreads) S.hist[k] = 0u;
                lds_barrier();
            }
        }
        OSQ_SSTAMP(7);
        // ---- histogram levels: level 0 always; 1-2 only while the chosen bin is too crowded for the S.list
        bool listed = false;
        for (int level = 0; level < 3; ++level) {
            if (sel_done) break;
            if (level > 0) {
                if (sel_count <= kListCap) { listed = true; break; }
                for (int k = tid; k < kSelBins; k += kSelThreads) S.hist[k] = 0u;      // everybody read its bins before the last barrier
                lds_barrier();
            }
            if (!(level == 0 && prehist)) {
                // The range check also keeps poisoned slots (key 0x7fc00000) out.  Measured alternatives: all of them
                // into ONE trash bin is 5x slower (same-address LDS atomics serialise); one trash bin per lane with a
                // v_min instead of the compare + exec masking is no faster (6.6k vs 6.4k cycles at 32768 slots) --
                // the pass is bound by the LDS atomic rate (~10 clocks per wave instruction), and masking does fewer.
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (sel_used<PERIOD, MAINS>(i, n_tail)) {
                        const unsigned int d = abs_key(v[i]) - sel_lo;
                        if (d < sel_width) atomicAdd(&S.hist[d >> sel_shift], 1u);
                    }
                }
                lds_barrier();
            }
            // Scan over the 2048 bins by FOUR waves, 8 bins per thread; the thread whose bins straddle the rank reports.  Every
            // instruction of a 16-wave workgroup costs its SIMD ~16 clocks whether 64 threads have work or 1024: with all
            // sixteen waves on 2 bins each the two halves of this scan took 1.4 + 0.8 us; the other twelve waves now go
            // straight to the barriers and leave each SIMD to one wave (round 6; the counters of the stand-alone kernel,
            // profiles/r06_token_select_pmc.txt: the SIMDs of the selecting CU issue 81 % of the time).
            constexpr int kScanWaves = 4, kScanBins = kSelBins / (kScanWaves * OSQ_WAVE);      // 8
            unsigned int hb[kScanBins], hsum = 0u, incl_w = 0u;
            if (wv < kScanWaves) {
                const uint4 ha = *reinterpret_cast<const uint4*>(&S.hist[kScanBins * tid]);
                const uint4 hc = *reinterpret_cast<const uint4*>(&S.hist[kScanBins * tid + 4]);
                hb[0] = ha.x; hb[1] = ha.y; hb[2] = ha.z; hb[3] = ha.w; hb[4] = hc.x; hb[5] = hc.y; hb[6] = hc.z; hb[7] = hc.w;
                hsum = ((hb[0] + hb[1]) + (hb[2] + hb[3])) + ((hb[4] + hb[5]) + (hb[6] + hb[7]));
                incl_w = wave_inclusive_scan_u32(hsum);
                if (lane == OSQ_WAVE - 1) S.s_wtot[wv] = incl_w;
            }
            lds_barrier();
            const uint4 t4 = *reinterpret_cast<const uint4*>(&S.s_wtot[0]);
            const unsigned int inside = uniform(t4.x + t4.y + t4.z + t4.w);
            const unsigned int base = wv == 0 ? 0u : (wv == 1 ? t4.x : (wv == 2 ? t4.x + t4.y : t4.x + t4.y + t4.z));
            if (level == 0) OSQ_SSTAMP(8);
            if (level == 0 && prehist && sel_rank >= inside) {
                // the rank lies above the window (uniform: every thread sees the same sum): the full-range level after all
                prehist = false;
                if (!have_range) {
                    range_from_registers<R, PERIOD, MAINS>(v, n_tail, S, &kmin, &kmax, &plain_o);
                    have_range = true;
                }
                for (int k = tid; k < kSelBins; k += kSelThreads) S.hist[k] = 0u;
                sel_lo = kmin; sel_width = kmax - kmin + 1u; sel_rank = k_lo; sel_shift = level_shift(sel_width); sel_le = 0u;
                lds_barrier();
                --level;
                continue;
            }
            if (wv < kScanWaves) {
                const unsigned int incl = base + incl_w, excl = incl - hsum;
                if (sel_rank >= excl && sel_rank < incl) {     // exactly one thread: which of its 8 bins
                    unsigned int bel = excl, bin_k = 0u, cnt_k = hb[0];
#pragma unroll
                    for (int k = 1; k < kScanBins; ++k) {
                        const bool beyond = sel_rank >= bel + cnt_k;      // the rank lies behind bin bin_k: move on to bin k
                        bel = beyond ? bel + cnt_k : bel;
                        bin_k = beyond ? static_cast<unsigned int>(k) : bin_k;
                        cnt_k = beyond ? hb[k] : cnt_k;
                    }
                    uint4 pk;
                    pk.x = static_cast<unsigned int>(kScanBins) * static_cast<unsigned int>(tid) + bin_k;      // the bin
                    pk.y = bel;                                                                // keys of the range below it
                    pk.z = cnt_k;                                                              // keys in it
                    pk.w = 0u;
                    *reinterpret_cast<uint4*>(&S.pick[0]) = pk;
                }
            }
            lds_barrier();
            {
                const uint4 pk = *reinterpret_cast<const uint4*>(&S.pick[0]);
                const unsigned int bin = uniform(pk.x), below_b = uniform(pk.y), cnt = uniform(pk.z);
                const unsigned int off = bin << sel_shift;
                sel_lo += off;
                sel_count = cnt;
                if (sel_shift == 0u) {             // single-key bins: found
                    sel_le += below_b + cnt;
                    sel_width = 0u;
                    sel_done = true;
                } else {
                    const unsigned int rest = sel_width - off, cap = 1u << sel_shift;
                    sel_le += below_b;
                    sel_rank -= below_b;
                    sel_width = rest < cap ? rest : cap;
                    sel_shift = level_shift(sel_width);
                }
            }
        }
        OSQ_SSTAMP(3);
#ifdef OSQ_FINAL_TIMING
        if (tid == 0) { S.t_stamp[11] = prehist; S.t_stamp[12] = p0.n_below; S.t_stamp[13] = k_lo; S.t_stamp[14] = sel_count; S.t_stamp[15] = p0.win.on; }
#endif
        unsigned int v_lo, v_hi;     // keys at floor(rank) / ceil(rank)
        bool shortcut_done = false;  // uniform
        if (!sel_done && listed) {
            // ---- compact the chosen bin's keys; the smallest key above the bin only if rank+1 leaves the bin
            const unsigned int lo = sel_lo, wd = sel_width;
            const bool need_next = (k_hi != k_lo) && (sel_rank + 1u >= sel_count);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (sel_used<PERIOD, MAINS>(i, n_tail)) {
                    const unsigned int key = abs_key(v[i]);
                    if (key - lo < wd) S.list[atomicAdd(&S.s_fill, 1u)] = __float_as_uint(v[i]);    // sign kept: see the shortcut below
                }
            }
            if (need_next) {
                // smallest key at or above the bin's end: keys below it wrap to huge values under the
                // unsigned subtraction and lose the min (poisoned slots are above every valid key)
                const unsigned int edge = lo + wd;
                unsigned int nx = 0xffffffffu;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (sel_used<PERIOD, MAINS>(i, n_tail)) nx = min(nx, abs_key(v[i]) - edge);
                }
                nx = wave_min_u32(nx);
                if (lane == 0 && nx < 0x80000000u) atomicMin(&S.s_next, nx + edge);   // >= 2^31: only wrapped keys in this wave
            }
            lds_barrier();
            OSQ_SSTAMP(9);
            // ---- rank by counting.  Entry `mine` is the key at rank r iff (#keys < mine) <= r < (#keys <= mine): every
            // holder of the keys at ranks want / want + 1 reports itself and whether it is non-negative -- one barrier
            // for the two order statistics AND the sign facts of the shortcut below
            const unsigned int cnt = S.s_fill, want = sel_rank;
            if (static_cast<unsigned int>(tid) < cnt) {
                const unsigned int ent = S.list[tid], mine = ent & 0x7fffffffu;
                unsigned int lt = 0u, le = 0u;
                for (unsigned int j = 0; j < cnt; j += 4u) {        // 16-byte LDS reads; entries at or beyond cnt are stale, never counted
                    const uint4 o4 = *reinterpret_cast<const uint4*>(&S.list[j]);
                    const unsigned int o[4] = {o4.x & 0x7fffffffu, o4.y & 0x7fffffffu, o4.z & 0x7fffffffu, o4.w & 0x7fffffffu};
#pragma unroll
                    for (unsigned int e = 0; e < 4u; ++e) {
                        const bool in = j + e < cnt;
                        lt += (in && o[e] < mine) ? 1u : 0u;
                        le += (in && o[e] <= mine) ? 1u : 0u;
                    }
                }
                if (lt <= want && want < le) {
                    S.s_found[0] = mine;
                    if (!(ent >> 31)) atomicOr(&S.s_pos, 1u);
                }
                if (lt <= want + 1u && want + 1u < le) {
                    S.s_found[1] = mine;
                    if (!(ent >> 31)) atomicOr(&S.s_pos, 2u);
                }
            }
            lds_barrier();
            const uint4 fnd = *reinterpret_cast<const uint4*>(&S.s_found[0]);        // s_found[0], s_found[1], s_next, s_pos: one read
            v_lo = fnd.x;
            const bool hi_listed = fnd.y != 0xffffffffu;
            v_hi = hi_listed ? fnd.y : fnd.z;
            // Shortcut for the threshold pass.  The keys at ranks floor/ceil are neighbours in sorted order, so
            // no key lies strictly between them, thr lies in [lo_v, hi_v], and every value with a larger key
            // is either negative or above thr.  If some element with key lo_v is non-negative, then
            // max(v[v <= thr]) is lo_v -- or hi_v when thr reaches it and a non-negative element has that
            // key.  Both facts are in the S.list (it holds every element of the bin, with sign) as long as the
            // upper key is listed or not reached; otherwise the register pass below decides.
            if (use_shortcut) {
                const unsigned int pos = fnd.w;
                const float lo_f = __uint_as_float(v_lo), hi_f = __uint_as_float(k_hi == k_lo ? v_lo : v_hi);
                const float d = hi_f - lo_f;
                const float t = (w < 0.5f) ? __builtin_fmaf(w, d, lo_f) : __builtin_fmaf(w - 1.0f, d, hi_f);
                const bool reaches_hi = hi_f > lo_f && t >= hi_f;
                if ((pos & 1u) && (!reaches_hi || hi_listed)) {
                    shortcut_done = true;
                    result = (reaches_hi && (pos & 2u)) ? hi_f : lo_f;
                }
            }
        } else {
            // every level ran (massive duplicates): sel_lo is the key at rank k_lo, sel_le = #keys <= it
            v_lo = sel_lo;
            if (k_hi != k_lo && sel_le <= k_hi) {             // rank k_hi is the smallest key above
                unsigned int nx = 0xffffffffu;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const unsigned int key = abs_key(v[i]);
                    if (sel_used<PERIOD, MAINS>(i, n_tail) && key > v_lo) nx = min(nx, key);
                }
                nx = wave_min_u32(nx);
                if (lane == 0) atomicMin(&S.s_next, nx);
                lds_barrier();
                v_hi = S.s_next;
            } else {
                v_hi = v_lo;
            }
        }
        if (k_hi == k_lo) v_hi = v_lo;
        OSQ_SSTAMP(4);
        const float lo_v = __uint_as_float(v_lo), hi_v = __uint_as_float(v_hi), diff = hi_v - lo_v;
        float thr = (w < 0.5f) ? __builtin_fmaf(w, diff, lo_v) : __builtin_fmaf(w - 1.0f, diff, hi_v);   // torch lerp
        thr = __uint_as_float(uniform(__float_as_uint(thr)));
        if (!shortcut_done) {
            // ---- max(v[v <= thr]) over the registers
            float best = -__builtin_inff();
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (sel_used<PERIOD, MAINS>(i, n_tail)) best = (v[i] <= thr) ? fmaxf(best, v[i]) : best;
            }
            best = wave_max(best);
            if (lane == 0) atomicMax(&S.s_sel, ordered_bits(best));
            lds_barrier();
            result = from_ordered_bits(S.s_sel);
        }
    }
    OSQ_SSTAMP(5);
    return SideResult{result, any_bad, false};
}


// select_side: one side's statistic of a token array in memory (slots b*T + t, valid iff t < lengths[b]; the arrays
// were written by an earlier launch: plain loads), computed by the calling 1024-thread workgroup; every thread gets
// the result.  R4 = 16-byte groups per thread; group g = tid + 1024*j covers slots 4g .. 4g+3.
template <int R4>
__device__ __forceinline__ SideResult select_side(const float* src, const int side, const int64_t aB, const int64_t aT,
                                                  const int64_t* lengths, const int prune, const float aq,
                                                  const int use_shortcut, SelShared& S, long long* stamps) {
    constexpr int R = 4 * R4;

    const int tid = threadIdx.x;
    const unsigned int Tu = static_cast<unsigned int>(aT);
    const unsigned int groups = static_cast<unsigned int>((aB * aT) >> 2);
    const unsigned int Bm1 = static_cast<unsigned int>(aB) - 1u;
    const bool straddle = (Tu & 3u) != 0u;                 // T % 4 == 0: the four slots of a group share their sample
    const unsigned int flip = side ? 0x80000000u : 0u;     // side 1 works on -token_min

    OSQ_SSTAMP(0);
    // ---- lengths first (L2 hits, needed before the data), then every data load, all unconditional.
    // Slot k of a group is valid iff k < rem_a (same sample as slot 0) or, behind the sample boundary
    // k >= wrap (T % 4 != 0 only), iff k < rem_b.
    int rem_a[R4], rem_b[R4], wrap[R4];
    {
        const unsigned int step_b = (4u * kSelThreads) / Tu, step_t = 4u * kSelThreads - step_b * Tu;
        unsigned int bb = (4u * static_cast<unsigned int>(tid)) / Tu, tt = 4u * static_cast<unsigned int>(tid) - bb * Tu;
#pragma unroll
        for (int j = 0; j < R4; ++j) {
            const unsigned int g = static_cast<unsigned int>(tid) + static_cast<unsigned int>(j) * kSelThreads;
            int64_t la = aT, lb = aT;
            if (lengths) {
                la = lengths[bb < Bm1 ? bb : Bm1];
                lb = straddle ? lengths[bb + 1u < Bm1 ? bb + 1u : Bm1] : la;
            }
            const int ia = la > aT ? static_cast<int>(aT) : (la < 0 ? 0 : static_cast<int>(la));
            const int ib = lb > aT ? static_cast<int>(aT) : (lb < 0 ? 0 : static_cast<int>(lb));
            const int to_end = static_cast<int>(Tu - tt);           // slots left in this sample, >= 1
            wrap[j] = to_end;
            rem_a[j] = g < groups ? ia - static_cast<int>(tt) : 0;
            rem_b[j] = g < groups ? ib + to_end : 0;
            bb += step_b;
            tt += step_t;
            if (tt >= Tu) { tt -= Tu; ++bb; }
        }
    }
    float4 raw[R4];
    {
        const float4* src4 = reinterpret_cast<const float4*>(src);
#pragma unroll
        for (int j = 0; j < R4; ++j) {
            const unsigned int g = static_cast<unsigned int>(tid) + static_cast<unsigned int>(j) * kSelThreads;
            raw[j] = src4[g < groups ? g : groups - 1u];
        }
    }
    // LDS set-up overlaps the loads
    for (int k = tid; k < kSelBins; k += kSelThreads) S.hist[k] = 0u;
    if (tid == 0) {
        S.s_fill = 0u; S.s_next = 0xffffffffu; S.s_found[0] = S.s_found[1] = 0xffffffffu; S.s_sel = 0u; S.s_pos = 0u;
    }
    OSQ_SSTAMP(1);

    // ---- pass 0 (as the values arrive): poison invalid slots, N, NaN flag, range of |v|, plain maximum
    float v[R];
    unsigned int n = 0u;
    bool bad = false;                          // a lane mask in SGPRs: the OR below is scalar work
    float amin = __builtin_inff(), amax = 0.0f, plain = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < R4; ++j) {
        const float e[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
        if (!straddle && __all(rem_a[j] >= 4)) {
            // the whole wave's groups are valid (all but one or two waves of a problem): no validity selects
            n += 4u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float x = __uint_as_float(__float_as_uint(e[k]) ^ flip);
                bad |= x != x;
                amin = fminf(amin, __builtin_fabsf(x));
                amax = fmaxf(amax, __builtin_fabsf(x));
                plain = fmaxf(plain, x);
                v[4 * j + k] = x;
            }
            continue;
        }
        bool ok[4];
        if (!straddle) {
            const int r = rem_a[j];
            n += static_cast<unsigned int>(r < 0 ? 0 : (r > 4 ? 4 : r));
#pragma unroll
            for (int k = 0; k < 4; ++k) ok[k] = k < r;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ok[k] = (k < wrap[j]) ? (k < rem_a[j]) : (k < rem_b[j]);
                n += ok[k] ? 1u : 0u;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xs = __uint_as_float(__float_as_uint(e[k]) ^ flip);
            bad |= ok[k] && (xs != xs);
            const float x = ok[k] ? xs : __builtin_nanf("");
            amin = fminf(amin, __builtin_fabsf(x));
            amax = fmaxf(amax, __builtin_fabsf(x));
            plain = fmaxf(plain, x);
            v[4 * j + k] = x;
        }
    }
    const SelPass0 p0 = fold_pass0(S, n, bad, amin, amax, plain, 0u, SelWindow{false, 0u, 0u, 0u});
    OSQ_SSTAMP(2);
    return select_from_registers<R>(v, p0, prune, aq, use_shortcut, S, stamps);
}

// Rendezvous of the two sides (thread 0 of each side's workgroup): the first arriver leaves
// {value, present | bad} in the zero-idle word, the second takes it, puts the word back to zero and returns true
// with the batch's (cur_min, cur_max): the reference's clip rule aminmax(clip(value, lo, up)), observer.py:68,227.
__device__ __forceinline__ bool meet_sides(unsigned long long* word, const int side, const SideResult& r,
                                           float* cur_min, float* cur_max) {
    const unsigned long long mine = (static_cast<unsigned long long>(0x80000000u | (r.bad ? 1u : 0u)) << 32) |
                                    __float_as_uint(r.value);
    const unsigned long long other = __hip_atomic_exchange(word, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!(other >> 63)) return false;
    __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float theirs = __uint_as_float(static_cast<unsigned int>(other & 0xffffffffull));
    const bool poisoned = r.bad || ((other >> 32) & 1ull);
    const float up = side ? theirs : r.value;
    const float lo = -(side ? r.value : theirs);
    *cur_min = clipped_min(lo, up);
    *cur_max = up;
    if (poisoned) { *cur_min = __builtin_nanf(""); *cur_max = *cur_min; }
    return true;
}

// Stand-alone launch: TWO workgroups per problem (blockIdx.x = side, blockIdx.y = problem).
template <int R4>
__global__ __launch_bounds__(kSelThreads) void token_select_kernel(SelectArgs a, Finish fin, FinalBatch fb) {
    const int side = blockIdx.x;
    const int64_t p = blockIdx.y;
    const float* src = side ? a.tok_min : a.tok_max;
    const int64_t* lengths = a.lengths;
    int prune = a.prune;
    if (fb.n_batches > 0) {
        const int64_t qi = p / fb.n_batches, bi = p - qi * fb.n_batches;
        src += p * fb.problem_stride;
        if (lengths) lengths += (fb.lengths_per_problem ? p : bi) * a.B;
        prune = fb.prune_flags ? fb.prune_flags[qi] : prune;
        fin.cur += 2 * (bi * fb.n_quantizers + qi);
    }
    __shared__ SelShared S;
    // running state for the finish step, fetched now so that the second arriver's tail has no dependent load
    float st_min = 0.f, st_max = 0.f;
    const bool have_state = fin.rule != OSQ_UPDATE_NONE && fin.min_val && fin.max_val;
    if (threadIdx.x == 0 && have_state) { st_min = fin.min_val[0]; st_max = fin.max_val[0]; }
#ifdef OSQ_FINAL_TIMING
    long long* stamps = (blockIdx.x == 0 && fin.cur) ? reinterpret_cast<long long*>(fin.cur + 2) : nullptr;
#else
    long long* stamps = nullptr;
#endif
    const SideResult r = select_side<R4>(src, side, a.B, a.T, lengths, prune, a.q, a.shortcut, S, stamps);
    if (r.empty) return;                       // nothing observed, nothing updated
    if (threadIdx.x == 0) {
        float cur_min, cur_max;
        if (meet_sides(&a.meet[p], side, r, &cur_min, &cur_max))
            finish_entry(fin, 0, cur_min, cur_max, have_state, st_min, st_max);
    }
#ifdef OSQ_FINAL_TIMING
    OSQ_SSTAMP(6);
    OSQ_SDUMP();
#endif
}

}  // namespace osq
