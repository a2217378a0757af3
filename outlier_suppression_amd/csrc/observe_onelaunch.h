// A whole masked observation -- per-token extrema, token-wise clipping (prune_token + cac_thres + quantile_range,
// observer.py:50-70), the clip rule (observer.py:68,227), the running statistic (observer.py:194-202) and
// calculate_qparams (observer.py:101-119) -- in ONE launch.  Included by observer.hip after token_select.h.
//
// The two-launch form (token_minmax, then token_select on two CUs) costs a kernel boundary and a selection that only
// starts when the last token has been reduced: 11.6 + 11.9 us on [256,128,768] with 54 % valid tokens.  Here the two
// SELECTOR workgroups (one per side: token_max / -token_min) belong to the same launch as the STREAMING workgroups
// (16 tokens each, the body of token_minmax_vec_kernel) and work while those stream:
//
//   * every workgroup knows the PIVOT of a side before it reads a byte: 7/8 of the observer's running statistic, which
//     is a mean of the previous batches' thresholds.  A token whose |extremum| reaches the pivot is a CANDIDATE for the
//     percentile's neighbourhood (the top ~10-15 % of the tokens at p = 0.95); the others only matter through their
//     count and their maximum.  A streaming workgroup publishes, per side, ONE 32-byte record = two self-tagged 16-byte
//     halves {tag, candidate mask | NaN flag, max of the non-candidates, v0} {tag, v1, v2, v3}: its first four
//     candidates travel inside the record (93 % of the groups have no more), the rest are picked from the token
//     arrays, which every workgroup also writes (and drains before the record) -- they are the fall-back's input.
//   * a selector (256 threads, like every workgroup of the launch) polls the records of the groups it owns, absorbs
//     candidates as they arrive -- LDS list + first histogram level over one octave above the pivot -- and when the
//     last record is in, what is left is a 1024-bin scan, a short list, a rank by counting and the exchange with the
//     other side.  No workgroup ever waits for a selector, so nothing depends on the grid being resident together.
//
// EXACT whatever the pivot: candidates and non-candidates are counted, the histogram serves only if the wanted rank
// falls into it, and otherwise -- first batch (no running statistic), a threshold that dropped by more than 1/8, more
// than 4096 candidates -- the same selector runs the full-range search over its list or, if that does not hold the
// rank, over the token arrays in memory (several passes of 4 B per token: slower than the two-launch form, and rare).
#pragma once

namespace osq {

constexpr int kOlThreads = kThreads;                 // 256: streaming and selecting workgroups share one launch
constexpr int kOlWaves = kOlThreads / OSQ_WAVE;
constexpr int kOlGroupTokens = kTokPerBlock;         // 16 tokens per streaming workgroup
constexpr int kOlMaxGroups = 4096;                   // 65536 token slots
constexpr int kOlMaxBatch = 1024;
constexpr int kOlListCap = 4096;                     // candidates a selector keeps (LDS)
constexpr int kOlBinBits = 10;
constexpr int kOlBins = 1 << kOlBinBits;
constexpr int kOlPickCap = 512;                      // keys of the chosen bin ranked by counting
constexpr unsigned int kOlSpinLimit = 1u << 21;
constexpr unsigned int kOlNoCandidates = 0x7f800001u; // a pivot above every key: nothing is a candidate (no pruning)

struct OneLaunchState {                              // workspace slice; ALL-ZERO before the first launch
    unsigned int epoch, pad0[15];                    // tag of a launch = epoch + 1; advanced by the selector that finishes
    unsigned int status, pad1[15];                   // sticky: 1 = a selector gave up waiting for a record
    unsigned int rec[2][kOlMaxGroups][8];            // per side and streaming workgroup: {tag, mask | bad << 16, max of the others, v0} {tag, v1, v2, v3}
};

struct OneLaunchArgs {
    const float* x;
    osq_token_view v;
    const int64_t* lengths;
    float* tok_min;
    float* tok_max;
    int lgG, inner4, chunks;          // chunks = tokens / 16
    int prune;
    float q;
    int shortcut;
    int use_hint;                     // 0: no pivot (tests: every token is a candidate)
    OneLaunchState* st;
    unsigned long long* meet;
    unsigned int spin_limit;
    int phase;                        // 0: streaming and selecting workgroups in one launch; 1: the streaming ones only; 2: the selectors only (after a phase-1 launch)
};

__device__ __forceinline__ unsigned int ol_pivot_key(const float hint, const int prune, const bool usable) {
    if (!prune) return kOlNoCandidates;
    if (usable && hint > 1e-30f && hint < 1e30f) return __float_as_uint(hint * 0.875f);     // false for NaN / inf / zero / negative hints
    return 0u;                                                                              // every valid token is a candidate
}
__device__ __forceinline__ unsigned int ol_level_shift(const unsigned int width) {
    const unsigned int bits = width <= 1u ? 0u : 32u - __builtin_clz(width - 1u);
    return bits > kOlBinBits ? bits - kOlBinBits : 0u;
}
__device__ __forceinline__ unsigned int ol_lanes_below(const unsigned long long m) {     // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned int>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned int>(m), 0u));
}

struct alignas(16) OlSelShared {
    int lens[kOlMaxBatch];
    unsigned int list[kOlListCap];
    unsigned int hist[kOlBins];
    unsigned int pick[kOlPickCap];
    alignas(16) unsigned int w_tot[kOlWaves];
    unsigned int w_n[kOlWaves], w_bad[kOlWaves], w_plain[kOlWaves], w_nc[kOlWaves], w_kmin[kOlWaves], w_kmax[kOlWaves];
    alignas(16) unsigned int sel[4];                  // bin, keys below it, keys in it
    alignas(16) unsigned int s_found[2];
    unsigned int s_next, s_pos;
    unsigned int s_count, s_fill, s_best, s_timeout;
};
struct OlStreamShared {
    float val[2][kOlGroupTokens];
    unsigned int mask[2][kOlWaves], nc[2][kOlWaves], bad[kOlWaves];
};
union OlShared {
    OlSelShared sel;
    OlStreamShared str;
};

// ---------------------------------------------------------------- streaming workgroup
template <bool SINGLE_SEGMENT, bool NT>
__device__ __forceinline__ void ol_stream(const OneLaunchArgs& a, const Finish& fin, OlStreamShared& S) {
    const int tid = threadIdx.x, lane = tid & (OSQ_WAVE - 1), w = tid / OSQ_WAVE;
    const int64_t b = static_cast<int64_t>(blockIdx.y) - 1;
    int64_t len = a.v.tokens;
    if (a.lengths) {
        const int64_t l = a.lengths[b];
        len = l < len ? (l < 0 ? 0 : l) : len;
    }
    // chunk index rotated by the sample index: see token_minmax_vec_kernel (XCD balance)
    const int64_t chunk = (static_cast<int64_t>(blockIdx.x) + b) % gridDim.x;
    if (chunk * kOlGroupTokens >= len) return;                   // nothing valid here: the selectors know it from the lengths
#ifdef OSQ_FINAL_TIMING
    const long long dbg_t0 = wall_clock64();
#endif
    // the launch's tag and the pivots are only needed behind the streaming: their loads travel with it
    const unsigned int epoch = __hip_atomic_load(&a.st->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool have_state = fin.rule != OSQ_UPDATE_NONE && fin.min_val && fin.max_val;
    const float h0 = have_state ? fin.max_val[0] : 0.0f, h1 = have_state ? -fin.min_val[0] : 0.0f;
    const int64_t t0 = chunk * kOlGroupTokens + w * kTokPerWave;
    const int64_t group = b * a.chunks + chunk;
    unsigned int m[2] = {0u, 0u}, nc[2] = {ordered_bits(-__builtin_inff()), ordered_bits(-__builtin_inff())}, wbad = 0u;
    if (t0 < len) {
        const int ntok = (len - t0) < kTokPerWave ? static_cast<int>(len - t0) : kTokPerWave;
        const float* base = a.x + b * a.v.stride_batch + t0 * a.v.stride_token;
        MinMax acc[kTokPerWave];
        token_extrema<SINGLE_SEGMENT, NT>(base, a.v, ntok, a.lgG, a.inner4, lane, acc);
        float mn = acc[0].mn, mx = acc[0].mx;
#pragma unroll
        for (int k = 1; k < kTokPerWave; ++k)
            if (lane == k) { mn = acc[k].mn; mx = acc[k].mx; }
        const bool mine = lane < ntok;
        if (mine) {                                              // the fall-back's input (and the cached grid search's layout)
            const int64_t slot = group * kOlGroupTokens + w * kTokPerWave + lane;
            publish_f32(&a.tok_min[slot], mn);
            publish_f32(&a.tok_max[slot], mx);
        }
        const unsigned int pivot[2] = {uniform(ol_pivot_key(h0, a.prune, have_state && a.use_hint)),
                                       uniform(ol_pivot_key(h1, a.prune, have_state && a.use_hint))};
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const float xs = side ? -mn : mx;                    // a poisoned token is NaN on both sides
            const bool isnan = xs != xs;
            const bool cand = mine && !isnan && abs_key(xs) >= pivot[side];
            if (mine) S.val[side][w * kTokPerWave + lane] = xs;
            m[side] = (static_cast<unsigned int>(__ballot(cand)) & 0xfu) << (w * kTokPerWave);
            nc[side] = ordered_bits(wave_max((mine && !isnan && !cand) ? xs : -__builtin_inff()));
            if (wave_any(mine && isnan)) wbad |= 1u << side;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the token arrays are in memory before the record says so
    }
    if (lane == 0) {                                             // every wave leaves its part: nothing to initialise, no atomics
        S.mask[0][w] = m[0]; S.mask[1][w] = m[1];
        S.nc[0][w] = nc[0]; S.nc[1][w] = nc[1];
        S.bad[w] = wbad;
    }
    __syncthreads();
    if (tid < 2) {
        const int side = tid;
        const unsigned int tag = epoch + 1u;
        const unsigned int mask = S.mask[side][0] | S.mask[side][1] | S.mask[side][2] | S.mask[side][3];
        const unsigned int ncm = max(max(S.nc[side][0], S.nc[side][1]), max(S.nc[side][2], S.nc[side][3]));
        const unsigned int badm = S.bad[0] | S.bad[1] | S.bad[2] | S.bad[3];
        float first[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned int rest = mask;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (rest) {
                const int pos = __builtin_ctz(rest);
                rest &= rest - 1u;
                first[i] = S.val[side][pos];
            }
        }
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(&a.st->rec[side][0][0], 0, static_cast<int>(sizeof(a.st->rec[side])), 0x00020000);
        osq_v4u32 ha, hb;
        ha.x = tag; ha.y = mask | (((badm >> side) & 1u) << 16); ha.z = __float_as_uint(from_ordered_bits(ncm)); ha.w = __float_as_uint(first[0]);
        hb.x = tag; hb.y = __float_as_uint(first[1]); hb.z = __float_as_uint(first[2]); hb.w = __float_as_uint(first[3]);
        const unsigned int off = static_cast<unsigned int>(group) * 32u;
        __builtin_amdgcn_raw_buffer_store_b128(ha, rs, off, 0, 16 /* sc1 */);
        __builtin_amdgcn_raw_buffer_store_b128(hb, rs, off + 16u, 0, 16 /* sc1 */);
    }
#ifdef OSQ_FINAL_TIMING
    if (tid == 0 && (blockIdx.y == 1 || blockIdx.y == gridDim.y - 1 || blockIdx.y == gridDim.y / 2) && blockIdx.x < 2)
        printf("[onelaunch] stream block (%u,%u) start %lld end %lld (%.2f us)\n", blockIdx.x, blockIdx.y, dbg_t0, wall_clock64(), (wall_clock64() - dbg_t0) / 100.0);
#endif
}

// ---------------------------------------------------------------- selecting workgroup (blockIdx = (side, 0))

// The values a search walks: the selector's LDS list (candidates) or every valid token of the side in memory.
struct OlSource {
    bool memory;
    const OlSelShared* S;
    unsigned int count;           // list entries
    __amdgpu_buffer_rsrc_t tok;   // the side's token array
    unsigned int slots, tokens, chunks;
    unsigned int flip;
    template <typename F>
    __device__ __forceinline__ void each(F f) const {
        if (!memory) {
            for (unsigned int i = threadIdx.x; i < count; i += kOlThreads) f(__uint_as_float(S->list[i]));
        } else {
            for (unsigned int i = threadIdx.x; i < slots; i += kOlThreads) {
                const unsigned int bb = (i >> 4) / chunks, t = i - bb * tokens;
                if (static_cast<int>(t) < S->lens[bb])
                    f(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(tok, i * 4u, 0, 16 /* sc1 */) ^ flip));
            }
        }
    }
};

// max(v[v <= quantile(|v|, q)]) of the source's values; N valid values of which `below` lie under the source (the
// non-candidates: all smaller than every source value).  hist: the first level is already in S.hist (window win_lo /
// 2^23 keys / bins of 2^13).  *missed (uniform): the wanted rank is not in the source / not in the window -- nothing decided.
__device__ __forceinline__ float ol_search(const OlSource& src, OlSelShared& S, const unsigned int N, const unsigned int below,
                                           const float aq, const bool prehist, const unsigned int win_lo, unsigned int kmin,
                                           unsigned int kmax, const float others_max, const int use_shortcut, bool* missed) {
    const int tid = threadIdx.x, lane = tid & (OSQ_WAVE - 1), wv = tid / OSQ_WAVE;
    *missed = false;
    const float rank = aq * static_cast<float>(N - 1u);
    const float rlo = floorf(rank);
    const unsigned int k_lo = static_cast<unsigned int>(rlo), k_hi = static_cast<unsigned int>(ceilf(rank));
    const float w = rank - rlo;
    if (below > k_lo) { *missed = true; return 0.0f; }
    unsigned int sel_lo, sel_width, sel_shift, sel_rank = k_lo - below, sel_le = below, sel_count = N - below;
    if (prehist) { sel_lo = win_lo; sel_width = 1u << 23; sel_shift = 23 - kOlBinBits; }
    else { sel_lo = kmin; sel_width = kmax - kmin + 1u; sel_shift = ol_level_shift(sel_width); }
    bool done = false, listed = false;
    for (int level = 0; level < 5 && !done; ++level) {
        if (level > 0 || !prehist) {
            if (level > 0 && sel_count <= kOlPickCap) { listed = true; break; }
            for (int k = tid; k < kOlBins; k += kOlThreads) S.hist[k] = 0u;
            __syncthreads();
            src.each([&](const float x) {
                const unsigned int d = abs_key(x) - sel_lo;
                if (d < sel_width) atomicAdd(&S.hist[d >> sel_shift], 1u);
            });
            __syncthreads();
        }
        const uint4 hh = *reinterpret_cast<const uint4*>(&S.hist[4 * tid]);
        const unsigned int sum4 = hh.x + hh.y + hh.z + hh.w;
        const unsigned int incl_w = wave_inclusive_scan_u32(sum4);
        if (lane == OSQ_WAVE - 1) S.w_tot[wv] = incl_w;
        __syncthreads();
        const uint4 t4 = *reinterpret_cast<const uint4*>(&S.w_tot[0]);
        const unsigned int t[4] = {t4.x, t4.y, t4.z, t4.w};
        unsigned int base = 0u, inside = 0u;
#pragma unroll
        for (int e = 0; e < kOlWaves; ++e) { base += (e < wv) ? t[e] : 0u; inside += t[e]; }
        if (sel_rank >= inside) {                       // uniform.  Level 0 of a window: the rank lies above it
            *missed = true;
            return 0.0f;
        }
        const unsigned int incl = base + incl_w, excl = incl - sum4;
        if (sel_rank >= excl && sel_rank < incl) {      // exactly one thread
            const unsigned int h[4] = {hh.x, hh.y, hh.z, hh.w};
            unsigned int run = excl, bin = 0u, cnt = 0u, bel = 0u;
            bool found = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (!found && sel_rank < run + h[e]) { found = true; bin = 4u * tid + e; cnt = h[e]; bel = run; }
                run += h[e];
            }
            uint4 pk;
            pk.x = bin; pk.y = bel; pk.z = cnt; pk.w = 0u;
            *reinterpret_cast<uint4*>(&S.sel[0]) = pk;
        }
        __syncthreads();
        const uint4 pk = *reinterpret_cast<const uint4*>(&S.sel[0]);
        const unsigned int bin = uniform(pk.x), below_b = uniform(pk.y), cnt = uniform(pk.z);
        const unsigned int off = bin << sel_shift;
        sel_lo += off;
        sel_count = cnt;
        if (sel_shift == 0u) {                          // single-key bins: found
            sel_le += below_b + cnt;
            sel_width = 0u;
            done = true;
        } else {
            const unsigned int rest = sel_width - off, cap = 1u << sel_shift;
            sel_le += below_b;
            sel_rank -= below_b;
            sel_width = rest < cap ? rest : cap;
            sel_shift = ol_level_shift(sel_width);
        }
        if (!done && sel_count <= kOlPickCap) { listed = true; break; }
    }
    unsigned int v_lo, v_hi;
    float result = 0.0f;
    bool shortcut_done = false;
    if (!done && listed) {
        const unsigned int lo = sel_lo, wd = sel_width, edge = lo + wd;
        const bool need_next = (k_hi != k_lo) && (sel_rank + 1u >= sel_count);
        unsigned int nx = 0xffffffffu;
        src.each([&](const float x) {
            const unsigned int key = abs_key(x);
            if (key - lo < wd) S.pick[atomicAdd(&S.s_fill, 1u)] = __float_as_uint(x);     // sign kept: see the shortcut
            if (need_next) nx = min(nx, key - edge);                                       // keys below the edge wrap to huge values
        });
        if (need_next) {
            nx = wave_min_u32(nx);
            if (lane == 0 && nx < 0x80000000u) atomicMin(&S.s_next, nx + edge);
        }
        __syncthreads();
        const unsigned int cnt = S.s_fill, want = sel_rank;
        for (unsigned int e = tid; e < cnt; e += kOlThreads) {
            const unsigned int ent = S.pick[e], mine = ent & 0x7fffffffu;
            unsigned int lt = 0u, le = 0u;
            for (unsigned int j = 0; j < cnt; j += 4u) {            // entries at or beyond cnt are stale, never counted
                const uint4 o4 = *reinterpret_cast<const uint4*>(&S.pick[j]);
                const unsigned int o[4] = {o4.x & 0x7fffffffu, o4.y & 0x7fffffffu, o4.z & 0x7fffffffu, o4.w & 0x7fffffffu};
#pragma unroll
                for (unsigned int u = 0; u < 4u; ++u) {
                    const bool in = j + u < cnt;
                    lt += (in && o[u] < mine) ? 1u : 0u;
                    le += (in && o[u] <= mine) ? 1u : 0u;
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
        __syncthreads();
        const uint4 fnd = *reinterpret_cast<const uint4*>(&S.s_found[0]);          // s_found[0], s_found[1], s_next, s_pos
        v_lo = fnd.x;
        const bool hi_listed = fnd.y != 0xffffffffu;
        v_hi = hi_listed ? fnd.y : fnd.z;
        // the shortcut of token_select.h: the keys at ranks floor / ceil are neighbours in sorted order, so if an element
        // with the lower key is non-negative the answer is that key -- or the upper one when thr reaches it and a
        // non-negative element has it; both facts are in the pick list as long as the upper key is listed or not reached
        if (use_shortcut) {
            const unsigned int pos = fnd.w;
            const float lo_f = __uint_as_float(v_lo), hi_f = __uint_as_float(k_hi == k_lo ? v_lo : v_hi);
            const float d = hi_f - lo_f;
            const float tt = (w < 0.5f) ? __builtin_fmaf(w, d, lo_f) : __builtin_fmaf(w - 1.0f, d, hi_f);
            const bool reaches_hi = hi_f > lo_f && tt >= hi_f;
            if ((pos & 1u) && (!reaches_hi || hi_listed)) {
                shortcut_done = true;
                result = (reaches_hi && (pos & 2u)) ? hi_f : lo_f;
            }
        }
    } else {
        // every level ran (massive duplicates): sel_lo is the key at rank k_lo, sel_le = #keys <= it
        v_lo = sel_lo;
        if (k_hi != k_lo && sel_le <= k_hi) {                      // rank k_hi is the smallest key above
            unsigned int nx = 0xffffffffu;
            src.each([&](const float x) {
                const unsigned int key = abs_key(x);
                if (key > v_lo) nx = min(nx, key);
            });
            nx = wave_min_u32(nx);
            if (lane == 0) atomicMin(&S.s_next, nx);
            __syncthreads();
            v_hi = S.s_next;
        } else {
            v_hi = v_lo;
        }
    }
    if (k_hi == k_lo) v_hi = v_lo;
    if (!shortcut_done) {
        const float lo_v = __uint_as_float(v_lo), hi_v = __uint_as_float(v_hi), diff = hi_v - lo_v;
        float thr = (w < 0.5f) ? __builtin_fmaf(w, diff, lo_v) : __builtin_fmaf(w - 1.0f, diff, hi_v);   // torch lerp
        thr = __uint_as_float(uniform(__float_as_uint(thr)));
        float best = -__builtin_inff();
        src.each([&](const float x) { best = (x <= thr) ? fmaxf(best, x) : best; });
        // the values below a list source (non-candidates) are all smaller in magnitude than the pivot, hence <= thr
        if (!src.memory) best = fmaxf(best, others_max);
        best = wave_max(best);
        if (lane == 0) atomicMax(&S.s_best, ordered_bits(best));
        __syncthreads();
        result = from_ordered_bits(S.s_best);
    }
    return result;
}

__device__ __forceinline__ void ol_reset_search(OlSelShared& S) {
    if (threadIdx.x == 0) { S.s_fill = 0u; S.s_next = 0xffffffffu; S.s_found[0] = S.s_found[1] = 0xffffffffu; S.s_pos = 0u; S.s_best = 0u; }
    __syncthreads();
}

__device__ __forceinline__ void ol_select(const OneLaunchArgs& a, const Finish& fin, OlSelShared& S) {
    const int side = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & (OSQ_WAVE - 1), wv = tid / OSQ_WAVE;
#ifdef OSQ_FINAL_TIMING
    long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned int dbg_sweeps = 0u;
#define OSQ_OLSTAMP(k) do { ts[k] = wall_clock64(); } while (0)
#else
#define OSQ_OLSTAMP(k) do { } while (0)
#endif
    OSQ_OLSTAMP(0);
    const unsigned int B = static_cast<unsigned int>(a.v.batch), T = static_cast<unsigned int>(a.v.tokens);
    const unsigned int chunks = static_cast<unsigned int>(a.chunks), groups = B * chunks;
    const unsigned int flip = side ? 0x80000000u : 0u;
    // ---- lengths, N, set-up
    unsigned int n_local = 0u;
    for (unsigned int bb = tid; bb < B; bb += kOlThreads) {
        int64_t l = a.lengths ? a.lengths[bb] : static_cast<int64_t>(T);
        l = l < 0 ? 0 : (l > static_cast<int64_t>(T) ? static_cast<int64_t>(T) : l);
        S.lens[bb] = static_cast<int>(l);
        n_local += static_cast<unsigned int>(l);
    }
    n_local = wave_inclusive_scan_u32(n_local);
    if (lane == OSQ_WAVE - 1) S.w_n[wv] = n_local;
    for (int k = tid; k < kOlBins; k += kOlThreads) S.hist[k] = 0u;
    if (tid == 0) { S.s_count = 0u; S.s_timeout = 0u; S.s_fill = 0u; S.s_next = 0xffffffffu; S.s_found[0] = S.s_found[1] = 0xffffffffu; S.s_pos = 0u; S.s_best = 0u; }
    const unsigned int tag = uniform(__hip_atomic_load(&a.st->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1u;
    const bool have_state = fin.rule != OSQ_UPDATE_NONE && fin.min_val && fin.max_val;
    float st_min = 0.f, st_max = 0.f;
    if (have_state) { st_min = fin.min_val[0]; st_max = fin.max_val[0]; }
    const unsigned int pivot = uniform(ol_pivot_key(side ? -st_min : st_max, a.prune, have_state && a.use_hint));
    const bool window = a.prune && pivot != 0u && pivot != kOlNoCandidates;      // S.hist: one octave above the pivot, bins of 2^13 keys
    __syncthreads();
    const unsigned int N = S.w_n[0] + S.w_n[1] + S.w_n[2] + S.w_n[3];
    if (N == 0u) {                                                 // nothing observed, nothing updated (both sides agree)
        if (side == 0 && tid == 0) __hip_atomic_store(&a.st->epoch, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // ---- gather.  Thread t owns groups j * 256 + t (j < per <= 16; bit j of `pending`: that group still has to arrive):
    // a wave's 64 groups of one j are neighbours in dispatch order, so they tend to arrive together -- and the late
    // sweeps only revisit the last j.  Everything below the uniform `wave_any` tests is per lane: LDS atomics on one
    // address are serialised by the LDS itself at a lane per clock, cheaper than a ballot-and-prefix per value.
    const unsigned int per = (groups + kOlThreads - 1) / kOlThreads;
    unsigned int pending = 0u;
    for (unsigned int j = 0; j < per; ++j) {
        const unsigned int g = j * kOlThreads + static_cast<unsigned int>(tid);
        if (g < groups) {
            const unsigned int bb = g / chunks, c = g - bb * chunks;
            if (static_cast<int>(c * kOlGroupTokens) < S.lens[bb]) pending |= 1u << j;
        }
    }
    const auto recs = __builtin_amdgcn_make_buffer_rsrc(&a.st->rec[side][0][0], 0, static_cast<int>(sizeof(a.st->rec[side])), 0x00020000);
    const auto toks = __builtin_amdgcn_make_buffer_rsrc(side ? a.tok_min : a.tok_max, 0, static_cast<int>(B * T * 4u), 0x00020000);
    float plain = -__builtin_inff(), others = -__builtin_inff();
    unsigned int kmin = 0xffffffffu, kmax = 0u;
    bool bad = false, timed_out = false;
    auto absorb = [&](const float x) {                             // one candidate of this lane
        const unsigned int key = abs_key(x);
        plain = fmaxf(plain, x);
        kmin = min(kmin, key);
        kmax = max(kmax, key);
        const unsigned int pos = atomicAdd(&S.s_count, 1u);
        if (pos < static_cast<unsigned int>(kOlListCap)) S.list[pos] = __float_as_uint(x);
        if (window) {
            const unsigned int d = key - pivot;
            if (d < (1u << 23)) atomicAdd(&S.hist[d >> (23 - kOlBinBits)], 1u);
        }
    };
    auto take = [&](const unsigned int j, const bool want, const osq_v4u32& ra, const osq_v4u32& rb) -> bool {
        const bool ready = want && ra.x == tag && rb.x == tag;      // both halves carry this launch's tag
        if (ready) {
            pending &= ~(1u << j);
            bad |= ((ra.y >> 16) & 1u) != 0u;
            others = fmaxf(others, __uint_as_float(ra.z));
            const unsigned int mask = ra.y & 0xffffu;
            const unsigned int cnt = static_cast<unsigned int>(__builtin_popcount(mask));
            if (cnt <= 4u) {
                if (cnt > 0u) absorb(__uint_as_float(ra.w));
                if (cnt > 1u) absorb(__uint_as_float(rb.y));
                if (cnt > 2u) absorb(__uint_as_float(rb.z));
                if (cnt > 3u) absorb(__uint_as_float(rb.w));
            } else {                                               // more than four candidates: picked from the token array
                const unsigned int g = j * kOlThreads + static_cast<unsigned int>(tid);
                osq_v4u32 tk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) tk[u] = __builtin_amdgcn_raw_buffer_load_b128(toks, g * 64u + u * 16u, 0, 16 /* sc1 */);
#pragma unroll
                for (int k = 0; k < kOlGroupTokens; ++k) {
                    const unsigned int raw = (k & 3) == 0 ? tk[k >> 2].x : (k & 3) == 1 ? tk[k >> 2].y : (k & 3) == 2 ? tk[k >> 2].z : tk[k >> 2].w;
                    if ((mask >> k) & 1u) absorb(__uint_as_float(raw ^ flip));
                }
            }
        }
        return ready;
    };
    unsigned int spins = 0u;
    OSQ_OLSTAMP(1);
    while (wave_any(pending != 0u)) {
        if (++spins > a.spin_limit) { timed_out = true; break; }
        bool progress = false;
#ifdef OSQ_FINAL_TIMING
        ++dbg_sweeps;
#endif
        for (unsigned int j = 0; j < per; j += 2u) {                // two groups' records in flight per lane
            const bool w0 = ((pending >> j) & 1u) != 0u, w1 = ((pending >> (j + 1u)) & 1u) != 0u;
            if (!wave_any(w0 || w1)) continue;
            osq_v4u32 a0 = {0u, 0u, 0u, 0u}, b0 = a0, a1 = a0, b1 = a0;
            const unsigned int off0 = (j * kOlThreads + static_cast<unsigned int>(tid)) * 32u, off1 = off0 + kOlThreads * 32u;
            if (w0) { a0 = __builtin_amdgcn_raw_buffer_load_b128(recs, off0, 0, 16 /* sc1 */); b0 = __builtin_amdgcn_raw_buffer_load_b128(recs, off0 + 16u, 0, 16); }
            if (w1) { a1 = __builtin_amdgcn_raw_buffer_load_b128(recs, off1, 0, 16 /* sc1 */); b1 = __builtin_amdgcn_raw_buffer_load_b128(recs, off1 + 16u, 0, 16); }
            const bool r0 = take(j, w0, a0, b0), r1 = take(j + 1u, w1, a1, b1);
            progress |= wave_any(r0 || r1);
        }
        if (!progress) __builtin_amdgcn_s_sleep(2);
    }
    OSQ_OLSTAMP(2);
    // ---- fold the waves' partials
    plain = wave_max(plain);
    others = wave_max(others);
    kmin = wave_min_u32(kmin);
    kmax = wave_max_u32(kmax);
    const bool wbad = wave_any(bad), wto = wave_any(timed_out);
    if (lane == 0) {
        S.w_plain[wv] = ordered_bits(plain);
        S.w_nc[wv] = ordered_bits(others);
        S.w_kmin[wv] = kmin;
        S.w_kmax[wv] = kmax;
        S.w_bad[wv] = wbad ? 1u : 0u;
        if (wto) S.s_timeout = 1u;
    }
    __syncthreads();
    unsigned int o_plain = 0u, o_nc = 0u, any_bad_u = 0u;
    kmin = 0xffffffffu;
    kmax = 0u;
#pragma unroll
    for (int k = 0; k < kOlWaves; ++k) {
        o_plain = max(o_plain, S.w_plain[k]);
        o_nc = max(o_nc, S.w_nc[k]);
        kmin = min(kmin, S.w_kmin[k]);
        kmax = max(kmax, S.w_kmax[k]);
        any_bad_u |= S.w_bad[k];
    }
    const bool gave_up = uniform(S.s_timeout) != 0u;
    const bool any_bad = uniform(any_bad_u) != 0u || gave_up;
    const unsigned int M = uniform(S.s_count);
    const float others_max = from_ordered_bits(uniform(o_nc));
    kmin = uniform(kmin);
    kmax = uniform(kmax);
    float result = fmaxf(from_ordered_bits(uniform(o_plain)), others_max);        // the plain maximum of the side
    if (gave_up && tid == 0) __hip_atomic_fetch_or(&a.st->status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    OSQ_OLSTAMP(3);
#ifdef OSQ_FINAL_TIMING
    int dbg_mode = 0;
#endif
    if (a.prune && !any_bad) {
        const bool list_ok = M <= static_cast<unsigned int>(kOlListCap);
        OlSource src{false, &S, M, toks, B * T, T, chunks, flip};
        bool missed = true;
        if (list_ok && window) result = ol_search(src, S, N, N - M, a.q, true, pivot, kmin, kmax, others_max, a.shortcut, &missed);
        if (missed && list_ok && M > 0u) {              // the rank lies above the window, or there is no window: full-range levels over the list
#ifdef OSQ_FINAL_TIMING
            dbg_mode = 1;
#endif
            ol_reset_search(S);
            result = ol_search(src, S, N, N - M, a.q, false, 0u, kmin, kmax, others_max, a.shortcut, &missed);
        }
        if (missed) {                                   // the rank lies among the non-candidates, or the list overflowed: the token array
#ifdef OSQ_FINAL_TIMING
            dbg_mode = 2;
#endif
            ol_reset_search(S);
            src.memory = true;
            unsigned int lo = 0xffffffffu, hi = 0u;
            src.each([&](const float x) { const unsigned int key = abs_key(x); lo = min(lo, key); hi = max(hi, key); });
            lo = wave_min_u32(lo);
            hi = wave_max_u32(hi);
            if (lane == 0) { S.w_kmin[wv] = lo; S.w_kmax[wv] = hi; }
            __syncthreads();
            lo = min(min(S.w_kmin[0], S.w_kmin[1]), min(S.w_kmin[2], S.w_kmin[3]));
            hi = max(max(S.w_kmax[0], S.w_kmax[1]), max(S.w_kmax[2], S.w_kmax[3]));
            result = ol_search(src, S, N, 0u, a.q, false, 0u, uniform(lo), uniform(hi), others_max, a.shortcut, &missed);
        }
    }
    OSQ_OLSTAMP(4);
    if (tid == 0) {
        const SideResult r{result, any_bad, false};
        float cur_min, cur_max;
        if (meet_sides(&a.meet[0], side, r, &cur_min, &cur_max)) {
            finish_entry(fin, 0, cur_min, cur_max, have_state, st_min, st_max);
            __hip_atomic_store(&a.st->epoch, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#ifdef OSQ_FINAL_TIMING
    OSQ_OLSTAMP(5);
    if (tid == 0)
        printf("[onelaunch] side %d entry %lld: setup %.2f gather %.2f (%u sweeps, per %u) fold %.2f search %.2f (mode %d, N %u, M %u) meet %.2f total %.2f us\n",
               side, ts[0], (ts[1] - ts[0]) / 100.0, (ts[2] - ts[1]) / 100.0, dbg_sweeps, per, (ts[3] - ts[2]) / 100.0, (ts[4] - ts[3]) / 100.0,
               dbg_mode, N, M, (ts[5] - ts[4]) / 100.0, (ts[5] - ts[0]) / 100.0);
#endif
#undef OSQ_OLSTAMP
}

template <bool SINGLE_SEGMENT, bool NT>
__global__ __launch_bounds__(kOlThreads) void observe_tokens_onelaunch_kernel(OneLaunchArgs a, Finish fin) {
    __shared__ OlShared S;
    if (a.phase == 2) {                       // grid (2, 1): the selectors of a two-launch run find every record in place
        ol_select(a, fin, S.sel);
        return;
    }
    if (blockIdx.y == 0) {
        if (blockIdx.x < 2 && a.phase == 0) ol_select(a, fin, S.sel);
        return;
    }
    ol_stream<SINGLE_SEGMENT, NT>(a, fin, S.str);
}

}  // namespace osq
