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
#define OSQ_SSTAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && fin.cur) reinterpret_cast<long long*>(fin.cur + 2)[k] = __builtin_readcyclecounter(); } while (0)
#else
#define OSQ_SSTAMP(k) do { } while (0)
#endif

// One CU handles all of a side's values (<= 32768) at 64 lane-operations per clock, so every VALU
// instruction per value costs ~0.2 us: the passes below are written to a handful of instructions each
// (validity as one compare against a per-group count, NaN poisoning as one select, range through
// float min/max of |v| with source modifiers, the NaN flag as a scalar lane-mask OR).
// R4 = 16-byte groups per thread; group g = tid + 1024*j covers slots 4g .. 4g+3.
template <int R4>
__global__ __launch_bounds__(kSelThreads) void token_select_kernel(SelectArgs a, Finish fin, FinalBatch fb) {
    constexpr int R = 4 * R4;
    const int side = blockIdx.x;
    const int64_t p = blockIdx.y;
    const float* src = side ? a.tok_min : a.tok_max;
    const int64_t* lengths = a.lengths;
    int prune = a.prune;
    if (fb.n_batches > 0) {
        const int64_t qi = p / fb.n_batches, bi = p - qi * fb.n_batches;
        src += p * fb.problem_stride;
        if (lengths) lengths += bi * a.B;
        prune = fb.prune_flags ? fb.prune_flags[qi] : prune;
        fin.cur += 2 * (bi * fb.n_quantizers + qi);
    }
    __shared__ unsigned int hist[kSelBins];
    __shared__ unsigned int list[kListCap];
    __shared__ SelState sel;
    __shared__ unsigned int s_n, s_bad, s_kmin, s_kmax, s_plain, s_fill, s_next, s_found[2], s_sel, s_pos;
    __shared__ unsigned int s_wtot[kSelWaves];

    const int tid = threadIdx.x, lane = tid & (OSQ_WAVE - 1), wv = tid / OSQ_WAVE;
    const unsigned int Tu = static_cast<unsigned int>(a.T);
    const unsigned int groups = static_cast<unsigned int>((a.B * a.T) >> 2);
    const unsigned int Bm1 = static_cast<unsigned int>(a.B) - 1u;
    const bool straddle = (Tu & 3u) != 0u;          // T % 4 == 0: the four slots of a group share their sample
    const unsigned int flip = side ? 0x80000000u : 0u;     // side 1 works on -token_min

    // running state for the finish step, fetched now so that the second arriver's tail has no dependent load
    float st_min = 0.f, st_max = 0.f;
    const bool have_state = fin.rule != OSQ_UPDATE_NONE && fin.min_val && fin.max_val;
    if (tid == 0 && have_state) { st_min = fin.min_val[0]; st_max = fin.max_val[0]; }

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
            int64_t la = a.T, lb = a.T;
            if (lengths) {
                la = lengths[bb < Bm1 ? bb : Bm1];
                lb = straddle ? lengths[bb + 1u < Bm1 ? bb + 1u : Bm1] : la;
            }
            const int ia = la > a.T ? static_cast<int>(a.T) : (la < 0 ? 0 : static_cast<int>(la));
            const int ib = lb > a.T ? static_cast<int>(a.T) : (lb < 0 ? 0 : static_cast<int>(lb));
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
    for (int k = tid; k < kSelBins; k += kSelThreads) hist[k] = 0u;
    if (tid == 0) {
        s_n = 0u; s_bad = 0u; s_kmin = 0xffffffffu; s_kmax = 0u; s_plain = 0u; s_fill = 0u;
        s_next = 0xffffffffu; s_found[0] = s_found[1] = 0xffffffffu; s_sel = 0u; s_pos = 0u;
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
    {
        amin = wave_min(amin);
        amax = wave_max(amax);
        plain = wave_max(plain);
        const bool wbad = wave_any(bad);
        n = wave_inclusive_scan_u32(n);
        __syncthreads();                       // LDS initialisation above is complete
        if (lane == OSQ_WAVE - 1) atomicAdd(&s_n, n);
        if (lane == 0) {
            if (wbad) atomicOr(&s_bad, 1u);
            atomicMin(&s_kmin, __float_as_uint(amin));       // non-negative floats order like their bit patterns
            atomicMax(&s_kmax, __float_as_uint(amax));
            atomicMax(&s_plain, ordered_bits(plain));
        }
    }
    __syncthreads();
    OSQ_SSTAMP(2);
    const unsigned int N = s_n;
    if (N == 0u) return;                       // both sides agree: nothing observed, nothing updated
    const bool any_bad = s_bad != 0u;
    float result = from_ordered_bits(s_plain);

    if (prune && !any_bad) {
        const float rank = a.q * static_cast<float>(N - 1u);
        const float rlo = floorf(rank);
        const unsigned int k_lo = static_cast<unsigned int>(rlo);
        const unsigned int k_hi = static_cast<unsigned int>(ceilf(rank));
        const float w = rank - rlo;
        if (tid == 0) {
            sel.lo = s_kmin;
            sel.width = s_kmax - s_kmin + 1u;
            sel.rank = k_lo;
            sel.shift = level_shift(sel.width);
            sel.le = 0u;
            sel.done = 0u;
            sel.count = N;
        }
        __syncthreads();
        // ---- histogram levels: level 0 always; 1-2 only while the chosen bin is too crowded for the list
        bool listed = false;
        for (int level = 0; level < 3; ++level) {
            if (sel.done) break;
            if (level > 0) {
                if (sel.count <= kListCap) { listed = true; break; }
                for (int k = tid; k < kSelBins; k += kSelThreads) hist[k] = 0u;
                __syncthreads();
            }
            const unsigned int lo = uniform(sel.lo), wd = uniform(sel.width), sh = uniform(sel.shift);
            // The range check also keeps poisoned slots (key 0x7fc00000) out.  Measured alternatives: all of them
            // into ONE trash bin is 5x slower (same-address LDS atomics serialise); one trash bin per lane with a
            // v_min instead of the compare + exec masking is no faster (6.6k vs 6.4k cycles at 32768 slots) --
            // the pass is bound by the LDS atomic rate (~10 clocks per wave instruction), and masking does fewer.
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const unsigned int d = abs_key(v[i]) - lo;
                if (d < wd) atomicAdd(&hist[d >> sh], 1u);
            }
            __syncthreads();
            // block-wide scan over the 2048 bins (2 per thread); the thread whose bins straddle the rank narrows
            const unsigned int h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
            const unsigned int incl_w = wave_inclusive_scan_u32(h0 + h1);
            if (lane == OSQ_WAVE - 1) s_wtot[wv] = incl_w;
            __syncthreads();
            unsigned int base = 0u;
#pragma unroll
            for (int k = 0; k < kSelWaves; ++k) base += (k < wv) ? s_wtot[k] : 0u;
            const unsigned int incl = base + incl_w, excl = incl - (h0 + h1);
            const unsigned int want = sel.rank;
            __syncthreads();
            if (want >= excl && want < incl) {     // exactly one thread
                const bool second = want >= excl + h0;
                const unsigned int below = second ? excl + h0 : excl;
                const unsigned int bin = 2u * tid + (second ? 1u : 0u), cnt = second ? h1 : h0;
                const unsigned int shv = sel.shift, off = bin << shv;
                sel.lo += off;
                sel.count = cnt;
                if (shv == 0u) {                   // single-key bins: found
                    sel.le += below + cnt;
                    sel.width = 0u;
                    sel.done = 1u;
                } else {
                    const unsigned int rest = sel.width - off, cap = 1u << shv;
                    sel.le += below;
                    sel.rank = want - below;
                    sel.width = rest < cap ? rest : cap;
                    sel.shift = level_shift(sel.width);
                }
            }
            __syncthreads();
        }
        OSQ_SSTAMP(3);
        unsigned int v_lo, v_hi;     // keys at floor(rank) / ceil(rank)
        bool shortcut_done = false;  // uniform
        if (!sel.done && listed) {
            // ---- compact the chosen bin's keys; the smallest key above the bin only if rank+1 leaves the bin
            const unsigned int lo = uniform(sel.lo), wd = uniform(sel.width);
            const bool need_next = (k_hi != k_lo) && (uniform(sel.rank) + 1u >= uniform(sel.count));
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const unsigned int key = abs_key(v[i]);
                if (key - lo < wd) list[atomicAdd(&s_fill, 1u)] = __float_as_uint(v[i]);    // sign kept: see the shortcut below
            }
            if (need_next) {
                // smallest key at or above the bin's end: keys below it wrap to huge values under the
                // unsigned subtraction and lose the min (poisoned slots are above every valid key)
                const unsigned int edge = lo + wd;
                unsigned int nx = 0xffffffffu;
#pragma unroll
                for (int i = 0; i < R; ++i) nx = min(nx, abs_key(v[i]) - edge);
                nx = wave_min_u32(nx);
                if (lane == 0 && nx < 0x80000000u) atomicMin(&s_next, nx + edge);   // >= 2^31: only wrapped keys in this wave
            }
            __syncthreads();
            const unsigned int cnt = s_fill, want = sel.rank;
            if (static_cast<unsigned int>(tid) < cnt) {     // rank by counting, ties broken by position
                const unsigned int mine = list[tid] & 0x7fffffffu;
                unsigned int r = 0u;
                for (unsigned int j = 0; j < cnt; ++j) {
                    const unsigned int o = list[j] & 0x7fffffffu;
                    r += (o < mine || (o == mine && j < static_cast<unsigned int>(tid))) ? 1u : 0u;
                }
                if (r == want) s_found[0] = mine;
                if (r == want + 1u) s_found[1] = mine;
            }
            __syncthreads();
            v_lo = s_found[0];
            const bool hi_listed = s_found[1] != 0xffffffffu;
            v_hi = hi_listed ? s_found[1] : s_next;
            // Shortcut for the threshold pass.  The keys at ranks floor/ceil are neighbours in sorted order, so
            // no key lies strictly between them, thr lies in [lo_v, hi_v], and every value with a larger key
            // is either negative or above thr.  If some element with key lo_v is non-negative, then
            // max(v[v <= thr]) is lo_v -- or hi_v when thr reaches it and a non-negative element has that
            // key.  Both facts are in the list (it holds every element of the bin, with sign) as long as the
            // upper key is listed or not reached; otherwise the register pass below decides.
            if (a.shortcut) {
                if (static_cast<unsigned int>(tid) < cnt) {
                    const unsigned int e = list[tid];
                    if (!(e >> 31)) {
                        if (e == v_lo) atomicOr(&s_pos, 1u);
                        if (e == v_hi) atomicOr(&s_pos, 2u);
                    }
                }
                __syncthreads();
                const unsigned int pos = s_pos;
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
            // every level ran (massive duplicates): sel.lo is the key at rank k_lo, sel.le = #keys <= it
            v_lo = uniform(sel.lo);
            if (k_hi != k_lo && sel.le <= k_hi) {           // rank k_hi is the smallest key above
                unsigned int nx = 0xffffffffu;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const unsigned int key = abs_key(v[i]);
                    if (key > v_lo) nx = min(nx, key);
                }
                nx = wave_min_u32(nx);
                if (lane == 0) atomicMin(&s_next, nx);
                __syncthreads();
                v_hi = s_next;
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
            for (int i = 0; i < R; ++i) best = (v[i] <= thr) ? fmaxf(best, v[i]) : best;
            best = wave_max(best);
            if (lane == 0) atomicMax(&s_sel, ordered_bits(best));
            __syncthreads();
            result = from_ordered_bits(s_sel);
        }
    }
    OSQ_SSTAMP(5);
    // ---- rendezvous of the two sides: first arriver leaves {value, present | bad}, second finishes
    if (tid == 0) {
        const unsigned long long mine = (static_cast<unsigned long long>(0x80000000u | (any_bad ? 1u : 0u)) << 32) |
                                        __float_as_uint(result);
        const unsigned long long other = __hip_atomic_exchange(&a.meet[p], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (other >> 63) {
            __hip_atomic_store(&a.meet[p], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float theirs = __uint_as_float(static_cast<unsigned int>(other & 0xffffffffull));
            const bool poisoned = any_bad || ((other >> 32) & 1ull);
            const float up = side ? theirs : result;
            const float lo = -(side ? result : theirs);
            float cur_min = (lo > up) ? up : lo;          // aminmax(clip(value, lo, up)), observer.py:68,227
            float cur_max = up;
            if (poisoned) { cur_min = __builtin_nanf(""); cur_max = cur_min; }
            finish_entry(fin, 0, cur_min, cur_max, have_state, st_min, st_max);
        }
    }
    OSQ_SSTAMP(6);
}

}  // namespace osq
