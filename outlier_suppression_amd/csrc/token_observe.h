// The streaming launch of a masked observation (observer.py:50-70, 176-237) that ALSO builds the first level of the
// token-wise-clipping selection.  Included by observer.hip after token_select.h (TokObsState, bucket_window).
//
// Round 4's two launches spent 11-12 us in token_select_kernel: ONE workgroup per side walks all <= 32768 per-token
// extrema (range, histogram with LDS atomics, scan, list, rank) while the rest of the chip idles.  Here every wave of
// the per-token reduction (token_minmax_vec_kernel's, unchanged) files its tokens' extrema as soon as it has them:
//     lane 2k     token k's maximum            -> side 0
//     lane 2k + 1 minus token k's minimum      -> side 1
// each with ONE device-scope returning atomic on the counter of the value's bin -- 2^14 bins over the window
// hint * [1/2, 3/2] round the observer's running statistic, which predicts this batch's threshold (token_select.h, HINT)
// -- and a store of the value into the bin's bucket at the slot the atomic returned (the first 16 arrivals of a bin are
// kept).  Keys above the window bump a sharded counter, keys below it are implied by the number of valid tokens, a NaN
// raises a flag.  Without pruning (AvgMinMaxObserver, MinMaxObserver, AvgPruneMinMaxObserver on attention_probs) the
// wave folds its tokens into 32 sharded per-side maxima instead.  The per-token arrays are written as before, so the
// selecting launch (token_select_kernel with SelectArgs::hist set) can always fall back on its full selection; with
// the buckets it only scans 16384 counters and ranks <= 16 values (select_from_buckets).
// The window comes from the running statistic as it stands BEFORE this batch; the selecting launch derives the same
// window from the same two words before it updates them.
#pragma once

namespace osq {

template <bool SINGLE_SEGMENT, bool NT>
__global__ __launch_bounds__(kThreads) void token_minmax_hist_kernel(const float* __restrict__ x, osq_token_view v,
                                                                     const int64_t* __restrict__ lengths,
                                                                     float* __restrict__ tok_min, float* __restrict__ tok_max,
                                                                     int lgG, int inner4, TokObsState* st, int prune,
                                                                     const float* hint_min, const float* hint_max) {
    const int64_t b = blockIdx.y;
    int64_t len = v.tokens;
    if (lengths) {
        const int64_t l = lengths[b];
        len = l < len ? l : len;
    }
    const int lane = threadIdx.x & (OSQ_WAVE - 1), w = threadIdx.x / OSQ_WAVE;
    const int64_t chunk = (static_cast<int64_t>(blockIdx.x) + blockIdx.y) % gridDim.x;        // XCD balance, see token_minmax_vec_kernel
    const int64_t t0 = chunk * kTokPerBlock + w * kTokPerWave;
    if (t0 >= len) return;
    // the window of each side (uniform: scalar loads that travel under the streaming loads below)
    float hmax = __builtin_nanf(""), hmin = hmax;
    if (prune && hint_min && hint_max) { hmax = hint_max[0]; hmin = hint_min[0]; }
    const int ntok = (len - t0) < kTokPerWave ? static_cast<int>(len - t0) : kTokPerWave;
    const float* base = x + b * v.stride_batch + t0 * v.stride_token;
    MinMax acc[kTokPerWave];
    token_extrema<SINGLE_SEGMENT, NT>(base, v, ntok, lgG, inner4, lane, acc);
    const int k = lane >> 1, sd = lane & 1;
    float mn = acc[0].mn, mx = acc[0].mx;
#pragma unroll
    for (int j = 1; j < kTokPerWave; ++j)
        if (k == j) { mn = acc[j].mn; mx = acc[j].mx; }
    const bool mine = lane < 2 * ntok;
    if (mine) {
        const int64_t slot = b * v.tokens + t0 + k;
        if (sd) tok_min[slot] = mn; else tok_max[slot] = mx;
    }
    const float val = sd ? -mn : mx;
    const unsigned int shard = (blockIdx.y * gridDim.x + blockIdx.x) % kBkShards;
    if (wave_any(mine && (val != val))) {
        if (lane == 0) __hip_atomic_fetch_or(&st->bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (!prune) {
        // plain maxima: lanes of equal parity combine (xor 2, xor 4 within the first eight lanes)
        float m = mine ? val : -__builtin_inff();
        m = fmaxf(m, __shfl_xor(m, 2, OSQ_WAVE));
        m = fmaxf(m, __shfl_xor(m, 4, OSQ_WAVE));
        if (lane < 2) __hip_atomic_fetch_max(&st->plain[sd][shard][0], ordered_bits(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const SelWindow win = bucket_window(__builtin_fabsf(sd ? hmin : hmax), prune);
    if (mine && win.on) {
        const unsigned int key = abs_key(val), d = key - win.lo;
        if (d < win.wd) {
            const unsigned int bin = d >> win.sh;
            const unsigned int slot = __hip_atomic_fetch_add(&st->count[sd][bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (slot < static_cast<unsigned int>(kBkCap)) st->entry[sd][bin][slot] = __float_as_uint(val);
        } else if (key >= win.lo) {
            __hip_atomic_fetch_add(&st->above[sd][shard][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace osq
