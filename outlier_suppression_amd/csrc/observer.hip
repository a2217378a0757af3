// Observer reductions for gfx950 (MI355X): global / per-channel / per-token min-max,
// the token-wise percentile clipping of AvgPruneMinMaxObserver, running statistics and
// calculate_qparams.  Replaces quant_transformer/quantization/observer.py:50-237.
//
// One HBM read of the observed tensor (4 B/elem); padded tokens are never read.
// Wave64 shuffle reductions, LDS across waves, and a last-workgroup finaliser so
// that a whole observer call (reduce -> running statistic -> scale/zero_point) is
// one launch for flat / per-channel tensors and two for masked activations.
#include <cstdlib>
#include <mutex>
#include <string>
#include <hip/hip_ext.h>
#include "osq_device.h"
#include "osq_host.h"

namespace osq {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / OSQ_WAVE;
constexpr int kFinalThreads = 1024;

struct MinMax {
    float mn, mx;
    bool bad;   // a NaN was seen: torch's aminmax / max(dim) propagate it
    __device__ __forceinline__ void init() { mn = __builtin_inff(); mx = -__builtin_inff(); bad = false; }
    __device__ __forceinline__ void add(float v) {
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
        bad |= (v != v);
    }
    __device__ __forceinline__ void add4(const float4& v) { add(v.x); add(v.y); add(v.z); add(v.w); }
    __device__ __forceinline__ void wave_reduce() {
        mn = wave_min(mn);
        mx = wave_max(mx);
        bad = wave_any(bad);
    }
    __device__ __forceinline__ void poison() {
        if (bad) { mn = __builtin_nanf(""); mx = __builtin_nanf(""); }
    }
};

// block-wide combine; valid in thread 0
__device__ __forceinline__ MinMax block_reduce(MinMax v) {
    __shared__ float s_mn[kFinalThreads / OSQ_WAVE], s_mx[kFinalThreads / OSQ_WAVE];
    __shared__ int s_bad[kFinalThreads / OSQ_WAVE];
    v.wave_reduce();
    const int lane = threadIdx.x & (OSQ_WAVE - 1), w = threadIdx.x / OSQ_WAVE;
    const int nw = (blockDim.x + OSQ_WAVE - 1) / OSQ_WAVE;
    __syncthreads();
    if (lane == 0) { s_mn[w] = v.mn; s_mx[w] = v.mx; s_bad[w] = v.bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < nw; ++k) {
            v.mn = fminf(v.mn, s_mn[k]);
            v.mx = fmaxf(v.mx, s_mx[k]);
            v.bad |= (s_bad[k] != 0);
        }
    }
    return v;
}

struct Finish {   // what happens once a batch's (min, max) is known
    int rule;
    int64_t cnt;
    float* min_val;
    float* max_val;
    float* cur;
    int quant_min, quant_max, symmetric;
    float* scale_out;
    void* zp_out;
    int zp_type;
};

// have_state: the caller already holds min_val[idx] / max_val[idx] in (st_min, st_max)
__device__ __forceinline__ void finish_entry(const Finish& f, int64_t idx, float cur_min, float cur_max,
                                             bool have_state = false, float st_min = 0.f, float st_max = 0.f) {
    if (f.cur) { f.cur[2 * idx] = cur_min; f.cur[2 * idx + 1] = cur_max; }
    float mn = cur_min, mx = cur_max;
    if (f.rule != OSQ_UPDATE_NONE && f.min_val && f.max_val) {
        if (!have_state) { st_min = f.min_val[idx]; st_max = f.max_val[idx]; }
        mn = st_min;
        mx = st_max;
        apply_update(f.rule, f.cnt, cur_min, cur_max, &mn, &mx);
        f.min_val[idx] = mn;
        f.max_val[idx] = mx;
    }
    if (f.scale_out) {
        float s, z;
        qparams_from_range(mn, mx, f.quant_min, f.quant_max, f.symmetric, &s, &z);
        f.scale_out[idx] = s;
        if (f.zp_out) store_zp(f.zp_out, f.zp_type, idx, z);
    }
}

// ---------------------------------------------------------------- calculate_qparams / update

__global__ void qparams_kernel(const float* __restrict__ mn, const float* __restrict__ mx, int64_t n, int quant_min,
                               int quant_max, int symmetric, float* __restrict__ scale_out, void* __restrict__ zp_out,
                               int zp_type) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, z;
    qparams_from_range(mn[i], mx[i], quant_min, quant_max, symmetric, &s, &z);
    scale_out[i] = s;
    if (zp_out) store_zp(zp_out, zp_type, i, z);
}

// Replay of gathered per-batch statistics (sharded / cached calibration): quantizer q applies its update rule
// to the rows of table[n_batches, n_q, 2] in batch order -- the reference's sequential running statistic,
// observer.py:143-144 / 194-202 -- and refreshes its (scale, zero_point), all through per-quantizer pointers:
// one launch instead of n_batches + n_q launches and as many tensor copies.
struct ReplayArgs {
    const float* table;
    int n_batches, n_q;
    const int32_t* rules;
    int64_t cnt0;
    int fresh;                         // 1: start from the untouched (+inf, -inf) state whatever the buffers hold
    const uint64_t* min_ptrs;
    const uint64_t* max_ptrs;
    const int32_t* qmin;
    const int32_t* qmax;
    const int32_t* symmetric;
    const uint64_t* scale_ptrs;
    const uint64_t* zp_ptrs;
    const int32_t* zp_types;
};

__global__ void replay_kernel(ReplayArgs a) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.n_q) return;
    float* min_p = reinterpret_cast<float*>(a.min_ptrs[q]);
    float* max_p = reinterpret_cast<float*>(a.max_ptrs[q]);
    float mn = a.fresh ? __builtin_inff() : *min_p;
    float mx = a.fresh ? -__builtin_inff() : *max_p;
    const int rule = a.rules[q];
    for (int b = 0; b < a.n_batches; ++b) {
        const float* cur = a.table + (static_cast<int64_t>(b) * a.n_q + q) * 2;
        apply_update(rule, a.cnt0 + b, cur[0], cur[1], &mn, &mx);
    }
    *min_p = mn;
    *max_p = mx;
    if (a.scale_ptrs && a.scale_ptrs[q]) {
        float s, z;
        qparams_from_range(mn, mx, a.qmin[q], a.qmax[q], a.symmetric[q], &s, &z);
        *reinterpret_cast<float*>(a.scale_ptrs[q]) = s;
        if (a.zp_ptrs[q]) store_zp(reinterpret_cast<void*>(a.zp_ptrs[q]), a.zp_types[q], 0, z);
    }
}

__global__ void update_kernel(const float* __restrict__ cur_min, const float* __restrict__ cur_max, int64_t n, int rule,
                              int64_t cnt, float* __restrict__ min_val, float* __restrict__ max_val) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    apply_update(rule, cnt, cur_min[i], cur_max[i], &min_val[i], &max_val[i]);
}

// ---------------------------------------------------------------- flat min/max (K4)

__global__ __launch_bounds__(kThreads) void observe_flat_kernel(const float4* __restrict__ x, int64_t n4,
                                                                const float* __restrict__ xt, int tail,
                                                                float* __restrict__ partials,
                                                                unsigned int* __restrict__ counter, Finish fin) {
    MinMax acc;
    acc.init();
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
    int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    // running state for the finish step: fetched now by the thread that may need it, so that the last
    // workgroup's tail has no dependent load left (every serial memory round trip there costs ~1 us)
    float st_min = 0.f, st_max = 0.f;
    if (threadIdx.x == 0 && fin.rule != OSQ_UPDATE_NONE && fin.min_val && fin.max_val) {
        st_min = fin.min_val[0];
        st_max = fin.max_val[0];
    }
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = load_stream(&x[i]), b = load_stream(&x[i + stride]), c = load_stream(&x[i + 2 * stride]),
                     d = load_stream(&x[i + 3 * stride]);
        acc.add4(a); acc.add4(b); acc.add4(c); acc.add4(d);
    }
    for (; i < n4; i += stride) acc.add4(x[i]);
    if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < tail) acc.add(xt[threadIdx.x]);
    acc = block_reduce(acc);
    // ONE 8-byte partial per workgroup: {min, max}, a NaN minimum flags "NaN seen" (fminf never yields one)
    unsigned long long* part64 = reinterpret_cast<unsigned long long*>(partials);
    if (threadIdx.x == 0) {
        const float pm = acc.bad ? __builtin_nanf("") : acc.mn;
        __hip_atomic_store(&part64[blockIdx.x],
                           (static_cast<unsigned long long>(__float_as_uint(acc.mx)) << 32) | __float_as_uint(pm),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (grid_last_block(counter, gridDim.x)) {
        // every load is issued before any is consumed: read in a loop that uses each value at once, the
        // (ordered) agent-scope loads cost one memory round trip per iteration -- most of the old 8 us tail
        constexpr int kPer = kMaxBlocks / kThreads;
        unsigned long long raw[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const unsigned int k = threadIdx.x + j * kThreads;
            raw[j] = __hip_atomic_load(&part64[k < gridDim.x ? k : gridDim.x - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        MinMax t;
        t.init();
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const float pm = __uint_as_float(static_cast<unsigned int>(raw[j] & 0xffffffffull));
            const float px = __uint_as_float(static_cast<unsigned int>(raw[j] >> 32));
            t.mn = fminf(t.mn, pm);
            t.mx = fmaxf(t.mx, px);
            t.bad |= (pm != pm);
        }
        t = block_reduce(t);
        if (threadIdx.x == 0) {
            t.poison();
            finish_entry(fin, 0, t.mn, t.mx, true, st_min, st_max);
            grid_reset(counter, gridDim.x);
        }
    }
}

// ---------------------------------------------------------------- per-channel min/max (K5)

// rows = channels (outer == 1), inner % 4 == 0: one wave per row
__global__ __launch_bounds__(kThreads) void observe_rows_kernel(const float4* __restrict__ x, int64_t rows, int inner4,
                                                                Finish fin) {
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    const int64_t wave = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) / OSQ_WAVE;
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float4* xr = x + r * inner4;
        MinMax acc;
        acc.init();
        int j = lane;
        for (; j + 3 * OSQ_WAVE < inner4; j += 4 * OSQ_WAVE) {
            const float4 a = xr[j], b = xr[j + OSQ_WAVE], c = xr[j + 2 * OSQ_WAVE], d = xr[j + 3 * OSQ_WAVE];
            acc.add4(a); acc.add4(b); acc.add4(c); acc.add4(d);
        }
        for (; j < inner4; j += OSQ_WAVE) acc.add4(xr[j]);
        acc.wave_reduce();
        if (lane == 0) {
            acc.poison();
            finish_entry(fin, r, acc.mn, acc.mx);
        }
    }
}

// generic [outer, channels, inner]: one workgroup per channel
__global__ __launch_bounds__(kThreads) void observe_channels_kernel(const float* __restrict__ x, int64_t outer,
                                                                    int64_t channels, int64_t inner, Finish fin) {
    const int64_t c = blockIdx.x;
    MinMax acc;
    acc.init();
    for (int64_t o = 0; o < outer; ++o) {
        const float* p = x + (o * channels + c) * inner;
        for (int64_t j = threadIdx.x; j < inner; j += kThreads) acc.add(p[j]);
    }
    acc = block_reduce(acc);
    if (threadIdx.x == 0) {
        acc.poison();
        finish_entry(fin, c, acc.mn, acc.mx);
    }
}

// ---------------------------------------------------------------- per-token min/max (K6/K7)

constexpr int kTokPerWave = 4;                                  // tokens a wave keeps in flight together
constexpr int kTokPerBlock = kTokPerWave * kWavesPerBlock;      // consecutive tokens of ONE sample per workgroup

// Fast path: feat_inner contiguous (stride 1), everything 16-byte aligned.
// Workgroup = 16 consecutive tokens of one sample (blockIdx.y = sample); a workgroup that
// starts beyond the sample's valid length exits at once, so padding is never read.
// Each wave owns 4 tokens and issues all their 16-byte loads before reducing any.
// Lane layout inside a token: the wave is split into 64/G groups of G lanes; group g walks
// feature segments g, g + 64/G, ...; G = 64 for one long segment ([B,T,768]), G = 16 for
// head_dim 64 ([B,h,T,64] and its views) so that no lane idles on short segments.
template <bool NT>
__device__ __forceinline__ float4 tok_load(const float4* p) { return NT ? load_stream(p) : *p; }

// The per-token reduction of one wave: the extrema of its (up to) 4 tokens starting at `base`, NaN-poisoned, every
// lane holding every token's result.
template <bool SINGLE_SEGMENT, bool NT>
__device__ __forceinline__ void token_extrema(const float* __restrict__ base, const osq_token_view& v, const int ntok,
                                              const int lgG, const int inner4, const int lane, MinMax (&acc)[kTokPerWave]) {
#pragma unroll
    for (int k = 0; k < kTokPerWave; ++k) acc[k].init();
    if (SINGLE_SEGMENT) {
        // three column steps at a time (768 columns = exactly one trip): 12 independent 16-byte loads per lane
        // are issued before the first is reduced; the short loop below takes what is left
        int j = lane;
        for (; j + 2 * OSQ_WAVE < inner4; j += 3 * OSQ_WAVE) {
            float4 val[3][kTokPerWave];
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int k = 0; k < kTokPerWave; ++k) {
                    const int kk = k < ntok ? k : 0;             // short tail: re-read token 0, result unused
                    val[u][k] = tok_load<NT>(reinterpret_cast<const float4*>(base + kk * v.stride_token) + j + u * OSQ_WAVE);
                }
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int k = 0; k < kTokPerWave; ++k) acc[k].add4(val[u][k]);
        }
        for (; j < inner4; j += OSQ_WAVE) {
            float4 val[kTokPerWave];
#pragma unroll
            for (int k = 0; k < kTokPerWave; ++k) {
                const int kk = k < ntok ? k : 0;
                val[k] = tok_load<NT>(reinterpret_cast<const float4*>(base + kk * v.stride_token) + j);
            }
#pragma unroll
            for (int k = 0; k < kTokPerWave; ++k) acc[k].add4(val[k]);
        }
    } else {
        const int G = 1 << lgG;
        const int grp = lane >> lgG, li = lane & (G - 1), ngrp = OSQ_WAVE >> lgG;
        for (int64_t o = grp; o < v.feat_outer; o += ngrp) {
            const float* seg = base + o * v.stride_outer;
            for (int j = li; j < inner4; j += G) {
                float4 val[kTokPerWave];
#pragma unroll
                for (int k = 0; k < kTokPerWave; ++k) {
                    const int kk = k < ntok ? k : 0;
                    val[k] = tok_load<NT>(reinterpret_cast<const float4*>(seg + kk * v.stride_token) + j);
                }
#pragma unroll
                for (int k = 0; k < kTokPerWave; ++k) acc[k].add4(val[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kTokPerWave; ++k) {
        acc[k].wave_reduce();
        acc[k].poison();
    }
}

template <bool SINGLE_SEGMENT, bool NT>
__global__ __launch_bounds__(kThreads) void token_minmax_vec_kernel(const float* __restrict__ x, osq_token_view v,
                                                                    const int64_t* __restrict__ lengths,
                                                                    float* __restrict__ tok_min,
                                                                    float* __restrict__ tok_max, int lgG, int inner4) {
    const int64_t b = blockIdx.y;
    int64_t len = v.tokens;
    if (lengths) {
        const int64_t l = lengths[b];
        len = l < len ? l : len;
    }
    const int lane = threadIdx.x & (OSQ_WAVE - 1), w = threadIdx.x / OSQ_WAVE;
    // Workgroups reach the XCDs round-robin in linear order (y * gridDim.x + x): with 8 chunks per sample
    // chunk x would always land on XCD x, and since late chunks are mostly padding, XCD 0 would read 7x
    // the bytes of XCD 7 (measured: the 54 %-valid tensor took as long as the full one).  Rotating the
    // chunk index by the sample index spreads every chunk position over all XCDs.
    const int64_t chunk = (static_cast<int64_t>(blockIdx.x) + blockIdx.y) % gridDim.x;
    const int64_t t0 = chunk * kTokPerBlock + w * kTokPerWave;
    if (t0 >= len) return;
    const int ntok = (len - t0) < kTokPerWave ? static_cast<int>(len - t0) : kTokPerWave;
    const float* base = x + b * v.stride_batch + t0 * v.stride_token;
    MinMax acc[kTokPerWave];
    token_extrema<SINGLE_SEGMENT, NT>(base, v, ntok, lgG, inner4, lane, acc);
    if (lane < ntok) {
        float mn = acc[0].mn, mx = acc[0].mx;
#pragma unroll
        for (int k = 1; k < kTokPerWave; ++k)
            if (lane == k) { mn = acc[k].mn; mx = acc[k].mx; }
        const int64_t slot = b * v.tokens + t0 + lane;
        tok_min[slot] = mn;
        tok_max[slot] = mx;
    }
}

// Any strides (scalar loads): correctness path for layouts the fast path rejects.
__global__ __launch_bounds__(kThreads) void token_minmax_generic_kernel(const float* __restrict__ x, osq_token_view v,
                                                                        const int64_t* __restrict__ lengths,
                                                                        float* __restrict__ tok_min,
                                                                        float* __restrict__ tok_max) {
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    const int64_t ntok = v.batch * v.tokens;
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) / OSQ_WAVE;
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
    const int64_t F = v.feat_outer * v.feat_inner;
    for (int64_t tok = wave0; tok < ntok; tok += nwaves) {
        const int64_t b = tok / v.tokens, t = tok - b * v.tokens;
        if (lengths && t >= lengths[b]) continue;
        const float* base = x + b * v.stride_batch + t * v.stride_token;
        MinMax acc;
        acc.init();
        for (int64_t j = lane; j < F; j += OSQ_WAVE) {
            const int64_t o = j / v.feat_inner, i = j - o * v.feat_inner;
            acc.add(base[o * v.stride_outer + i * v.stride_inner]);
        }
        acc.wave_reduce();
        if (lane == 0) {
            acc.poison();
            tok_min[tok] = acc.mn;
            tok_max[tok] = acc.mx;
        }
    }
}

// Many sites, one launch (observer passes of a calibration forward: 98 masked sites for BERT-base, each 4-9 us of
// kernel for 6-50 MB -- launch-bound).  The sites of a forward are recorded while the model runs
// (quantization/deferred.py) and reduced together: wave = one token of one site, found by bisection over the
// table's running token counts; padded tokens are skipped without a read; any layout (16-byte loads when the
// innermost feature axis is contiguous and aligned).
constexpr int kMultiLdsSites = 512;            // running token counts of that many sites are bisected in LDS
__global__ __launch_bounds__(kThreads) void token_minmax_multi_kernel(const osq_site_desc* __restrict__ descs,
                                                                      const int64_t* __restrict__ tok_end, int n_sites,
                                                                      int64_t total_tokens) {
    // A wave's token costs a chain of dependent reads before its data load leaves: the bisection (7 steps for 96 sites),
    // the site's descriptor, the sample's length.  Out of global memory that chain was ~5 us per 3 KiB token (2.6 TB/s at
    // full occupancy); the table now sits in LDS and the descriptor is read through the scalar cache (wave-uniform index).
    __shared__ int64_t s_end[kMultiLdsSites];
    const bool in_lds = n_sites <= kMultiLdsSites;
    if (in_lds) {
        for (int k = threadIdx.x; k < n_sites; k += kThreads) s_end[k] = tok_end[k];
        __syncthreads();
    }
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) / OSQ_WAVE;
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
    for (int64_t g = wave0; g < total_tokens; g += nwaves) {
        int lo = 0, hi = n_sites - 1;                 // first site whose tok_end exceeds g
        if (in_lds) {
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_end[mid] > g) hi = mid; else lo = mid + 1;
            }
        } else {
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (tok_end[mid] > g) hi = mid; else lo = mid + 1;
            }
        }
        lo = __builtin_amdgcn_readfirstlane(lo);      // the wave's token is one: the descriptor load is a scalar load
        const osq_site_desc d = descs[lo];
        const int64_t tok = g - (lo ? (in_lds ? s_end[lo - 1] : tok_end[lo - 1]) : 0);
        const int64_t b = tok / d.view.tokens, t = tok - b * d.view.tokens;
        if (d.lengths && t >= d.lengths[b]) continue;
        const float* base = d.x + b * d.view.stride_batch + t * d.view.stride_token;
        MinMax acc;
        acc.init();
        if (d.vec) {
            const int inner4 = static_cast<int>(d.view.feat_inner / 4);
            const int64_t F4 = d.view.feat_outer * inner4;
            if (d.view.feat_outer == 1) {
                const float4* row = reinterpret_cast<const float4*>(base);
                int64_t j = lane;
                for (; j + 2 * OSQ_WAVE < F4; j += 3 * OSQ_WAVE) {        // 768 features = exactly one trip
                    const float4 a = load_stream(row + j), c = load_stream(row + j + OSQ_WAVE), e = load_stream(row + j + 2 * OSQ_WAVE);
                    acc.add4(a); acc.add4(c); acc.add4(e);
                }
                for (; j < F4; j += OSQ_WAVE) acc.add4(load_stream(row + j));
            } else {
                // head-split views ([B,h,T,64] and its transposes): three independent loads per trip here too
                const int ii = inner4;
                int64_t j = lane;
                for (; j + 2 * OSQ_WAVE < F4; j += 3 * OSQ_WAVE) {
                    const int64_t j1 = j + OSQ_WAVE, j2 = j + 2 * OSQ_WAVE;
                    const int64_t o0 = j / ii, o1 = j1 / ii, o2 = j2 / ii;
                    const float4 a = load_stream(reinterpret_cast<const float4*>(base + o0 * d.view.stride_outer) + (j - o0 * ii));
                    const float4 c = load_stream(reinterpret_cast<const float4*>(base + o1 * d.view.stride_outer) + (j1 - o1 * ii));
                    const float4 e = load_stream(reinterpret_cast<const float4*>(base + o2 * d.view.stride_outer) + (j2 - o2 * ii));
                    acc.add4(a); acc.add4(c); acc.add4(e);
                }
                for (; j < F4; j += OSQ_WAVE) {
                    const int64_t o = j / ii, i = j - o * ii;
                    acc.add4(load_stream(reinterpret_cast<const float4*>(base + o * d.view.stride_outer) + i));
                }
            }
        } else {
            const int64_t F = d.view.feat_outer * d.view.feat_inner;
            for (int64_t j = lane; j < F; j += OSQ_WAVE) {
                const int64_t o = j / d.view.feat_inner, i = j - o * d.view.feat_inner;
                acc.add(base[o * d.view.stride_outer + i * d.view.stride_inner]);
            }
        }
        acc.wave_reduce();
        if (lane == 0) {
            acc.poison();
            d.token_min[tok] = acc.mn;
            d.token_max[tok] = acc.mx;
        }
    }
}

// ---------------------------------------------------------------- token range finaliser (K7b + K8 + K9)

// Single workgroup of 1024 threads.  Valid slots: b*T + t with t < lengths[b] (all if
// lengths == NULL).  Up to SLOTS slots per thread are loaded ONCE into registers (SLOTS = 4,
// 8, 16 or 32 covers batch*tokens <= 4096 ... 32768, e.g. the [256,128,768] benchmark
// tensor); larger inputs (SLOTS = 0) re-read the L2-resident arrays in every pass.
//
// prune: torch.quantile(|token_max|, p) needs the order statistics floor(rank) and
// ceil(rank), rank = fp32(p) * fp32(N-1).  Selection works on the fp32 bit patterns of the
// absolute values (non-negative floats order like their bit patterns):
//   pass 0  N, NaN flag, plain extrema, [min key, max key] of both arrays
//   pass 1  range histogram: 2048 bins spread linearly (power-of-two bin width) over the
//           observed key range -- bins follow the data, so the LDS atomics do not pile up on
//           one exponent bin the way a digit-wise radix pass would; a block-wide DPP scan
//           finds the bin that holds the wanted rank
//   pass 2  the (few) keys of that bin are compacted into an LDS list, together with the
//           smallest key above the bin; the list is ranked by counting, which yields the keys
//           at floor(rank) and ceil(rank) exactly
//           (if a bin holds more than 1024 keys -- massive duplicates -- fall back to two more
//           histogram levels of 11 bits each plus a neighbour pass)
//   pass 3  up = max(token_max[token_max <= upper]), lo = min(token_min[token_min >= lower])
// Interpolation is torch's lerp (one fused multiply-add per branch, pinned in
// tests/test_oracle_pinning.py).
constexpr int kSelBins = 2048;
constexpr int kSelBits = 11;
constexpr int kListCap = 1024;

struct SelState {
    unsigned int lo;       // first key of the current range; the selected key once done
    unsigned int width;    // number of key values in the range (0 once done: nothing matches any more)
    unsigned int rank;     // wanted rank among the keys inside the range
    unsigned int shift;    // log2(bin width) of the level being histogrammed
    unsigned int le;       // keys below the range so far; once done: keys <= the selected key
    unsigned int done;
    unsigned int count;    // keys inside the current range
};

__device__ __forceinline__ unsigned int abs_key(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned int uniform(unsigned int v) {
    return static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
}

__device__ __forceinline__ unsigned int level_shift(unsigned int width) {
    const unsigned int bits = width <= 1u ? 0u : 32u - __builtin_clz(width - 1u);
    return bits > kSelBits ? bits - kSelBits : 0u;
}

// development aid (-DOSQ_FINAL_TIMING): thread 0 of every workgroup stamps s_memtime into a debug buffer
#ifdef OSQ_FINAL_TIMING
__device__ long long* g_osq_dbg = nullptr;
#define OSQ_WSTAMP(k) do { if (threadIdx.x == 0 && g_osq_dbg) g_osq_dbg[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#define OSQ_DBGVAL(k, v) do { if (g_osq_dbg) g_osq_dbg[blockIdx.x * 8 + (k)] = (v); } while (0)
#else
#define OSQ_WSTAMP(k) do { } while (0)
#define OSQ_DBGVAL(k, v) do { } while (0)
#endif
// development aid: -DOSQ_FINAL_TIMING makes thread 0 write s_memtime stamps after cur_minmax[0..1]
#ifdef OSQ_FINAL_TIMING
#define OSQ_STAMP(k) do { if (threadIdx.x == 0 && fin.cur) reinterpret_cast<long long*>(fin.cur + 2)[k] = __builtin_readcyclecounter(); } while (0)
#else
#define OSQ_STAMP(k) do { } while (0)
#endif

// Batched form (token-wise-clipping grid search): workgroup p = quantizer*n_batches + batch works on
// token arrays at p*problem_stride, lengths of its batch, prune flag of its quantizer, and writes
// its (min, max) to cur[(batch*n_quantizers + quantizer)*2] -- the [batches, quantizers, 2] table
// that calibration.replay() consumes.  n_batches == 0: the single-problem call.
struct FinalBatch {
    int64_t problem_stride;
    int n_batches, n_quantizers;
    const int* prune_flags;
    int lengths_per_problem;      // lengths is [n_quantizers, n_batches, B] (every site its own mask) instead of [n_batches, B]
};

template <int SLOTS>
__global__ __launch_bounds__(kFinalThreads) void token_finalize_kernel(const float* __restrict__ tok_min,
                                                                       const float* __restrict__ tok_max, int64_t B,
                                                                       int64_t T, const int64_t* __restrict__ lengths,
                                                                       int prune, float q, Finish fin, FinalBatch fb) {
    if (fb.n_batches > 0) {
        const int p = blockIdx.x, qi = p / fb.n_batches, bi = p - qi * fb.n_batches;
        tok_min += static_cast<int64_t>(p) * fb.problem_stride;
        tok_max += static_cast<int64_t>(p) * fb.problem_stride;
        if (lengths) lengths += static_cast<int64_t>(fb.lengths_per_problem ? blockIdx.x : bi) * B;
        prune = fb.prune_flags ? fb.prune_flags[qi] : prune;
        fin.cur += 2 * (static_cast<int64_t>(bi) * fb.n_quantizers + qi);
    }
    constexpr bool CACHED = SLOTS > 0;
    constexpr int R = CACHED ? SLOTS : 1;
    constexpr int kWaves = kFinalThreads / OSQ_WAVE;
    __shared__ unsigned int hist[2][kSelBins];
    __shared__ unsigned int list[2][kListCap];
    __shared__ SelState sel[2];
    __shared__ unsigned int s_n, s_bad, s_next[2], s_kmin[2], s_kmax[2], s_omin, s_omax, s_fill[2], s_found[2][2];
    __shared__ unsigned int s_wtot[2][kWaves];

    const int tid = threadIdx.x, lane = tid & (OSQ_WAVE - 1), wv = tid / OSQ_WAVE;
    const int64_t slots = B * T;

    // ---- register cache: slot = tid + i*1024.  Every load is unconditional (index clamped) and
    // independent, so all of them are in flight together; validity comes from the lengths.
    float r_mn[R], r_mx[R];
    unsigned int vmask = 0u;
    OSQ_STAMP(0);
    if (CACHED) {
        const unsigned int Tu = static_cast<unsigned int>(T), last = static_cast<unsigned int>(slots) - 1u;
        {
            // (b, t) of slot tid + i*1024 by stepping, one division per thread instead of one per slot
            const unsigned int step_b = kFinalThreads / Tu, step_t = kFinalThreads - step_b * Tu;
            const unsigned int Bm1 = static_cast<unsigned int>(B) - 1u;
            unsigned int bb = static_cast<unsigned int>(tid) / Tu, tt = static_cast<unsigned int>(tid) - bb * Tu;
            int len_i[R];
            unsigned int t_i[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                int64_t l = T;
                if (lengths) l = lengths[bb < Bm1 ? bb : Bm1];
                len_i[i] = l > T ? static_cast<int>(T) : (l < 0 ? 0 : static_cast<int>(l));
                t_i[i] = tt;
                bb += step_b;
                tt += step_t;
                if (tt >= Tu) { tt -= Tu; ++bb; }
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const unsigned int s = static_cast<unsigned int>(tid) + static_cast<unsigned int>(i) * kFinalThreads;
                if (s <= last && static_cast<int>(t_i[i]) < len_i[i]) vmask |= (1u << i);
            }
        }
        asm volatile("" : "+v"(vmask));   // lengths are dead before the value loads start
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const unsigned int s = static_cast<unsigned int>(tid) + static_cast<unsigned int>(i) * kFinalThreads;
            const unsigned int sc = s < last ? s : last;
            r_mn[i] = tok_min[sc];
            r_mx[i] = tok_max[sc];
        }
    }
// run the statements for every valid slot with vmn / vmx bound to its token minimum / maximum
#define OSQ_FOR_EACH_VALID(...)                                                                    \
    if (CACHED) {                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < R; ++i_) {                                         \
            if ((vmask >> i_) & 1u) {                                                              \
                float vmn = r_mn[i_], vmx = r_mx[i_];                                              \
                /* opaque copies: values derived from them are recomputed per pass, not kept live */ \
                asm volatile("" : "+v"(vmn), "+v"(vmx));                                           \
                __VA_ARGS__                                                                        \
            }                                                                                      \
        }                                                                                          \
    } else {                                                                                       \
        for (int64_t b_ = wv; b_ < B; b_ += kWaves) {                                              \
            int64_t len_ = T;                                                                      \
            if (lengths) { const int64_t l_ = lengths[b_]; len_ = l_ < len_ ? l_ : len_; }         \
            const float* pmn_ = tok_min + b_ * T;                                                  \
            const float* pmx_ = tok_max + b_ * T;                                                  \
            for (int64_t t_ = lane; t_ < len_; t_ += OSQ_WAVE) {                                   \
                const float vmn = pmn_[t_], vmx = pmx_[t_];                                        \
                __VA_ARGS__                                                                        \
            }                                                                                      \
        }                                                                                          \
    }

    OSQ_STAMP(1);
    if (tid == 0) {
        s_n = 0u; s_bad = 0u; s_omin = 0xffffffffu; s_omax = 0u;
        s_kmin[0] = s_kmin[1] = 0xffffffffu;
        s_kmax[0] = s_kmax[1] = 0u;
        s_next[0] = s_next[1] = 0xffffffffu;
        s_fill[0] = s_fill[1] = 0u;
    }
    __syncthreads();

    // ---- pass 0: N, NaN flag, plain extrema, key ranges
    {
        unsigned int n = 0u, kmin0 = 0xffffffffu, kmax0 = 0u, kmin1 = 0xffffffffu, kmax1 = 0u;
        MinMax plain;
        plain.init();
        OSQ_FOR_EACH_VALID({
            ++n;
            plain.mn = fminf(plain.mn, vmn);
            plain.mx = fmaxf(plain.mx, vmx);
            plain.bad |= (vmn != vmn) || (vmx != vmx);
            const unsigned int k0 = abs_key(vmx), k1 = abs_key(vmn);
            kmin0 = min(kmin0, k0); kmax0 = max(kmax0, k0);
            kmin1 = min(kmin1, k1); kmax1 = max(kmax1, k1);
        })
        plain.wave_reduce();
        kmin0 = wave_min_u32(kmin0); kmax0 = wave_max_u32(kmax0);
        kmin1 = wave_min_u32(kmin1); kmax1 = wave_max_u32(kmax1);
        n = wave_inclusive_scan_u32(n);      // lane 63 holds the wave total
        if (lane == OSQ_WAVE - 1) atomicAdd(&s_n, n);
        if (lane == 0) {
            if (plain.bad) atomicOr(&s_bad, 1u);
            atomicMin(&s_omin, ordered_bits(plain.mn));
            atomicMax(&s_omax, ordered_bits(plain.mx));
            atomicMin(&s_kmin[0], kmin0); atomicMax(&s_kmax[0], kmax0);
            atomicMin(&s_kmin[1], kmin1); atomicMax(&s_kmax[1], kmax1);
        }
    }
    __syncthreads();
    OSQ_STAMP(2);
    const unsigned int N = s_n;
    const bool bad = s_bad != 0u;
    float cur_min = from_ordered_bits(s_omin), cur_max = from_ordered_bits(s_omax);

    if (N > 0u && prune && !bad) {
        const float rank = q * static_cast<float>(N - 1u);
        const float rlo = floorf(rank);
        const unsigned int k_lo = static_cast<unsigned int>(rlo);
        const unsigned int k_hi = static_cast<unsigned int>(ceilf(rank));
        const float w = rank - rlo;

        if (tid < 2) {
            sel[tid].lo = s_kmin[tid];
            sel[tid].width = s_kmax[tid] - s_kmin[tid] + 1u;
            sel[tid].rank = k_lo;
            sel[tid].shift = level_shift(sel[tid].width);
            sel[tid].le = 0u;
            sel[tid].done = 0u;
            sel[tid].count = N;
        }
        __syncthreads();
        // ---- histogram levels, both arrays at once: [0] = |token_max|, [1] = |token_min|.
        // Level 0 always runs; levels 1-2 only when a bin is too crowded for the list.
        bool listed = false;
        for (int level = 0; level < 3; ++level) {
            if (sel[0].done && sel[1].done) break;          // uniform: sel is only written between barriers
            if (level > 0 && (sel[0].done || sel[0].count <= kListCap) && (sel[1].done || sel[1].count <= kListCap)) {
                listed = true;
                break;
            }
            for (int k = tid; k < 2 * kSelBins; k += kFinalThreads) (&hist[0][0])[k] = 0u;
            __syncthreads();
            // block-uniform values go to SGPRs: no LDS-read dependency inside the per-slot code
            const unsigned int lo0 = uniform(sel[0].lo), w0 = uniform(sel[0].width), sh0 = uniform(sel[0].shift);
            const unsigned int lo1 = uniform(sel[1].lo), w1 = uniform(sel[1].width), sh1 = uniform(sel[1].shift);
            OSQ_FOR_EACH_VALID({
                const unsigned int d0 = abs_key(vmx) - lo0, d1 = abs_key(vmn) - lo1;
                if (d0 < w0) atomicAdd(&hist[0][d0 >> sh0], 1u);
                if (d1 < w1) atomicAdd(&hist[1][d1 >> sh1], 1u);
            })
            __syncthreads();
            // block-wide exclusive scan over the 2048 bins of each array (2 bins per thread),
            // then the one thread whose bins straddle the wanted rank narrows the range
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const unsigned int h0 = hist[a][2 * tid], h1 = hist[a][2 * tid + 1];
                const unsigned int incl_w = wave_inclusive_scan_u32(h0 + h1);
                if (lane == OSQ_WAVE - 1) s_wtot[a][wv] = incl_w;
                __syncthreads();
                unsigned int base = 0u;
#pragma unroll
                for (int k = 0; k < kWaves; ++k) base += (k < wv) ? s_wtot[a][k] : 0u;
                const unsigned int incl = base + incl_w, excl = incl - (h0 + h1);
                const unsigned int want = sel[a].rank;
                const bool active = !sel[a].done;
                __syncthreads();                              // everyone has read sel[a] / s_wtot[a]
                if (active && want >= excl && want < incl) {  // exactly one thread
                    const bool second = want >= excl + h0;
                    const unsigned int below = second ? excl + h0 : excl;
                    const unsigned int bin = 2u * tid + (second ? 1u : 0u), cnt = second ? h1 : h0;
                    const unsigned int sh = sel[a].shift, off = bin << sh;
                    sel[a].lo += off;
                    sel[a].count = cnt;
                    if (sh == 0u) {                           // bins are single keys: found
                        sel[a].le += below + cnt;
                        sel[a].width = 0u;
                        sel[a].done = 1u;
                    } else {
                        const unsigned int rest = sel[a].width - off, cap = 1u << sh;
                        sel[a].le += below;
                        sel[a].rank = want - below;
                        sel[a].width = rest < cap ? rest : cap;
                        sel[a].shift = level_shift(sel[a].width);
                    }
                }
            }
            __syncthreads();
        }
        OSQ_STAMP(3);
        unsigned int v0, v1, hi0, hi1;      // keys at floor(rank) / ceil(rank) of |token_max|, |token_min|
        if (listed) {
            // ---- pass 2: compact the keys of the chosen bins; remember the smallest key above each bin
            const unsigned int lo0 = uniform(sel[0].lo), w0 = uniform(sel[0].width), d0ne = uniform(sel[0].done);
            const unsigned int lo1 = uniform(sel[1].lo), w1 = uniform(sel[1].width), d1ne = uniform(sel[1].done);
            unsigned int n0 = 0xffffffffu, n1 = 0xffffffffu;
            OSQ_FOR_EACH_VALID({
                const unsigned int k0 = abs_key(vmx), k1 = abs_key(vmn);
                const unsigned int e0 = k0 - lo0, e1 = k1 - lo1;
                if (d0ne) { if (k0 > lo0) n0 = min(n0, k0); }
                else if (e0 < w0) list[0][atomicAdd(&s_fill[0], 1u)] = k0;
                else if (k0 >= lo0) n0 = min(n0, k0);
                if (d1ne) { if (k1 > lo1) n1 = min(n1, k1); }
                else if (e1 < w1) list[1][atomicAdd(&s_fill[1], 1u)] = k1;
                else if (k1 >= lo1) n1 = min(n1, k1);
            })
            n0 = wave_min_u32(n0);
            n1 = wave_min_u32(n1);
            if (lane == 0) { atomicMin(&s_next[0], n0); atomicMin(&s_next[1], n1); }
            if (tid < 4) s_found[tid >> 1][tid & 1] = 0xffffffffu;
            __syncthreads();
            // rank every list entry by counting (ties broken by position): entries at rank and rank+1
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const unsigned int cnt = sel[a].done ? 0u : s_fill[a];
                if (static_cast<unsigned int>(tid) < cnt) {
                    const unsigned int mine = list[a][tid];
                    unsigned int r = 0u;
                    for (unsigned int j = 0; j < cnt; ++j) {
                        const unsigned int o = list[a][j];
                        r += (o < mine || (o == mine && j < static_cast<unsigned int>(tid))) ? 1u : 0u;
                    }
                    if (r == sel[a].rank) s_found[a][0] = mine;
                    if (r == sel[a].rank + 1u) s_found[a][1] = mine;
                }
            }
            __syncthreads();
            // key at ceil(rank): next list entry, else the smallest key above the bin
            v0 = sel[0].done ? sel[0].lo : s_found[0][0];
            v1 = sel[1].done ? sel[1].lo : s_found[1][0];
            hi0 = sel[0].done ? (sel[0].le > k_hi ? v0 : s_next[0]) : (s_found[0][1] != 0xffffffffu ? s_found[0][1] : s_next[0]);
            hi1 = sel[1].done ? (sel[1].le > k_hi ? v1 : s_next[1]) : (s_found[1][1] != 0xffffffffu ? s_found[1][1] : s_next[1]);
        } else {
            // all levels ran: sel[a].lo is the key at rank k_lo, sel[a].le the number of keys <= it
            v0 = uniform(sel[0].lo);
            v1 = uniform(sel[1].lo);
            if (k_hi != k_lo) {   // smallest key above, used when duplicates do not cover rank k_lo + 1
                unsigned int n0 = 0xffffffffu, n1 = 0xffffffffu;
                OSQ_FOR_EACH_VALID({
                    const unsigned int k0 = abs_key(vmx), k1 = abs_key(vmn);
                    if (k0 > v0) n0 = min(n0, k0);
                    if (k1 > v1) n1 = min(n1, k1);
                })
                n0 = wave_min_u32(n0);
                n1 = wave_min_u32(n1);
                if (lane == 0) { atomicMin(&s_next[0], n0); atomicMin(&s_next[1], n1); }
                __syncthreads();
            }
            hi0 = sel[0].le > k_hi ? v0 : s_next[0];
            hi1 = sel[1].le > k_hi ? v1 : s_next[1];
        }
        OSQ_STAMP(4);
        if (k_hi == k_lo) { hi0 = v0; hi1 = v1; }
        float thr[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float lo_v = __uint_as_float(a == 0 ? v0 : v1), hi_v = __uint_as_float(a == 0 ? hi0 : hi1);
            const float diff = hi_v - lo_v;
            thr[a] = (w < 0.5f) ? __builtin_fmaf(w, diff, lo_v) : __builtin_fmaf(w - 1.0f, diff, hi_v);
        }
        const float upper = __uint_as_float(uniform(__float_as_uint(thr[0])));
        const float lower = -__uint_as_float(uniform(__float_as_uint(thr[1])));
        // ---- pass 3: up = max(token_max[token_max <= upper]) ; lo = min(token_min[token_min >= lower])
        __syncthreads();
        if (tid == 0) { s_omin = 0xffffffffu; s_omax = 0u; }
        __syncthreads();
        MinMax pr;
        pr.init();
        OSQ_FOR_EACH_VALID({
            if (vmn >= lower) pr.mn = fminf(pr.mn, vmn);
            if (vmx <= upper) pr.mx = fmaxf(pr.mx, vmx);
        })
        pr.mn = wave_min(pr.mn);
        pr.mx = wave_max(pr.mx);
        if (lane == 0) {
            atomicMin(&s_omin, ordered_bits(pr.mn));
            atomicMax(&s_omax, ordered_bits(pr.mx));
        }
        __syncthreads();
        const float lo_sel = from_ordered_bits(s_omin), up_sel = from_ordered_bits(s_omax);
        // aminmax(clip(value, lo, up)) (observer.py:68,227): (lo, up), or (up, up) if lo > up
        cur_min = clipped_min(lo_sel, up_sel);
        cur_max = up_sel;
        OSQ_STAMP(5);
    }
    if (tid == 0 && N > 0u) {
        if (bad) { cur_min = __builtin_nanf(""); cur_max = __builtin_nanf(""); }
        finish_entry(fin, 0, cur_min, cur_max);
    }
    OSQ_STAMP(6);
#undef OSQ_FOR_EACH_VALID
}

}  // namespace osq

#include "token_select.h"   // two-workgroup fast path (one per side), used whenever its layout rules hold
#include "fused_step.h"     // one persistent launch for observe + fake-quant of a dense [B, T, H] activation

namespace osq {

// ---------------------------------------------------------------- wide finaliser (many token slots)

// One CU pulls only ~10 B/clk from the fabric, so a single workgroup needs ~10 us just to READ the
// extrema of 32768 tokens.  Above g_wide_min_slots the same selection runs as three multi-workgroup
// launches (a kernel boundary costs ~1.5 us, a hand-rolled grid barrier more):
//   A  coarse histogram: bin = key >> 20 (exponent + 3 mantissa bits, 2048 bins) accumulated into
//      a global table with wave-aggregated atomics (keys of one wave share a handful of bins: one
//      atomic per distinct bin per wave), plus N / NaN flag / plain extrema;
//   B  every workgroup scans the 2 x 2048 table (8 KB, L2) to find the coarse bin holding
//      rank k_lo, then appends its slots' keys of that bin to a global list (wave-aggregated
//      reservation) and tracks the smallest key above the bin;
//   C  every workgroup selects ranks r and r+1 inside the (small) list redundantly -- two
//      LDS-histogram levels over the 20 remaining key bits -- forms the thresholds, reduces
//      max(token_max <= upper) / min(token_min >= lower) over its slots; the last workgroup
//      (sharded tickets) finishes (running statistic, qparams) and re-zeroes the global scratch.
// Measured on MI355X at 32768 slots: single workgroup 28.5 us, the three launches 30 us (6.4 + 9.5 + 14;
// each is dominated by launch + dependent-load latency, not work) -- so the wide path only takes over
// above the register-cache limit of the single-workgroup kernel, where that kernel would re-read the
// arrays from L2 in every pass (~25 us per pass at 65536 slots).  osq_set_wide_min_slots() overrides.
OSQ_SWITCH(int64_t, g_wide_min_slots, 32769);
// Grid cap of observe_flat (osq_set_tuning("obs_blocks", n), <= kMaxBlocks).  The four loads of a thread are
// gridDim.x * 4 KB apart: power-of-two grids (1024, 2048) put them on the same memory channels and measured
// 21.2 us on the [256,128,768] tensor against 18.8 us at 768 (tools/obs_sweep.py).
OSQ_AB_KNOB(int, g_obs_blocks, 768);
OSQ_AB_KNOB(int, g_tok_nt, 1);              // osq_set_tuning("tok_nt", 0): per-token kernel loads without the streaming hint
OSQ_SWITCH(int, g_select_shortcut, 1);     // osq_set_tuning("select_shortcut", 0): always run the register threshold pass (tests)
OSQ_SWITCH(int, g_final_fast, 1);          // osq_set_tuning("final_fast", 0) forces the single-workgroup kernel (tests)
// osq_set_tuning("fused_step", 0) or OSQ_FUSED_STEP=0 in the environment: observe + fake-quant as three launches
static std::atomic<int> g_fused_step{[] { const char* e = getenv("OSQ_FUSED_STEP"); return (e && e[0] == '0') ? 0 : 1; }()};
OSQ_AB_KNOB(int, g_fused_gate, 2);          // osq_set_tuning("fused_gate", 0|1|2): which padded loads wait for the selectors (fused_step.h, phase A2)
OSQ_AB_KNOB(int, g_select_hint, 1);          // osq_set_tuning("select_hint", 0|1): the fused step's selectors histogram a window around the running statistic while the extrema arrive (token_select.h, HINT)
OSQ_SWITCH(unsigned int, g_fused_spin_limit, 0u);     // osq_set_tuning("fused_spin_limit", n): 0 = kFusedSpinLimit, n > 0 = n - 1 polls (1: every wait gives up at once -- tests force the time-out path with it)
OSQ_AB_KNOB(int, g_fused_grid, 0);          // osq_set_tuning("fused_grid", n): workgroups of the fused launch (0 = one per CU)
constexpr int kWideThreads = 256;
constexpr int kWideSlotsPerBlock = 512;
constexpr int kCoarseShift = 20;

struct WideState {            // lives in the caller's workspace; ALL-ZERO is the idle state
    unsigned int hist[2][kSelBins];
    unsigned int n, bad;
    unsigned int omin_inv, omax;              // ~ordered(min) / ordered(max): both grow with atomicMax from 0
    unsigned int fill[2], next_inv[2];        // ~(smallest key above the chosen bin)
    unsigned int thr_omin_inv, thr_omax;
    unsigned int pad[6];
};
__device__ __forceinline__ void wide_state_clear(WideState* ws) {   // scalars only; hist is cleared by all threads
    ws->n = 0u; ws->bad = 0u; ws->omin_inv = 0u; ws->omax = 0u;
    ws->fill[0] = ws->fill[1] = 0u;
    ws->next_inv[0] = ws->next_inv[1] = 0u;
    ws->thr_omin_inv = 0u; ws->thr_omax = 0u;
}
__device__ __forceinline__ unsigned int peek(const unsigned int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct WideArgs {
    const float* tok_min;
    const float* tok_max;
    int64_t B, T;
    const int64_t* lengths;
    WideState* ws;
    unsigned int* list;        // 2 x slots keys
    unsigned int* tickets;
    int prune;
    float q;
};

// Load a slot's extrema unconditionally (index clamped) so that the loads do not wait for the length
// lookup; validity is decided afterwards.
__device__ __forceinline__ bool wide_slot(const WideArgs& a, int64_t s, float* mn, float* mx) {
    const int64_t slots = a.B * a.T;
    const int64_t sc = s < slots ? s : slots - 1;
    const unsigned int Tu = static_cast<unsigned int>(a.T), su = static_cast<unsigned int>(sc);
    const unsigned int b = su / Tu, t = su - b * Tu;
    int64_t len = a.T;
    if (a.lengths) len = a.lengths[b];
    *mn = a.tok_min[sc];
    *mx = a.tok_max[sc];
    return s < slots && static_cast<int64_t>(t) < len;
}

// Hot coarse bins are neighbours (same exponent, adjacent mantissa bits); in bin order they would share
// one or two 64-byte lines and every global atomic of every workgroup would queue on them.  Entry b of
// the global table therefore lives at perm(b): neighbours end up 128 entries (8 lines) apart.
__device__ __forceinline__ unsigned int coarse_slot(unsigned int bin) { return ((bin & 15u) << 7) | (bin >> 4); }

__global__ __launch_bounds__(kWideThreads) void wide_hist_kernel(WideArgs a, Finish fin) {
    __shared__ unsigned int lh[2][kSelBins];
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    if (a.prune) {
        for (int k = threadIdx.x; k < 2 * kSelBins; k += kWideThreads) (&lh[0][0])[k] = 0u;
        __syncthreads();
    }
    MinMax plain;
    plain.init();
    unsigned int n = 0u;
#pragma unroll
    for (int r = 0; r < kWideSlotsPerBlock / kWideThreads; ++r) {
        const int64_t s = static_cast<int64_t>(blockIdx.x) * kWideSlotsPerBlock + r * kWideThreads + threadIdx.x;
        float mn = 0.f, mx = 0.f;
        const bool ok = wide_slot(a, s, &mn, &mx);
        if (ok) {
            ++n;
            plain.mn = fminf(plain.mn, mn);
            plain.mx = fmaxf(plain.mx, mx);
            plain.bad |= (mn != mn) || (mx != mx);
        }
        if (a.prune && ok) {      // workgroup-local histogram first (LDS resolves same-bin lanes in hardware)
            atomicAdd(&lh[0][abs_key(mx) >> kCoarseShift], 1u);
            atomicAdd(&lh[1][abs_key(mn) >> kCoarseShift], 1u);
        }
    }
    if (a.prune) {          // flush the (few) non-empty bins: one global atomic each
        __syncthreads();
        for (int k = threadIdx.x; k < 2 * kSelBins; k += kWideThreads) {
            const unsigned int c = (&lh[0][0])[k];
            if (c) atomicAdd(&a.ws->hist[k >> 11][coarse_slot(k & (kSelBins - 1))], c);
        }
    }
    // block partial -> ONE set of global atomics from thread 0 (which then takes the ticket in order)
    __shared__ unsigned int s_n[kWideThreads / OSQ_WAVE];
    n = wave_inclusive_scan_u32(n);
    if (lane == OSQ_WAVE - 1) s_n[threadIdx.x / OSQ_WAVE] = n;
    plain = block_reduce(plain);
    if (threadIdx.x == 0) {
        unsigned int tot = 0u;
        for (int k = 0; k < kWideThreads / OSQ_WAVE; ++k) tot += s_n[k];
        if (tot) atomicAdd(&a.ws->n, tot);
        if (plain.bad) atomicOr(&a.ws->bad, 1u);
        atomicMax(&a.ws->omin_inv, ~ordered_bits(plain.mn));
        atomicMax(&a.ws->omax, ordered_bits(plain.mx));
    }
    if (!a.prune) {     // plain masked min/max: the last workgroup finishes right here
        if (grid_last_block(a.tickets, gridDim.x)) {
            if (threadIdx.x == 0) {
                const unsigned int N = peek(&a.ws->n);
                float cmin = from_ordered_bits(~peek(&a.ws->omin_inv));
                float cmax = from_ordered_bits(peek(&a.ws->omax));
                if (peek(&a.ws->bad)) { cmin = __builtin_nanf(""); cmax = cmin; }
                if (N > 0u) finish_entry(fin, 0, cmin, cmax);
                wide_state_clear(a.ws);
                grid_reset(a.tickets, gridDim.x);
            }
        }
    }
}

struct CoarsePick { unsigned int bin, below, count; };

// block-wide: which coarse bin holds rank `want` (2048 bins, 8 per thread); result in every thread
// hist: 2048 counts in bin order, in LDS or global memory
__device__ __forceinline__ CoarsePick pick_coarse_bin(const unsigned int* hist, unsigned int want) {
    __shared__ unsigned int s_tot[kWideThreads / OSQ_WAVE];
    __shared__ CoarsePick s_pick;
    constexpr int per = kSelBins / kWideThreads;   // 8
    const int lane = threadIdx.x & (OSQ_WAVE - 1), wv = threadIdx.x / OSQ_WAVE;
    unsigned int h[per], mine = 0u;
#pragma unroll
    for (int k = 0; k < per; ++k) { h[k] = hist[threadIdx.x * per + k]; mine += h[k]; }
    const unsigned int incl_w = wave_inclusive_scan_u32(mine);
    __syncthreads();
    if (lane == OSQ_WAVE - 1) s_tot[wv] = incl_w;
    __syncthreads();
    unsigned int base = 0u;
#pragma unroll
    for (int k = 0; k < kWideThreads / OSQ_WAVE; ++k) base += (k < wv) ? s_tot[k] : 0u;
    const unsigned int incl = base + incl_w, excl = incl - mine;
    if (want >= excl && want < incl) {
        unsigned int below = excl;
#pragma unroll
        for (int k = 0; k < per; ++k) {
            if (below + h[k] > want) { s_pick.bin = threadIdx.x * per + k; s_pick.below = below; s_pick.count = h[k]; break; }
            below += h[k];
        }
    }
    __syncthreads();
    return s_pick;
}

// Bring both permuted global coarse tables into LDS in bin order: coalesced 32-byte reads per thread in
// MEMORY order, scattered to their bin position (inverse of coarse_slot).
__device__ __forceinline__ void stage_coarse_tables(const WideState* ws, unsigned int (*lds)[kSelBins]) {
    constexpr int per = kSelBins / kWideThreads;   // 8 consecutive memory entries per thread
#pragma unroll
    for (int arr = 0; arr < 2; ++arr) {
        const uint4* src = reinterpret_cast<const uint4*>(&ws->hist[arr][threadIdx.x * per]);
        const uint4 a = src[0], b = src[1];
        const unsigned int v[per] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < per; ++k) {
            const unsigned int i = threadIdx.x * per + k;          // memory index = lo*128 + hi
            lds[arr][((i & 127u) << 4) | (i >> 7)] = v[k];          // bin = hi*16 + lo
        }
    }
    __syncthreads();
}

struct WideRanks {
    unsigned int k_lo, k_hi;
    float w;
};
__device__ __forceinline__ WideRanks wide_ranks(unsigned int N, float q) {
    const float rank = q * static_cast<float>(N - 1u);
    const float rlo = floorf(rank);
    return {static_cast<unsigned int>(rlo), static_cast<unsigned int>(ceilf(rank)), rank - rlo};
}

__global__ __launch_bounds__(kWideThreads) void wide_collect_kernel(WideArgs a) {
    const unsigned int N = a.ws->n;
    if (N == 0u || a.ws->bad) return;
    const WideRanks rk = wide_ranks(N, a.q);
    constexpr int kRows = kWideSlotsPerBlock / kWideThreads;
    // this workgroup's slots first: their loads fly while the coarse table is scanned
    float v_mn[kRows], v_mx[kRows];
    bool v_ok[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int64_t s = static_cast<int64_t>(blockIdx.x) * kWideSlotsPerBlock + r * kWideThreads + threadIdx.x;
        v_ok[r] = wide_slot(a, s, &v_mn[r], &v_mx[r]);
    }
    __shared__ unsigned int coarse[2][kSelBins];
    stage_coarse_tables(a.ws, coarse);
    const CoarsePick p0 = pick_coarse_bin(coarse[0], rk.k_lo);
    const CoarsePick p1 = pick_coarse_bin(coarse[1], rk.k_lo);
    const int64_t slots = a.B * a.T;
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    __shared__ unsigned int s_cnt[2], s_base[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    unsigned int n0 = 0xffffffffu, n1 = 0xffffffffu;
    unsigned int keys[kRows][2], pos[kRows][2];
    bool hits[kRows][2];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const float mn = v_mn[r], mx = v_mx[r];
        const bool ok = v_ok[r];
#pragma unroll
        for (int arr = 0; arr < 2; ++arr) {
            const unsigned int key = abs_key(arr ? mn : mx), c = key >> kCoarseShift, target = arr ? p1.bin : p0.bin;
            const bool hit = ok && c == target;
            keys[r][arr] = key;
            hits[r][arr] = hit;
            pos[r][arr] = 0u;
            const unsigned long long m = __ballot(hit);
            if (m) {                                 // position inside the workgroup: one LDS atomic per wave
                unsigned int wbase = 0u;
                const int leader = __ffsll(static_cast<long long>(m)) - 1;
                if (lane == leader) wbase = atomicAdd(&s_cnt[arr], static_cast<unsigned int>(__popcll(m)));
                wbase = static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(wbase), leader));
                pos[r][arr] = wbase + static_cast<unsigned int>(__popcll(m & ((1ull << lane) - 1ull)));
            }
            if (ok && c > target) { if (arr) n1 = min(n1, key); else n0 = min(n0, key); }
        }
    }
    __syncthreads();
    if (threadIdx.x < 2) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&a.ws->fill[threadIdx.x], s_cnt[threadIdx.x]) : 0u;
    __syncthreads();                                 // ONE global reservation per workgroup and array
#pragma unroll
    for (int r = 0; r < kRows; ++r)
#pragma unroll
        for (int arr = 0; arr < 2; ++arr)
            if (hits[r][arr]) a.list[static_cast<int64_t>(arr) * slots + s_base[arr] + pos[r][arr]] = keys[r][arr];
    n0 = wave_min_u32(n0);
    n1 = wave_min_u32(n1);
    if (lane == 0) {
        if (n0 != 0xffffffffu) atomicMax(&a.ws->next_inv[0], ~n0);
        if (n1 != 0xffffffffu) atomicMax(&a.ws->next_inv[1], ~n1);
    }
}

// Keys at ranks r and r+1 (r+1 only if it exists) of both lists at once.  Every key of list `arr` shares
// its top 11 bits; the remaining 20 are resolved by two LDS-histogram passes over the lists:
// bits [19:9] (2048 bins), then bits [8:0] inside the chosen bin (512 bins) -- during the second pass
// the smallest key of any higher first-level bin is tracked too, which is rank r+1 when the
// first-level bin ends at rank r.  found[arr][0/1] = low 20 bits, 0xffffffff if rank r+1 is not in the list.
__device__ __forceinline__ void list_select2(const unsigned int* list0, unsigned int cnt0, unsigned int r0,
                                             const unsigned int* list1, unsigned int cnt1, unsigned int r1,
                                             unsigned int (*hist)[kSelBins], unsigned int found[2][2]) {
    __shared__ unsigned int s_above[2];
    const unsigned int* lists[2] = {list0, list1};
    const unsigned int cnts[2] = {cnt0, cnt1}, ranks[2] = {r0, r1};
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    // ---- pass 1: bits [19:9]
    for (int k = threadIdx.x; k < 2 * kSelBins; k += kWideThreads) (&hist[0][0])[k] = 0u;
    if (threadIdx.x < 2) s_above[threadIdx.x] = 0xffffffffu;
    __syncthreads();
#pragma unroll
    for (int arr = 0; arr < 2; ++arr)
        for (unsigned int j = threadIdx.x; j < cnts[arr]; j += kWideThreads)
            atomicAdd(&hist[arr][(lists[arr][j] >> 9) & 0x7ffu], 1u);
    __syncthreads();
    CoarsePick top[2];
#pragma unroll
    for (int arr = 0; arr < 2; ++arr) top[arr] = pick_coarse_bin(hist[arr], ranks[arr]);
    // ---- pass 2: bits [8:0] of the keys inside the chosen first-level bin; min key of higher bins
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * kSelBins; k += kWideThreads) (&hist[0][0])[k] = 0u;
    __syncthreads();
#pragma unroll
    for (int arr = 0; arr < 2; ++arr) {
        unsigned int above = 0xffffffffu;
        for (unsigned int j = threadIdx.x; j < cnts[arr]; j += kWideThreads) {
            const unsigned int key = lists[arr][j] & 0xfffffu, b = key >> 9;
            if (b == top[arr].bin) atomicAdd(&hist[arr][key & 0x1ffu], 1u);
            else if (b > top[arr].bin) above = min(above, key);
        }
        above = wave_min_u32(above);
        if (lane == 0 && above != 0xffffffffu) atomicMin(&s_above[arr], above);
    }
    __syncthreads();
#pragma unroll
    for (int arr = 0; arr < 2; ++arr) {
        const unsigned int want = ranks[arr] - top[arr].below;
        const CoarsePick lo = pick_coarse_bin(hist[arr], want);
        found[arr][0] = (top[arr].bin << 9) | lo.bin;
        found[arr][1] = 0xffffffffu;
        if (ranks[arr] + 1u < cnts[arr]) {
            if (want + 1u < top[arr].count) {            // rank r+1 sits in the same first-level bin
                const CoarsePick hi = pick_coarse_bin(hist[arr], want + 1u);
                found[arr][1] = (top[arr].bin << 9) | hi.bin;
            } else {
                found[arr][1] = s_above[arr];
            }
        }
    }
}

__global__ __launch_bounds__(kWideThreads) void wide_select_kernel(WideArgs a, Finish fin) {
    __shared__ __attribute__((aligned(16))) unsigned int hist[2][kSelBins];
    __shared__ unsigned int s_omin, s_omax;
    const unsigned int N = a.ws->n;
    const bool bad = a.ws->bad != 0u;
    const int64_t slots = a.B * a.T;
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    float upper = 0.f, lower = 0.f;
    constexpr int kRows = kWideSlotsPerBlock / kWideThreads;
    float v_mn[kRows], v_mx[kRows];
    bool v_ok[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {     // issued first: in flight during the selection below
        const int64_t s = static_cast<int64_t>(blockIdx.x) * kWideSlotsPerBlock + r * kWideThreads + threadIdx.x;
        v_ok[r] = wide_slot(a, s, &v_mn[r], &v_mx[r]);
    }
    OSQ_WSTAMP(0);
    if (N > 0u && !bad) {
        const WideRanks rk = wide_ranks(N, a.q);
        float thr[2];
        stage_coarse_tables(a.ws, hist);
        const CoarsePick c0 = pick_coarse_bin(hist[0], rk.k_lo);
        const CoarsePick c1 = pick_coarse_bin(hist[1], rk.k_lo);
        __syncthreads();
        OSQ_WSTAMP(1);
        unsigned int found[2][2];
        list_select2(a.list, c0.count, rk.k_lo - c0.below, a.list + slots, c1.count, rk.k_lo - c1.below, hist, found);
        OSQ_WSTAMP(2);
        if (threadIdx.x == 0 && blockIdx.x == 0) { OSQ_DBGVAL(6, c0.count); OSQ_DBGVAL(7, c1.count); }
#pragma unroll
        for (int arr = 0; arr < 2; ++arr) {
            const unsigned int top = (arr ? c1.bin : c0.bin) << kCoarseShift;
            const unsigned int v_lo = top | found[arr][0];
            unsigned int v_hi = v_lo;
            if (rk.k_hi != rk.k_lo) v_hi = found[arr][1] != 0xffffffffu ? (top | found[arr][1]) : ~a.ws->next_inv[arr];
            const float lo_v = __uint_as_float(v_lo), hi_v = __uint_as_float(v_hi), diff = hi_v - lo_v;
            thr[arr] = (rk.w < 0.5f) ? __builtin_fmaf(rk.w, diff, lo_v) : __builtin_fmaf(rk.w - 1.0f, diff, hi_v);
        }
        upper = thr[0];
        lower = -thr[1];
        MinMax pr;
        pr.init();
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            if (v_ok[r]) {
                if (v_mn[r] >= lower) pr.mn = fminf(pr.mn, v_mn[r]);
                if (v_mx[r] <= upper) pr.mx = fmaxf(pr.mx, v_mx[r]);
            }
        }
        pr.mn = wave_min(pr.mn);
        pr.mx = wave_max(pr.mx);
        if (threadIdx.x == 0) { s_omin = 0xffffffffu; s_omax = 0u; }
        __syncthreads();
        if (lane == 0) { atomicMin(&s_omin, ordered_bits(pr.mn)); atomicMax(&s_omax, ordered_bits(pr.mx)); }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicMax(&a.ws->thr_omin_inv, ~s_omin);
            atomicMax(&a.ws->thr_omax, s_omax);
        }
    }
    OSQ_WSTAMP(3);
    __syncthreads();
    if (grid_last_block(a.tickets, gridDim.x)) {
        if (threadIdx.x == 0 && N > 0u) {
            float cmin, cmax;
            if (bad) {
                cmin = cmax = __builtin_nanf("");
            } else {
                const float lo_sel = from_ordered_bits(~peek(&a.ws->thr_omin_inv));
                const float up_sel = from_ordered_bits(peek(&a.ws->thr_omax));
                cmin = clipped_min(lo_sel, up_sel);     // aminmax(clip(value, lo, up)), observer.py:68,227
                cmax = up_sel;
            }
            finish_entry(fin, 0, cmin, cmax);
        }
        // leave the global scratch zeroed for the next call
        for (int k = threadIdx.x; k < 2 * kSelBins; k += kWideThreads) (&a.ws->hist[0][0])[k] = 0u;
        if (threadIdx.x == 0) {
            wide_state_clear(a.ws);
            grid_reset(a.tickets, gridDim.x);
        }
    }
}

static inline int grid_for(int64_t work_items, int per_block, int max_blocks) {
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return static_cast<int>(b);
}

static inline int check_finish_args(int update_rule, const float* min_val, const float* max_val, const char** why) {
    if (update_rule < OSQ_UPDATE_NONE || update_rule > OSQ_UPDATE_AVERAGE) { *why = "bad update rule"; return 0; }
    if (update_rule != OSQ_UPDATE_NONE && (!min_val || !max_val)) { *why = "update rule needs min_val/max_val"; return 0; }
    return 1;
}


// Layout rules of token_select_kernel: 16-byte groups (slot count and problem stride multiples of 4,
// aligned arrays), T >= 4 so that a group straddles at most two samples, at most 32 values per thread
// (64 would spill at 1024 threads per workgroup).
static inline bool select_fast_ok(const float* tmin, const float* tmax, int64_t B, int64_t T, int64_t stride,
                                  int64_t problems) {
    const int64_t S = B * T;
    return g_final_fast && T >= 4 && (S & 3) == 0 && S <= 4 * 8 * kSelThreads && aligned16(tmin) && aligned16(tmax) &&
           (stride & 3) == 0 && problems >= 1 && problems <= static_cast<int64_t>(kWsMeetBytes / 8) && problems <= 65535;
}

bool set_observer_tuning(const char* key, int value) {
    const std::string k(key);
    if (k == "final_fast") { g_final_fast = value != 0; return true; }
    if (k == "fused_step") { g_fused_step = value != 0; return true; }
    if (k == "fused_spin_limit") { if (value < 0) return false; g_fused_spin_limit = static_cast<unsigned int>(value); return true; }
    if (k == "select_shortcut") { g_select_shortcut = value != 0; return true; }
#ifdef OSQ_TUNABLE
    if (k == "fused_gate") { if (value < 0 || value > 2) return false; g_fused_gate = value; return true; }
    if (k == "fused_grid") { if (value != 0 && value < 3) return false; g_fused_grid = value; return true; }
    if (k == "tok_nt") { g_tok_nt = value != 0; return true; }
    if (k == "select_hint") { g_select_hint = value != 0; return true; }
    if (k == "obs_blocks") { if (value < 1 || value > kMaxBlocks) return false; g_obs_blocks = value; return true; }
#endif
    return false;
}

static inline void launch_select(hipStream_t st, const SelectArgs& a, const Finish& fin, const FinalBatch& fb,
                                 int64_t problems) {
    const int64_t groups_per_thread = ((a.B * a.T) / 4 + kSelThreads - 1) / kSelThreads;
    const dim3 grid(2, static_cast<unsigned>(problems));
    const TimingHook th = take_timing_hook(OSQ_TIME_TOKEN_SELECT);
#define OSQ_LAUNCH_SELECT(R4) \
    hipExtLaunchKernelGGL(token_select_kernel<R4>, grid, dim3(kSelThreads), 0, st, th.start, th.stop, 0, a, fin, fb)
    if (groups_per_thread <= 1) OSQ_LAUNCH_SELECT(1);
    else if (groups_per_thread <= 2) OSQ_LAUNCH_SELECT(2);
    else if (groups_per_thread <= 4) OSQ_LAUNCH_SELECT(4);
    else OSQ_LAUNCH_SELECT(8);
#undef OSQ_LAUNCH_SELECT
}

}  // namespace osq

using namespace osq;

extern "C" int osq_calculate_qparams(const float* min_val, const float* max_val, int64_t n,
                                     int quant_min, int quant_max, int symmetric,
                                     float* scale_out, void* zero_point_out, int zp_type,
                                     osq_stream stream) {
    OSQ_REQUIRE(n >= 0 && min_val && max_val && scale_out, "calculate_qparams: null pointer or n < 0");
    OSQ_REQUIRE(quant_max > quant_min, "calculate_qparams: quant_max must exceed quant_min");
    if (n == 0) return OSQ_OK;
    const int block = 256;
    hipLaunchKernelGGL(qparams_kernel, dim3(static_cast<unsigned>((n + block - 1) / block)), dim3(block), 0,
                       static_cast<hipStream_t>(stream), min_val, max_val, n, quant_min, quant_max, symmetric, scale_out,
                       zero_point_out, zp_type);
    return check_launch("calculate_qparams");
}

extern "C" int osq_observer_update(const float* cur_min, const float* cur_max, int64_t n,
                                   int update_rule, int64_t cnt, float* min_val, float* max_val,
                                   osq_stream stream) {
    OSQ_REQUIRE(n >= 0 && cur_min && cur_max && min_val && max_val, "observer_update: null pointer or n < 0");
    OSQ_REQUIRE(update_rule == OSQ_UPDATE_RUNNING || update_rule == OSQ_UPDATE_AVERAGE, "observer_update: bad rule");
    if (n == 0) return OSQ_OK;
    const int block = 256;
    hipLaunchKernelGGL(update_kernel, dim3(static_cast<unsigned>((n + block - 1) / block)), dim3(block), 0,
                       static_cast<hipStream_t>(stream), cur_min, cur_max, n, update_rule, cnt, min_val, max_val);
    return check_launch("observer_update");
}

extern "C" int osq_replay_statistics(const float* table, int n_batches, int n_quantizers, const int32_t* rules,
                                     int64_t cnt0, int fresh, const uint64_t* min_ptrs, const uint64_t* max_ptrs,
                                     const int32_t* quant_min, const int32_t* quant_max, const int32_t* symmetric,
                                     const uint64_t* scale_ptrs, const uint64_t* zp_ptrs, const int32_t* zp_types,
                                     osq_stream stream) {
    OSQ_REQUIRE(n_batches >= 0 && n_quantizers >= 0, "replay_statistics: negative size");
    if (n_quantizers == 0) return OSQ_OK;
    OSQ_REQUIRE(table && rules && min_ptrs && max_ptrs, "replay_statistics: null pointer");
    OSQ_REQUIRE(!scale_ptrs || (quant_min && quant_max && symmetric && zp_ptrs && zp_types),
                "replay_statistics: scale pointers need the quantizer descriptions");
    const ReplayArgs a{table, n_batches, n_quantizers, rules, cnt0, fresh, min_ptrs, max_ptrs, quant_min, quant_max,
                       symmetric, scale_ptrs, zp_ptrs, zp_types};
    hipLaunchKernelGGL(replay_kernel, dim3(static_cast<unsigned>((n_quantizers + 127) / 128)), dim3(128), 0,
                       static_cast<hipStream_t>(stream), a);
    return check_launch("replay_statistics");
}

extern "C" int osq_observe_flat(const float* x, int64_t n,
                                int update_rule, int64_t cnt, float* min_val, float* max_val,
                                float* cur_minmax,
                                int quant_min, int quant_max, int symmetric,
                                float* scale_out, void* zero_point_out, int zp_type,
                                void* workspace, osq_stream stream) {
    const char* why = "";
    OSQ_REQUIRE(n > 0 && x && workspace, "observe_flat: empty tensor or null pointer");
    OSQ_REQUIRE(check_finish_args(update_rule, min_val, max_val, &why), why);
    const Finish fin{update_rule, cnt, min_val, max_val, cur_minmax, quant_min, quant_max, symmetric, scale_out,
                     zero_point_out, zp_type};
    Workspace ws(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (aligned16(x)) {
        const int64_t n4 = n / 4;
        const int grid = grid_for(n4, kThreads * 4, g_obs_blocks);
        const TimingHook th = take_timing_hook(OSQ_TIME_OBSERVE_FLAT);
        hipExtLaunchKernelGGL(observe_flat_kernel, dim3(grid), dim3(kThreads), 0, st, th.start, th.stop, 0,
                              reinterpret_cast<const float4*>(x), n4,
                           x + n4 * 4, static_cast<int>(n - n4 * 4), ws.floats(kFamObserveFlat), ws.counter(kFamObserveFlat), fin);
    } else {
        // misaligned base: peel to the next 16-byte boundary by treating the head as the "tail" is not
        // possible with one pointer, so fall back to the per-channel kernel with a single channel.
        hipLaunchKernelGGL(observe_channels_kernel, dim3(1), dim3(kThreads), 0, st, x, int64_t(1), int64_t(1), n, fin);
    }
    return check_launch("observe_flat");
}

extern "C" int osq_observe_channels(const float* x, int64_t outer, int64_t channels, int64_t inner,
                                    int update_rule, int64_t cnt, float* min_val, float* max_val,
                                    int quant_min, int quant_max, int symmetric,
                                    float* scale_out, void* zero_point_out, int zp_type,
                                    osq_stream stream) {
    const char* why = "";
    OSQ_REQUIRE(outer > 0 && channels > 0 && inner > 0 && x, "observe_channels: empty tensor or null pointer");
    OSQ_REQUIRE(channels < (1ll << 31), "observe_channels: too many channels");
    OSQ_REQUIRE(check_finish_args(update_rule, min_val, max_val, &why), why);
    const Finish fin{update_rule, cnt, min_val, max_val, nullptr, quant_min, quant_max, symmetric, scale_out,
                     zero_point_out, zp_type};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (outer == 1 && inner % 4 == 0 && aligned16(x) && inner / 4 < (1 << 30)) {
        const int grid = grid_for(channels, kWavesPerBlock, kMaxBlocks * 4);
        const TimingHook th = take_timing_hook(OSQ_TIME_OBSERVE_CHANNELS);
        hipExtLaunchKernelGGL(observe_rows_kernel, dim3(grid), dim3(kThreads), 0, st, th.start, th.stop, 0,
                              reinterpret_cast<const float4*>(x), channels, static_cast<int>(inner / 4), fin);
    } else {
        hipLaunchKernelGGL(observe_channels_kernel, dim3(static_cast<unsigned>(channels)), dim3(kThreads), 0, st, x, outer,
                           channels, inner, fin);
    }
    return check_launch("observe_channels");
}

extern "C" int osq_token_minmax(const float* x, const osq_token_view* view, const int64_t* lengths,
                                float* token_min, float* token_max, osq_stream stream) {
    OSQ_REQUIRE(x && view && token_min && token_max, "token_minmax: null pointer");
    const osq_token_view v = *view;
    OSQ_REQUIRE(v.batch > 0 && v.tokens > 0 && v.feat_outer > 0 && v.feat_inner > 0, "token_minmax: empty view");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t ntok = v.batch * v.tokens;
    const int grid = grid_for(ntok, kWavesPerBlock, kMaxBlocks * 8);
    const bool vec = v.stride_inner == 1 && v.feat_inner % 4 == 0 && aligned16(x) && v.stride_batch % 4 == 0 &&
                     v.stride_token % 4 == 0 && (v.feat_outer == 1 || v.stride_outer % 4 == 0) &&
                     v.feat_inner / 4 < (1 << 30);
    if (vec) {
        const int inner4 = static_cast<int>(v.feat_inner / 4);
        int lgG = 6;
        if (v.feat_outer > 1) {
            lgG = 0;
            while ((1 << lgG) < inner4 && lgG < 6) ++lgG;
        }
        OSQ_REQUIRE(v.batch <= 65535, "token_minmax: batch exceeds grid.y");
        const dim3 tgrid(static_cast<unsigned>((v.tokens + kTokPerBlock - 1) / kTokPerBlock), static_cast<unsigned>(v.batch));
        const TimingHook th = take_timing_hook(OSQ_TIME_TOKEN_MINMAX);
#define OSQ_TOK(SEG, NT) \
    hipExtLaunchKernelGGL((token_minmax_vec_kernel<SEG, NT>), tgrid, dim3(kThreads), 0, st, th.start, th.stop, 0, x, v, lengths, \
                          token_min, token_max, lgG, inner4)
        if (v.feat_outer == 1) { if (g_tok_nt) OSQ_TOK(true, true); else OSQ_TOK(true, false); }
        else { if (g_tok_nt) OSQ_TOK(false, true); else OSQ_TOK(false, false); }
#undef OSQ_TOK
    } else {
        hipLaunchKernelGGL(token_minmax_generic_kernel, dim3(grid), dim3(kThreads), 0, st, x, v, lengths, token_min,
                           token_max);
    }
    return check_launch("token_minmax");
}

extern "C" int osq_token_minmax_multi(const osq_site_desc* descs, const int64_t* tok_end, int n_sites,
                                      int64_t total_tokens, osq_stream stream) {
    OSQ_REQUIRE(n_sites >= 0 && total_tokens >= 0, "token_minmax_multi: negative size");
    if (n_sites == 0 || total_tokens == 0) return OSQ_OK;
    OSQ_REQUIRE(descs && tok_end, "token_minmax_multi: null table");
    const int grid = grid_for(total_tokens, kWavesPerBlock, kMaxBlocks * 8);
    const TimingHook th = take_timing_hook(OSQ_TIME_TOKEN_MINMAX_MULTI);
    hipExtLaunchKernelGGL(token_minmax_multi_kernel, dim3(grid), dim3(kThreads), 0, static_cast<hipStream_t>(stream), th.start, th.stop, 0,
                          descs, tok_end, n_sites, total_tokens);
    return check_launch("token_minmax_multi");
}

extern "C" int osq_token_range_finalize(const float* token_min, const float* token_max,
                                        int64_t batch, int64_t tokens, const int64_t* lengths,
                                        int prune, double percentile,
                                        int update_rule, int64_t cnt, float* min_val, float* max_val,
                                        float* cur_minmax,
                                        int quant_min, int quant_max, int symmetric,
                                        float* scale_out, void* zero_point_out, int zp_type,
                                        void* workspace, void* list_scratch, osq_stream stream) {
    const char* why = "";
    OSQ_REQUIRE(token_min && token_max && batch > 0 && tokens > 0, "token_range_finalize: empty or null input");
    OSQ_REQUIRE(!prune || (percentile >= 0.0 && percentile <= 1.0), "token_range_finalize: percentile outside [0, 1]");
    OSQ_REQUIRE(check_finish_args(update_rule, min_val, max_val, &why), why);
    const Finish fin{update_rule, cnt, min_val, max_val, cur_minmax, quant_min, quant_max, symmetric, scale_out,
                     zero_point_out, zp_type};
    OSQ_REQUIRE(batch * tokens < (1ll << 31), "token_range_finalize: more than 2^31 token slots");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float qf = static_cast<float>(percentile);
    const int64_t per_thread = (batch * tokens + kFinalThreads - 1) / kFinalThreads;
    const bool wide = batch * tokens >= g_wide_min_slots && workspace && (list_scratch || !prune);
    if (!wide && workspace && select_fast_ok(token_min, token_max, batch, tokens, 0, 1)) {
        const SelectArgs a{token_min, token_max, batch, tokens, lengths, prune, qf, Workspace(workspace).meet(), g_select_shortcut};
        launch_select(st, a, fin, FinalBatch{0, 0, 0, nullptr, 0}, 1);
        return check_launch("token_range_finalize(select)");
    }
    if (wide) {
        Workspace wsp(workspace);
        WideArgs a{token_min, token_max, batch, tokens, lengths, reinterpret_cast<WideState*>(wsp.wide()),
                   static_cast<unsigned int*>(list_scratch), wsp.counter(kFamWideFinal), prune, qf};
        const int grid = static_cast<int>((batch * tokens + kWideSlotsPerBlock - 1) / kWideSlotsPerBlock);
        hipLaunchKernelGGL(wide_hist_kernel, dim3(grid), dim3(kWideThreads), 0, st, a, fin);
        if (prune) {
            hipLaunchKernelGGL(wide_collect_kernel, dim3(grid), dim3(kWideThreads), 0, st, a);
            hipLaunchKernelGGL(wide_select_kernel, dim3(grid), dim3(kWideThreads), 0, st, a, fin);
        }
        return check_launch("token_range_finalize(wide)");
    }
    const FinalBatch fb{0, 0, 0, nullptr, 0};
#define OSQ_LAUNCH_FINAL(S)                                                                                        \
    hipLaunchKernelGGL(token_finalize_kernel<S>, dim3(1), dim3(kFinalThreads), 0, st, token_min, token_max, batch, \
                       tokens, lengths, prune, qf, fin, fb)
    if (per_thread <= 4) OSQ_LAUNCH_FINAL(4);
    else if (per_thread <= 8) OSQ_LAUNCH_FINAL(8);
    else if (per_thread <= 16) OSQ_LAUNCH_FINAL(16);
    else if (per_thread <= 32) OSQ_LAUNCH_FINAL(32);
    else OSQ_LAUNCH_FINAL(0);
#undef OSQ_LAUNCH_FINAL
    return check_launch("token_range_finalize");
}

extern "C" int osq_observe_tokens(const float* x, const osq_token_view* view, const int64_t* lengths,
                                  float* token_min, float* token_max,
                                  int prune, double percentile,
                                  int update_rule, int64_t cnt, float* min_val, float* max_val,
                                  float* cur_minmax,
                                  int quant_min, int quant_max, int symmetric,
                                  float* scale_out, void* zero_point_out, int zp_type,
                                  void* workspace, void* list_scratch, osq_stream stream) {
    OSQ_REQUIRE(x && view && token_min && token_max, "observe_tokens: null pointer");
    const int rc = osq_token_minmax(x, view, lengths, token_min, token_max, stream);
    if (rc != OSQ_OK) return rc;
    return osq_token_range_finalize(token_min, token_max, view->batch, view->tokens, lengths, prune, percentile,
                                    update_rule, cnt, min_val, max_val, cur_minmax, quant_min, quant_max, symmetric,
                                    scale_out, zero_point_out, zp_type, workspace, list_scratch, stream);
}

// A whole quantizer call in the calibrate-and-quantize state (fake_quant.py:107-126 / 178-208 with both flags on)
// for a masked per-tensor activation.  Dense [B, T, H] rows: ONE persistent launch (fused_step.h); otherwise
// observe (two launches) then fake-quant with the refreshed parameters.

namespace osq {

// Workgroups that can be resident together: the fused kernel spins across workgroups, so the grid must not
// exceed what the device holds at once.  One 1024-thread workgroup per CU (its 16 waves own the CU's register file).
int persistent_grid_for(const void* kernel, int threads) {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus[dev] = n > 0 ? n : -1;
    }
    if (cus[dev] < 3) return 0;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        return 0;
    }
    return cus[dev];
}
static int fused_grid_for(const void* kernel) {
    int grid = persistent_grid_for(kernel, kFusedThreads);
    if (g_fused_grid >= 3 && g_fused_grid < grid) grid = g_fused_grid;
    return grid;
}

// Two fused launches must never be in flight together on one device: each wants every CU, and two half-resident
// grids would spin on each other until their time-outs.  Launches of one stream are ordered anyway; when the stream
// changes, the new stream first waits for everything the previous one has been given (one event, only at the switch).
// Other PROCESSES sharing the GPU cannot be ordered from here: set OSQ_FUSED_STEP=0 there.
bool persistent_serialize(hipStream_t st) {
    static std::mutex mu;
    static hipStream_t last[64];
    static hipEvent_t ev[64];
    static bool used[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (used[dev] && last[dev] != st) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cs);
        hipStreamCaptureStatus co = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(last[dev], &co);
        if (cs == hipStreamCaptureStatusNone && co == hipStreamCaptureStatusNone) {
            if (!ev[dev] && hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming) != hipSuccess) return false;
            if (hipEventRecord(ev[dev], last[dev]) != hipSuccess || hipStreamWaitEvent(st, ev[dev], 0) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
        }
    }
    used[dev] = true;
    last[dev] = st;
    return true;
}

static_assert(sizeof(FusedState) <= kWsFusedBytes, "FusedState must fit its slice of the workspace");

template <int NV>
static bool launch_fused(hipStream_t st, const FusedArgs& a, const Finish& fin) {
    static int grid = -1;       // per process; devices of one node are identical
    if (grid < 0 || g_fused_grid) grid = fused_grid_for(reinterpret_cast<const void*>(&observe_fq_fused_kernel<NV>));
    if (grid < 3) return false;
    // a selector wave holds 64 * 3 values of each of its sixteen chunks (fused_select<3>): with fewer workgroups than that
    // allows (a small "fused_grid", a device with few CUs) the call runs as three launches
    if ((a.B * a.T + grid - 3) / (grid - 2) > 3 * OSQ_WAVE || grid - 2 > kFusedWaves * kFusedWaves) return false;
    if (!persistent_serialize(st)) return false;
    const TimingHook th = take_timing_hook(OSQ_TIME_FUSED_STEP);
    hipExtLaunchKernelGGL(observe_fq_fused_kernel<NV>, dim3(grid), dim3(kFusedThreads), 0, st, th.start, th.stop, 0, a, fin);
    return true;
}

}  // namespace osq

extern "C" int osq_observe_tokens_fake_quant(const float* x, const osq_token_view* view, const int64_t* lengths,
                                             float* token_min, float* token_max,
                                             int prune, double percentile,
                                             int update_rule, int64_t cnt, float* min_val, float* max_val,
                                             float* cur_minmax,
                                             int quant_min, int quant_max, int symmetric,
                                             float* scale, void* zero_point, int zp_type,
                                             float* y, int64_t n, int mode, float grad_factor,
                                             void* workspace, void* list_scratch, osq_stream stream) {
    OSQ_REQUIRE(x && view && scale && zero_point && y, "observe_tokens_fake_quant: null pointer");
    const osq_token_view v = *view;
    const int64_t slots = v.batch * v.tokens;
    const bool dense_rows = v.feat_outer == 1 && v.stride_inner == 1 && v.stride_token == v.feat_inner &&
                            v.stride_batch == v.tokens * v.feat_inner && n == slots * v.feat_inner;
    const int64_t nv = v.feat_inner / 256;
    // OSQ_PARAM_NO_PERSISTENT in `mode`: THIS call runs as three ordinary launches whatever the process-wide switch says (a
    // caller with several streams or tenants on the device: the one-launch form wants every CU for itself)
    const bool want_persistent = g_fused_step && !(mode & OSQ_PARAM_NO_PERSISTENT);
    mode &= ~OSQ_PARAM_NO_PERSISTENT;
    if (want_persistent && workspace && token_min && token_max && dense_rows && v.feat_inner % 256 == 0 &&
        (nv == 3 || nv == 4 || nv == 12 || nv == 16) && v.batch >= 1 && v.batch <= kFusedMaxBatch && v.tokens >= 1 &&
        (slots & 3) == 0 && slots <= 4 * 8 * kSelThreads && n * 4 < (1ll << 32) && aligned16(x) && aligned16(y) && aligned16(token_min) && aligned16(token_max)) {
        const char* why = "";
        OSQ_REQUIRE(!prune || (percentile >= 0.0 && percentile <= 1.0), "observe_tokens_fake_quant: percentile outside [0, 1]");
        OSQ_REQUIRE(check_finish_args(update_rule, min_val, max_val, &why), why);
        OSQ_REQUIRE(quant_max > quant_min, "observe_tokens_fake_quant: quant_max must exceed quant_min");
        const Finish fin{update_rule, cnt, min_val, max_val, cur_minmax, quant_min, quant_max, symmetric, scale, zero_point, zp_type};
        const FusedArgs a{x, y, v.batch, v.tokens, lengths, token_min, token_max, prune, static_cast<float>(percentile),
                          g_select_shortcut, static_cast<FusedState*>(Workspace(workspace).fused()), scale, zero_point,
                          zp_type, mode, grad_factor, static_cast<float>(quant_min), static_cast<float>(quant_max), g_fused_gate, g_select_hint,
                          g_fused_spin_limit ? g_fused_spin_limit - 1u : kFusedSpinLimit};
        hipStream_t st = static_cast<hipStream_t>(stream);
        bool launched = false;
        switch (nv) {
            case 3: launched = launch_fused<3>(st, a, fin); break;
            case 4: launched = launch_fused<4>(st, a, fin); break;
            case 12: launched = launch_fused<12>(st, a, fin); break;
            default: launched = launch_fused<16>(st, a, fin); break;
        }
        if (launched) return check_launch("observe_tokens_fake_quant(fused)");
    }
    const int rc = osq_observe_tokens(x, view, lengths, token_min, token_max, prune, percentile, update_rule, cnt, min_val,
                                      max_val, cur_minmax, quant_min, quant_max, symmetric, scale, zero_point, zp_type, workspace,
                                      list_scratch, stream);
    if (rc != OSQ_OK) return rc;
    return osq_fake_quant_per_tensor(x, y, nullptr, n, scale, zero_point, zp_type, mode, grad_factor, quant_min, quant_max,
                                     stream);
}

extern "C" int osq_fused_step_status(void* workspace, int* status_out, osq_stream stream) {
    OSQ_REQUIRE(workspace && status_out, "fused_step_status: null pointer");
    FusedState* fs = static_cast<FusedState*>(Workspace(workspace).fused());
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned int v = 0u;
    if (hipMemcpyAsync(&v, &fs->status, sizeof(v), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        set_error("fused_step_status: copy failed");
        return OSQ_ERR_HIP;
    }
    // After a time-out the arrival flags are not trustworthy: a streaming workgroup that became resident only after
    // workgroup 2 had advanced the epoch stored the NEXT launch's tag.  Nothing is in flight after the synchronisation
    // above, so the whole block goes back to its all-zero start.
    if (v != 0u && (hipMemsetAsync(fs, 0, kWsFusedBytes, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) {
        set_error("fused_step_status: reset failed");
        return OSQ_ERR_HIP;
    }
    *status_out = static_cast<int>(v);
    return OSQ_OK;
}

/* Sticky time-out flags of BOTH persistent launch families on this workspace (the fused observe + fake-quant step and
 * the resident MSEFast searches), read and cleared after everything enqueued on `stream` has finished.  With
 * reset_on_error != 0 a non-zero flag also puts the two state blocks back to their all-zero start (nothing is in
 * flight on this workspace after the synchronisation): the next launch starts clean.  The fused step's block is wiped
 * after a time-out of that family in either case. */
extern "C" int osq_persistent_status(void* workspace, int* fused_status_out, int* resident_status_out, int reset_on_error,
                                     osq_stream stream) {
    OSQ_REQUIRE(workspace && fused_status_out && resident_status_out, "persistent_status: null pointer");
    const Workspace ws(workspace);
    FusedState* fs = static_cast<FusedState*>(ws.fused());
    unsigned int* rs_status = reinterpret_cast<unsigned int*>(static_cast<char*>(ws.resident()) + 64);   // ResidentState::status (msefast.hip)
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned int f = 0u, r = 0u;
    if (hipMemcpyAsync(&f, &fs->status, sizeof(f), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&r, rs_status, sizeof(r), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        set_error("persistent_status: copy failed");
        return OSQ_ERR_HIP;
    }
    if (f || r) {
        hipError_t e = hipSuccess;
        if (reset_on_error) e = hipMemsetAsync(ws.fused(), 0, kWsFusedBytes + kWsResidentBytes, st);
        else {
            // the fused step's block is wiped whenever ITS flag was raised (stale arrival flags, see osq_fused_step_status);
            // the resident searches' granules are tag-protected with a gap after a time-out: their flag alone is cleared
            if (f) e = hipMemsetAsync(ws.fused(), 0, kWsFusedBytes, st);
            if (e == hipSuccess && r) e = hipMemsetAsync(rs_status, 0, sizeof(r), st);
        }
        if (e != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            set_error("persistent_status: reset failed");
            return OSQ_ERR_HIP;
        }
    }
    *fused_status_out = static_cast<int>(f);
    *resident_status_out = static_cast<int>(r);
    return OSQ_OK;
}

extern "C" int osq_token_range_finalize_batched(const float* token_min, const float* token_max, int64_t problem_stride,
                                                int n_quantizers, int n_batches, int64_t batch, int64_t tokens,
                                                const int64_t* lengths, int lengths_per_quantizer,
                                                const int32_t* prune_flags, double percentile,
                                                float* cur_table, void* workspace, osq_stream stream) {
    OSQ_REQUIRE(token_min && token_max && cur_table && batch > 0 && tokens > 0, "token_range_finalize_batched: empty or null input");
    OSQ_REQUIRE(n_quantizers > 0 && n_batches > 0 && problem_stride >= batch * tokens, "token_range_finalize_batched: bad table shape");
    OSQ_REQUIRE(percentile >= 0.0 && percentile <= 1.0, "token_range_finalize_batched: percentile outside [0, 1]");
    OSQ_REQUIRE(batch * tokens < (1ll << 31), "token_range_finalize_batched: more than 2^31 token slots");
    const Finish fin{OSQ_UPDATE_NONE, 0, nullptr, nullptr, cur_table, 0, 1, 0, nullptr, nullptr, 0};
    const FinalBatch fb{problem_stride, n_batches, n_quantizers, prune_flags, lengths_per_quantizer ? 1 : 0};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float qf = static_cast<float>(percentile);
    const int64_t problems = static_cast<int64_t>(n_quantizers) * n_batches;
    if (workspace && select_fast_ok(token_min, token_max, batch, tokens, problem_stride, problems)) {
        const SelectArgs a{token_min, token_max, batch, tokens, lengths, 1, qf, Workspace(workspace).meet(), g_select_shortcut};
        launch_select(st, a, fin, fb, problems);
        return check_launch("token_range_finalize_batched(select)");
    }
    const int64_t per_thread = (batch * tokens + kFinalThreads - 1) / kFinalThreads;
    const dim3 grid(static_cast<unsigned>(n_quantizers) * static_cast<unsigned>(n_batches));
#define OSQ_LAUNCH_FINAL(S)                                                                                     \
    hipLaunchKernelGGL(token_finalize_kernel<S>, grid, dim3(kFinalThreads), 0, st, token_min, token_max, batch, \
                       tokens, lengths, 1, qf, fin, fb)
    if (per_thread <= 4) OSQ_LAUNCH_FINAL(4);
    else if (per_thread <= 8) OSQ_LAUNCH_FINAL(8);
    else if (per_thread <= 16) OSQ_LAUNCH_FINAL(16);
    else if (per_thread <= 32) OSQ_LAUNCH_FINAL(32);
    else OSQ_LAUNCH_FINAL(0);
#undef OSQ_LAUNCH_FINAL
    return check_launch("token_range_finalize_batched");
}

#ifdef OSQ_FINAL_TIMING
extern "C" int osq_debug_buffer(void* p) {
    long long* q = static_cast<long long*>(p);
    return hipMemcpyToSymbol(HIP_SYMBOL(osq::g_osq_dbg), &q, sizeof(q)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int osq_set_wide_min_slots(int64_t slots) {
    OSQ_REQUIRE(slots >= 1024, "set_wide_min_slots: threshold below 1024 slots");
    osq::g_wide_min_slots = slots;
    return OSQ_OK;
}
