// The remaining observers of the reference's ObserverDict (quantized_module.py:10-20) on gfx950:
//   LSQPlusObserver      observer.py:148-173   range = mean -+ 3 std
//   AvgQuantileObserver  observer.py:240-282   2048-bin histogram of |x|, clip at a cumulative fraction
//   MSEObserver / Avg    observer.py:285-409   brute-force grid: 100 ranges (x zero-points) by MSE
// All stream the observed tensor (padded tokens skipped); the grid search evaluates 32 candidates per
// pass over the data instead of one.
#include <algorithm>
#include <string>
#include "osq_device.h"
#include "osq_host.h"

namespace osq {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / OSQ_WAVE;

// ---------------------------------------------------------------- element iteration helpers

struct ElemSource {          // either a flat dense tensor or a token view with valid lengths
    const float* x;
    int64_t n;               // flat: element count
    osq_token_view v;        // tokens: logical view (batch == 0 means "flat")
    const int64_t* lengths;
    int vec;
};

// calls f(value) for every observed element handled by this thread (workgroups of THREADS threads)
template <int THREADS, class F>
__device__ __forceinline__ void for_each_element_t(const ElemSource& s, F f) {
    constexpr int kThreads = THREADS, kWaves = THREADS / OSQ_WAVE;      // shadow the file's 256-thread constants
    if (s.v.batch == 0) {
        const int64_t n4 = s.n / 4;
        const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
        if ((reinterpret_cast<uintptr_t>(s.x) & 15u) == 0) {
            const float4* x4 = reinterpret_cast<const float4*>(s.x);
            for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4; i += stride) {
                const float4 a = x4[i];
                f(a.x); f(a.y); f(a.z); f(a.w);
            }
            if (blockIdx.x == 0 && static_cast<int64_t>(threadIdx.x) < s.n - n4 * 4) f(s.x[n4 * 4 + threadIdx.x]);
        } else {
            for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < s.n; i += stride) f(s.x[i]);
        }
        return;
    }
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    const int64_t ntok = s.v.batch * s.v.tokens;
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) / OSQ_WAVE;
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWaves;
    const int64_t F_ = s.v.feat_outer * s.v.feat_inner;
    for (int64_t tok = wave0; tok < ntok; tok += nwaves) {
        const int64_t b = tok / s.v.tokens, t = tok - b * s.v.tokens;
        if (s.lengths && t >= s.lengths[b]) continue;
        const float* base = s.x + b * s.v.stride_batch + t * s.v.stride_token;
        if (s.vec) {
            const int inner4 = static_cast<int>(s.v.feat_inner / 4);
            const int64_t F4 = s.v.feat_outer * inner4;
            for (int64_t j = lane; j < F4; j += OSQ_WAVE) {
                const int64_t o = j / inner4, i = j - o * inner4;
                const float4 a = reinterpret_cast<const float4*>(base + o * s.v.stride_outer)[i];
                f(a.x); f(a.y); f(a.z); f(a.w);
            }
        } else {
            for (int64_t j = lane; j < F_; j += OSQ_WAVE) {
                const int64_t o = j / s.v.feat_inner, i = j - o * s.v.feat_inner;
                f(base[o * s.v.stride_outer + i * s.v.stride_inner]);
            }
        }
    }
}

template <class F>
__device__ __forceinline__ void for_each_element(const ElemSource& s, F f) { for_each_element_t<kThreads>(s, f); }

__device__ __forceinline__ double observed_count(const ElemSource& s) {
    if (s.v.batch == 0) return static_cast<double>(s.n);
    int64_t tot = 0;
    for (int64_t b = 0; b < s.v.batch; ++b) {
        int64_t l = s.lengths ? s.lengths[b] : s.v.tokens;
        l = l < 0 ? 0 : (l > s.v.tokens ? s.v.tokens : l);
        tot += l;
    }
    return static_cast<double>(tot * s.v.feat_outer * s.v.feat_inner);
}

// block sum of K doubles per thread -> partials[block][K]; last block adds them up into sums[K] (LDS)
template <int K>
__device__ __forceinline__ bool grid_sum(const double (&acc)[K], double* partials, unsigned int* tickets, double* sums) {
    __shared__ double sh[kWaves][K];
    const int lane = threadIdx.x & (OSQ_WAVE - 1), wv = threadIdx.x / OSQ_WAVE;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) sh[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double a = 0.0;
        for (int w = 0; w < kWaves; ++w) a += sh[w][threadIdx.x];
        publish_f64(&partials[static_cast<int64_t>(blockIdx.x) * K + threadIdx.x], a);
    }
    // threads 0..K-1 (K <= 64) are one wave: thread 0's s_waitcnt inside grid_last_block drains all their stores
    if (!grid_last_block(tickets, gridDim.x)) return false;
    for (int k = threadIdx.x; k < K; k += kThreads) sums[k] = 0.0;
    __syncthreads();
    if (threadIdx.x < K) {
        double a = 0.0;
        for (unsigned int b = 0; b < gridDim.x; ++b) a += consume_f64(&partials[static_cast<int64_t>(b) * K + threadIdx.x]);
        sums[threadIdx.x] = a;
    }
    __syncthreads();
    return true;
}

// ---------------------------------------------------------------- LSQPlusObserver

struct QOut {
    int quant_min, quant_max, symmetric;
    float* scale;
    void* zp;
    int zp_type;
};

__device__ __forceinline__ void write_moment_range(double sum, double sumsq, double n, int64_t idx, float* min_val,
                                                   float* max_val, const QOut& q) {
    const double mean_d = sum / n;
    const double var_d = n > 1.0 ? (sumsq - n * mean_d * mean_d) / (n - 1.0) : __builtin_nan("");
    const float mean = static_cast<float>(mean_d);
    const float sd = static_cast<float>(sqrt(var_d > 0.0 ? var_d : (var_d == var_d ? 0.0 : var_d)));
    const float three = 3.0f * sd;                      // mean - 3*std / mean + 3*std in fp32 (observer.py:171-172)
    const float mn = mean - three, mx = mean + three;
    min_val[idx] = mn;
    max_val[idx] = mx;
    if (q.scale) {
        float s, z;
        qparams_from_range(mn, mx, q.quant_min, q.quant_max, q.symmetric, &s, &z);
        q.scale[idx] = s;
        if (q.zp) store_zp(q.zp, q.zp_type, idx, z);
    }
}

__global__ __launch_bounds__(kThreads) void moments_flat_kernel(ElemSource src, float* min_val, float* max_val, QOut q,
                                                                double* partials, unsigned int* tickets) {
    __shared__ double sums[2];
    double acc[2] = {0.0, 0.0};
    float s1 = 0.f, s2 = 0.f;
    int run = 0;
    for_each_element(src, [&](float v) {
        s1 += v; s2 += v * v;
        if (++run == 32) { acc[0] += s1; acc[1] += s2; s1 = 0.f; s2 = 0.f; run = 0; }   // short fp32 runs, double totals
    });
    acc[0] += s1; acc[1] += s2;
    if (grid_sum<2>(acc, partials, tickets, sums)) {
        if (threadIdx.x == 0) {
            write_moment_range(sums[0], sums[1], observed_count(src), 0, min_val, max_val, q);
            grid_reset(tickets, gridDim.x);
        }
    }
}

// x viewed as [outer, channels, inner]; one workgroup per channel
__global__ __launch_bounds__(kThreads) void moments_channels_kernel(const float* __restrict__ x, int64_t outer,
                                                                    int64_t channels, int64_t inner, float* min_val,
                                                                    float* max_val, QOut q) {
    __shared__ double sh[2][kWaves];
    const int64_t c = blockIdx.x;
    double a0 = 0.0, a1 = 0.0;
    for (int64_t o = 0; o < outer; ++o) {
        const float* p = x + (o * channels + c) * inner;
        for (int64_t j = threadIdx.x; j < inner; j += kThreads) { const double v = p[j]; a0 += v; a1 += v * v; }
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    const int lane = threadIdx.x & (OSQ_WAVE - 1), wv = threadIdx.x / OSQ_WAVE;
    if (lane == 0) { sh[0][wv] = a0; sh[1][wv] = a1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0, ss = 0.0;
        for (int k = 0; k < kWaves; ++k) { s += sh[0][k]; ss += sh[1][k]; }
        write_moment_range(s, ss, static_cast<double>(outer * inner), c, min_val, max_val, q);
    }
}

// ---------------------------------------------------------------- AvgQuantileObserver

// torch.linspace element i (fp32, scalar form) -- the histogram's bin edges (observer.py:263 via torch.histc)
__device__ __forceinline__ float linspace_edge(int i, float start, float end, int steps) {
    const float step = (end - start) / static_cast<float>(steps - 1);
    return i < steps / 2 ? start + step * static_cast<float>(i) : end - step * static_cast<float>(steps - i - 1);
}

// torch.histc bin of v in [lo, hi]: linear estimate, then the local search against the edges; last bin closed
__device__ __forceinline__ int histc_bin(float v, float lo, float hi, int bins) {
    int pos = static_cast<int>((v - lo) * static_cast<float>(bins) / (hi - lo));
    int a = pos - 1 > 0 ? pos - 1 : 0;
    const int b = pos + 2 < bins + 1 ? pos + 2 : bins + 1;
    int j = a - 1;                                        // largest j in [a, b) with edge(j) <= v
    for (; a < b; ++a)
        if (linspace_edge(a, lo, hi, bins + 1) <= v) j = a;
    if (j == bins) j -= 1;
    return j;
}

constexpr int kHistBins = 2048;

__global__ __launch_bounds__(kThreads) void abs_hist_kernel(ElemSource src, const float* __restrict__ cur_minmax,
                                                            unsigned int* __restrict__ hist) {
    __shared__ unsigned int lh[kHistBins];
    for (int k = threadIdx.x; k < kHistBins; k += kThreads) lh[k] = 0u;
    __syncthreads();
    const float hi = fmaxf(-cur_minmax[0], cur_minmax[1]);   // max_hist_range (observer.py:262)
    for_each_element(src, [&](float v) {
        const float a = fabsf(v);
        if (a <= hi && hi > 0.0f) {
            const int j = histc_bin(a, 0.0f, hi, kHistBins);
            if (j >= 0) atomicAdd(&lh[j], 1u);
        }
    });
    __syncthreads();
    for (int k = threadIdx.x; k < kHistBins; k += kThreads)
        if (lh[k]) atomicAdd(&hist[k], lh[k]);
}

__global__ __launch_bounds__(kThreads) void quantile_finalize_kernel(unsigned int* __restrict__ hist, ElemSource src,
                                                                     double threshold, const float* __restrict__ cur_minmax,
                                                                     int rule, int64_t cnt, float* min_val, float* max_val,
                                                                     QOut q) {
    __shared__ unsigned int s_tot[kWaves];
    __shared__ int s_bin;
    constexpr int per = kHistBins / kThreads;      // 8 consecutive bins per thread
    const int lane = threadIdx.x & (OSQ_WAVE - 1), wv = threadIdx.x / OSQ_WAVE;
    unsigned int h[per], mine = 0u;
#pragma unroll
    for (int k = 0; k < per; ++k) { h[k] = hist[threadIdx.x * per + k]; mine += h[k]; }
    const unsigned int incl_w = wave_inclusive_scan_u32(mine);
    if (threadIdx.x == 0) s_bin = kHistBins;
    if (lane == OSQ_WAVE - 1) s_tot[wv] = incl_w;
    __syncthreads();
    unsigned int base = 0u;
#pragma unroll
    for (int k = 0; k < kWaves; ++k) base += (k < wv) ? s_tot[k] : 0u;
    // counts are integers below 2^24 for any realistic site, so the reference's sequential fp32 running
    // total (observer.py:266-271) equals the exact integer prefix
    const float target = static_cast<float>(threshold * observed_count(src));
    unsigned int cum = base + incl_w - mine;
#pragma unroll
    for (int k = 0; k < per; ++k) {
        cum += h[k];
        if (static_cast<float>(cum) >= target) { atomicMin(&s_bin, threadIdx.x * per + k); break; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < per; ++k) hist[threadIdx.x * per + k] = 0u;     // leave the table zeroed for the next call
    if (threadIdx.x == 0) {
        const float mn = cur_minmax[0], mx = cur_minmax[1];
        const float range = fmaxf(-mn, mx);
        float clip = range;
        if (s_bin < kHistBins) clip = (static_cast<float>(s_bin) + 0.5f) * (range / static_cast<float>(kHistBins));
        const float cmin = fmaxf(mn, -clip), cmax = fminf(mx, clip);
        apply_update(rule, cnt, cmin, cmax, &min_val[0], &max_val[0]);
        if (q.scale) {
            float s, z;
            qparams_from_range(min_val[0], max_val[0], q.quant_min, q.quant_max, q.symmetric, &s, &z);
            q.scale[0] = s;
            if (q.zp) store_zp(q.zp, q.zp_type, 0, z);
        }
    }
}

// ---------------------------------------------------------------- MSEObserver (grid search)

constexpr int kCandBatch = 32;

struct Cand { float lo, hi, scale, zp, rcp; int fast; };

// x / s for a divisor shared by many dividends, WITHOUT the division: y = RN(1 / s) once, q0 = x * y, then two rounds of
// (r = x - q * s exactly, by fma; q += r * y).  The second round's result is the correctly rounded quotient -- the bits
// of the IEEE division -- whenever nothing over- or underflows on the way and the significand of s is not all ones
// (Markstein 1990; Muller et al., Handbook of Floating-Point Arithmetic, 2nd ed., Theorem 4.9 / section 4.7.2; the float64
// search of msefast.hip uses the same sequence).  The guards: s in [2^-60, 2^60] with a significand that is not all ones
// (div_fast_divisor, per candidate), |x| in [2^-60, 2^60] or x == 0 (div_fast_dividend, per element); everything else
// takes the division.  Five full-rate fma-class instructions instead of v_rcp_f32 (quarter rate) + v_div_scale x 2 + four
// fma + v_div_fmas + v_div_fixup, whose VCC hand-over also keeps the chains of neighbouring candidates from interleaving:
// the grid search is VALU-bound on exactly this.  osq_selftest_division compares the two on caller-given operands.
__device__ __forceinline__ bool div_fast_divisor(float s) {
    return s >= 8.6736174e-19f && s <= 1.1529215e18f && (__float_as_uint(s) & 0x7fffffu) != 0x7fffffu;      // false for NaN
}
__device__ __forceinline__ bool div_fast_dividend(float x) {
    const float a = fabsf(x);
    return a <= 1.1529215e18f && (a >= 8.6736174e-19f || a == 0.0f);                                         // false for NaN / inf
}
__device__ __forceinline__ float div_by_reciprocal(float x, float s, float y) {
    const float q0 = x * y;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, s, x), y, q0);
    return __builtin_fmaf(__builtin_fmaf(-q1, s, x), y, q1);
}
// the squared error of one element under one candidate (observer.py:292-312), the quotient by either route; `fast` is
// wave-uniform (every lane's element passes its guard and the candidate passes its own): a real branch, not a select --
// the division stays out of line so that the compiler cannot fold the two routes into "compute both"
__device__ __attribute__((noinline)) float grid_true_division(float v, float s) { return v / s; }
__device__ __forceinline__ float grid_sq_err(float v, bool fast, float scale, float rcp, float zp, float qmin, float qmax) {
    float u;
    if (fast) u = div_by_reciprocal(v, scale, rcp);
    else u = grid_true_division(v, scale);
    const float r = rintf(u);
    const float x_int = ((r - u) + u) + zp;
    float q = x_int;
    q = (x_int < qmin) ? qmin : q;
    q = (x_int > qmax) ? qmax : q;
    const float d = fabsf(dequantize_value(q, scale, zp) - v);
    return d * d;
}
__device__ __forceinline__ float uniform_f32(float v) {
    return __uint_as_float(static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(__float_as_uint(v)))));
}

// candidate k of perform_1D_search / perform_2D_search (observer.py:314-364), fp32 throughout
__device__ __forceinline__ Cand make_candidate(int k, float x_min, float x_max, int qmin, int qmax, int sym, int side,
                                               int two_d, int num) {
    Cand c;
    if (!two_d) {
        const float xr = fmaxf(fabsf(x_min), x_max);
        const float thres = xr / static_cast<float>(num) * static_cast<float>(k + 1);
        c.lo = side == 1 ? 0.0f : -thres;
        c.hi = side == 2 ? 0.0f : thres;
    } else {
        const int nz = qmax - qmin + 1;
        const int i = k / nz + 1, zp = qmin + k % nz;
        const float xr = x_max - x_min;
        const float tmp_max = xr / static_cast<float>(num) * static_cast<float>(i);
        const float delta = (tmp_max - 0.0f) / static_cast<float>(qmax - qmin);
        c.lo = fmaxf(0.0f - static_cast<float>(zp) * delta, x_min);
        c.hi = fminf(tmp_max - static_cast<float>(zp) * delta, x_max);
    }
    qparams_from_range(c.lo, c.hi, qmin, qmax, sym, &c.scale, &c.zp);
    c.zp = static_cast<float>(static_cast<int>(c.zp));
    c.rcp = 1.0f / c.scale;
    c.fast = div_fast_divisor(c.scale) ? 1 : 0;
    return c;
}

struct GridArgs {
    int quant_min, quant_max, symmetric, side, two_d, num, n_cand;
};

__global__ __launch_bounds__(kThreads) void mse_grid_loss_kernel(ElemSource src, const float* __restrict__ cur_minmax,
                                                                 GridArgs g, int k0, float* __restrict__ losses,
                                                                 double* partials, unsigned int* tickets) {
    __shared__ double sums[kCandBatch];
    __shared__ Cand cands[kCandBatch];
    if (threadIdx.x < kCandBatch) {
        const int k = k0 + threadIdx.x < g.n_cand ? k0 + threadIdx.x : g.n_cand - 1;
        cands[threadIdx.x] = make_candidate(k, cur_minmax[0], cur_minmax[1], g.quant_min, g.quant_max, g.symmetric, g.side,
                                            g.two_d, g.num);
    }
    __syncthreads();
    const float qmin = static_cast<float>(g.quant_min), qmax = static_cast<float>(g.quant_max);
    float part[kCandBatch];
#pragma unroll
    for (int k = 0; k < kCandBatch; ++k) part[k] = 0.0f;
    double acc[kCandBatch];
#pragma unroll
    for (int k = 0; k < kCandBatch; ++k) acc[k] = 0.0;
    int run = 0;
    for_each_element(src, [&](float v) {
#pragma unroll
        for (int k = 0; k < kCandBatch; ++k) {
            const float y = dequantize_value(quantize_value(v, cands[k].scale, cands[k].zp, qmin, qmax), cands[k].scale, cands[k].zp);
            const float d = fabsf(y - v);
            part[k] += d * d;
        }
        if (++run == 16) {
#pragma unroll
            for (int k = 0; k < kCandBatch; ++k) { acc[k] += part[k]; part[k] = 0.0f; }
            run = 0;
        }
    });
#pragma unroll
    for (int k = 0; k < kCandBatch; ++k) acc[k] += part[k];
    if (grid_sum<kCandBatch>(acc, partials, tickets, sums)) {
        const double n = observed_count(src);
        if (threadIdx.x < kCandBatch && k0 + threadIdx.x < g.n_cand) losses[k0 + threadIdx.x] = static_cast<float>(sums[threadIdx.x] / n);
        __syncthreads();
        if (threadIdx.x == 0) grid_reset(tickets, gridDim.x);
    }
}

// ALL candidates in ONE launch (round 4).  The launch-per-batch form above runs 200 launches of 256 workgroups x 256
// threads for the asymmetric grid -- one wave per SIMD, every division chain exposed -- and measured 30 ms on a
// [32,128,768] site.  Here a workgroup has 1024 threads (16 waves per CU: the chains of four waves overlap on every SIMD),
// keeps its share of the elements for the whole search and walks the candidates 16 at a time itself: per batch of
// candidates one block reduction and 16 doubles to partials[workgroup][candidate]; mse_grid_reduce_kernel then adds the
// workgroups' partials in workgroup order (deterministic) and leaves the losses the commit kernel reads.  Same
// candidates, same fp32 squared errors, same 16-term fp32 runs inside a float64 sum as above.
#ifndef OSQ_GRIDALL_THREADS
#define OSQ_GRIDALL_THREADS 1024
#endif
#ifndef OSQ_GRIDALL_CANDS
#define OSQ_GRIDALL_CANDS 16
#endif
#ifndef OSQ_GRIDALL_BARRIER
#define OSQ_GRIDALL_BARRIER 1
#endif
constexpr int kGridAllThreads = OSQ_GRIDALL_THREADS;
constexpr int kGridAllWaves = kGridAllThreads / OSQ_WAVE;
constexpr int kGridAllMaxBlocks = 256;
constexpr int kGridAllMaxBatch = 1024;       // prefix sums of the lengths in LDS (the dealing of a dense masked site by pieces)
constexpr int kCandAll = OSQ_GRIDALL_CANDS;          // candidates per trip.  Measured on [32,128,768] / [32,128,3072] (tools/mse_grid_ab.py, ms): 1024 threads x 16: 21.8 / 40.2 (43 spilled VGPRs and still the fastest); 1024 x 8: 56 / 40; 256 x 32: 29 / 78; 512 x 16: 55 / 52; the launch-per-32-candidates form 30.2 / 76

__global__ __launch_bounds__(kGridAllThreads) void mse_grid_all_kernel(ElemSource src, const float* __restrict__ cur_minmax,
                                                                      GridArgs g, int n_pad, double* __restrict__ partials) {
    __shared__ Cand cands[kCandAll];
    __shared__ double sh[kGridAllWaves][kCandAll];
    __shared__ unsigned int pre[kGridAllMaxBatch + 1];
    const int lane = threadIdx.x & (OSQ_WAVE - 1), wv = threadIdx.x / OSQ_WAVE;
    const float qmin = static_cast<float>(g.quant_min), qmax = static_cast<float>(g.quant_max);
    // A masked dense site whose rows are whole 1 KiB pieces (768 / 1024 / 3072 / 4096 features): the VALID tokens' pieces
    // (one float4 per lane) are dealt round the waves of the grid -- piece u to wave u mod waves -- instead of one token
    // per wave: with 54 % of the tokens valid every wave is busy (token dealing left the waves of padded tokens idle),
    // and a piece is a finer unit than a token.  Only the dealing differs; the values and the candidates are the same.
    const bool pieces = src.v.batch > 0 && src.vec && src.v.feat_outer == 1 && src.v.feat_inner % 256 == 0 && src.v.batch <= kGridAllMaxBatch;
    unsigned int n_units = 0u, segs = 1u;
    if (pieces) {
        const unsigned int Bu = static_cast<unsigned int>(src.v.batch);
        for (unsigned int b = threadIdx.x; b < Bu; b += kGridAllThreads) {
            int64_t l = src.lengths ? src.lengths[b] : src.v.tokens;
            l = l < 0 ? 0 : (l > src.v.tokens ? src.v.tokens : l);
            pre[b + 1] = static_cast<unsigned int>(l);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            pre[0] = 0u;
            for (unsigned int b = 0; b < Bu; ++b) pre[b + 1] += pre[b];
        }
        __syncthreads();
        segs = static_cast<unsigned int>(src.v.feat_inner / 256);
        n_units = pre[Bu] * segs;
    }
    for (int k0 = 0; k0 < g.n_cand; k0 += kCandAll) {
        if (threadIdx.x < kCandAll) {
            const int k = k0 + threadIdx.x < g.n_cand ? k0 + threadIdx.x : g.n_cand - 1;
            cands[threadIdx.x] = make_candidate(k, cur_minmax[0], cur_minmax[1], g.quant_min, g.quant_max, g.symmetric, g.side,
                                                g.two_d, g.num);
        }
        __syncthreads();
        float part[kCandAll];
        double acc[kCandAll];
        float c_scale[kCandAll], c_rcp[kCandAll], c_zp[kCandAll];      // wave-uniform: scalar registers
        unsigned int c_fast = 0u;
#pragma unroll
        for (int k = 0; k < kCandAll; ++k) {
            part[k] = 0.0f; acc[k] = 0.0;
            c_scale[k] = uniform_f32(cands[k].scale); c_rcp[k] = uniform_f32(cands[k].rcp); c_zp[k] = uniform_f32(cands[k].zp);
            c_fast |= cands[k].fast ? (1u << k) : 0u;
        }
        c_fast = static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(c_fast)));
        int run = 0;
        auto one = [&](float v) {
            const bool all_ok = !wave_any(!div_fast_dividend(v));          // uniform: one ballot per element, shared by the candidates
#pragma unroll
            for (int k = 0; k < kCandAll; ++k)
                part[k] += grid_sq_err(v, all_ok && ((c_fast >> k) & 1u), c_scale[k], c_rcp[k], c_zp[k], qmin, qmax);
            if (OSQ_GRIDALL_BARRIER) __builtin_amdgcn_sched_barrier(0);   // one value's chains at a time: interleaving several costs the registers
            if (++run == 16) {
#pragma unroll
                for (int k = 0; k < kCandAll; ++k) { acc[k] += part[k]; part[k] = 0.0f; }
                run = 0;
            }
        };
        if (pieces) {
            const unsigned int Bu = static_cast<unsigned int>(src.v.batch), nwaves = gridDim.x * kGridAllWaves;
            for (unsigned int u = blockIdx.x * kGridAllWaves + wv; u < n_units; u += nwaves) {
                const unsigned int j = u / segs, seg = u - j * segs;           // valid token j (sample-major), its piece
                unsigned int lo = 0u, hi = Bu;                                  // pre[lo] <= j < pre[hi]
                while (lo + 1u < hi) {
                    const unsigned int mid = (lo + hi) >> 1;
                    if (pre[mid] <= j) lo = mid; else hi = mid;
                }
                const float4 a = reinterpret_cast<const float4*>(src.x + static_cast<int64_t>(lo) * src.v.stride_batch +
                                                                 static_cast<int64_t>(j - pre[lo]) * src.v.stride_token + seg * 256u)[lane];
                one(a.x); one(a.y); one(a.z); one(a.w);
            }
        } else {
            for_each_element_t<kGridAllThreads>(src, one);
        }
#pragma unroll
        for (int k = 0; k < kCandAll; ++k) {
            const double v = wave_sum(acc[k] + static_cast<double>(part[k]));
            if (lane == 0) sh[wv][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < kCandAll) {
            double a = 0.0;
            for (int w = 0; w < kGridAllWaves; ++w) a += sh[w][threadIdx.x];
            partials[static_cast<int64_t>(blockIdx.x) * n_pad + k0 + threadIdx.x] = a;
        }
        // the next batch's candidates / sh are written only after the barrier at its top
    }
}

__global__ __launch_bounds__(kThreads) void mse_grid_reduce_kernel(const double* __restrict__ partials, int n_blocks, int n_pad,
                                                                   ElemSource src, GridArgs g, float* __restrict__ losses) {
    __shared__ double s_n;
    if (threadIdx.x == 0) s_n = observed_count(src);
    __syncthreads();
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= g.n_cand) return;
    double a = 0.0;
    for (int b = 0; b < n_blocks; ++b) a += partials[static_cast<int64_t>(b) * n_pad + k];
    losses[k] = static_cast<float>(a / s_n);
}

// first strict minimum in candidate order (observer.py:339-341,359-361), then commit + qparams
__global__ __launch_bounds__(kThreads) void mse_grid_commit_kernel(const float* __restrict__ losses,
                                                                   const float* __restrict__ cur_minmax, GridArgs g,
                                                                   int rule, int64_t cnt, float* min_val, float* max_val,
                                                                   QOut q) {
    __shared__ float s_best[kThreads];
    __shared__ int s_idx[kThreads];
    float best = 1e10f;          // initial best_score (observer.py:325,353)
    int idx = -1;
    for (int k = threadIdx.x; k < g.n_cand; k += kThreads) {
        const float l = losses[k];
        if (l < best) { best = l; idx = k; }
    }
    s_best[threadIdx.x] = best;
    s_idx[threadIdx.x] = idx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 1; t < kThreads; ++t) {
            const bool better = s_idx[t] >= 0 && (s_best[t] < best || (s_best[t] == best && (idx < 0 || s_idx[t] < idx)));
            if (better) { best = s_best[t]; idx = s_idx[t]; }
        }
        float bmin = cur_minmax[0], bmax = cur_minmax[1];
        if (idx >= 0) {
            const Cand c = make_candidate(idx, cur_minmax[0], cur_minmax[1], g.quant_min, g.quant_max, g.symmetric, g.side,
                                          g.two_d, g.num);
            bmin = c.lo; bmax = c.hi;
        }
        apply_update(rule, cnt, bmin, bmax, &min_val[0], &max_val[0]);
        if (q.scale) {
            float s, z;
            qparams_from_range(min_val[0], max_val[0], q.quant_min, q.quant_max, q.symmetric, &s, &z);
            q.scale[0] = s;
            if (q.zp) store_zp(q.zp, q.zp_type, 0, z);
        }
    }
}

// per-channel 1-D / 2-D grid: one wave per row, all candidates inside the wave
__global__ __launch_bounds__(kThreads) void mse_grid_rows_kernel(const float* __restrict__ w, int64_t rows, int cols,
                                                                 GridArgs g, float* __restrict__ best_min,
                                                                 float* __restrict__ best_max) {
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    const int64_t row = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) / OSQ_WAVE;
    if (row >= rows) return;
    const float* xr = w + row * cols;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (int j = lane; j < cols; j += OSQ_WAVE) { const float v = xr[j]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (g.two_d) { mx = fmaxf(mx, 0.0f); mn = fminf(mn, 0.0f); }        // observer.py:319-321
    const float qmin = static_cast<float>(g.quant_min), qmax = static_cast<float>(g.quant_max);
    float best = 1e10f, bmin = mn, bmax = mx;
    for (int k = 0; k < g.n_cand; ++k) {
        const Cand c = make_candidate(k, mn, mx, g.quant_min, g.quant_max, g.symmetric, g.side, g.two_d, g.num);
        float part = 0.0f;
        for (int j = lane; j < cols; j += OSQ_WAVE) {
            const float v = xr[j];
            const float y = dequantize_value(quantize_value(v, c.scale, c.zp, qmin, qmax), c.scale, c.zp);
            const float d = fabsf(y - v);
            part += d * d;
        }
        const float loss = static_cast<float>(wave_sum(static_cast<double>(part)) / static_cast<double>(cols));
        if (loss < best) { best = loss; bmin = c.lo; bmax = c.hi; }
    }
    if (lane == 0) { best_min[row] = bmin; best_max[row] = bmax; }
}

static inline int grid_for(int64_t items, int per_block, int max_blocks) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return static_cast<int>(b);
}

static inline bool make_source(const float* x, int64_t n, const osq_token_view* view, const int64_t* lengths, ElemSource* s,
                               int* grid, int max_blocks) {
    s->x = x;
    s->n = n;
    s->lengths = lengths;
    s->vec = 0;
    s->v = osq_token_view{0, 0, 0, 0, 0, 0, 0, 0};
    if (view) {
        s->v = *view;
        if (s->v.batch <= 0 || s->v.tokens <= 0 || s->v.feat_outer <= 0 || s->v.feat_inner <= 0) return false;
        s->vec = s->v.stride_inner == 1 && s->v.feat_inner % 4 == 0 && aligned16(x) && s->v.stride_batch % 4 == 0 &&
                 s->v.stride_token % 4 == 0 && (s->v.feat_outer == 1 || s->v.stride_outer % 4 == 0);
        *grid = grid_for(s->v.batch * s->v.tokens, kWaves, max_blocks);
    } else {
        if (n <= 0) return false;
        *grid = grid_for(n / 4 + 1, kThreads * 4, max_blocks);
    }
    return true;
}

}  // namespace osq

using namespace osq;

extern "C" int osq_observe_moments(const float* x, int64_t outer, int64_t channels, int64_t inner,
                                   float* min_val, float* max_val, int quant_min, int quant_max, int symmetric,
                                   float* scale_out, void* zero_point_out, int zp_type,
                                   void* workspace, osq_stream stream) {
    OSQ_REQUIRE(x && min_val && max_val && outer > 0 && channels > 0 && inner > 0, "observe_moments: empty or null input");
    OSQ_REQUIRE(channels < (1ll << 31), "observe_moments: too many channels");
    const QOut q{quant_min, quant_max, symmetric, scale_out, zero_point_out, zp_type};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (channels == 1) {
        OSQ_REQUIRE(workspace, "observe_moments: per-tensor form needs the workspace");
        ElemSource src;
        int grid = 1;
        make_source(x, outer * inner, nullptr, nullptr, &src, &grid, 1024);
        Workspace ws(workspace);
        hipLaunchKernelGGL(moments_flat_kernel, dim3(grid), dim3(kThreads), 0, st, src, min_val, max_val, q, ws.doubles(kFamMoments),
                           ws.counter(kFamMoments));
    } else {
        hipLaunchKernelGGL(moments_channels_kernel, dim3(static_cast<unsigned>(channels)), dim3(kThreads), 0, st, x, outer,
                           channels, inner, min_val, max_val, q);
    }
    return check_launch("observe_moments");
}

extern "C" int osq_observe_quantile(const float* x, int64_t n, const osq_token_view* view, const int64_t* lengths,
                                    const float* cur_minmax, double threshold, uint32_t* hist_scratch,
                                    int update_rule, int64_t cnt, float* min_val, float* max_val,
                                    int quant_min, int quant_max, int symmetric,
                                    float* scale_out, void* zero_point_out, int zp_type, osq_stream stream) {
    OSQ_REQUIRE(x && cur_minmax && hist_scratch && min_val && max_val, "observe_quantile: null pointer");
    OSQ_REQUIRE(update_rule == OSQ_UPDATE_RUNNING || update_rule == OSQ_UPDATE_AVERAGE, "observe_quantile: bad rule");
    ElemSource src;
    int grid = 1;
    OSQ_REQUIRE(make_source(x, n, view, lengths, &src, &grid, 256), "observe_quantile: empty input");
    const QOut q{quant_min, quant_max, symmetric, scale_out, zero_point_out, zp_type};
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(abs_hist_kernel, dim3(grid), dim3(kThreads), 0, st, src, cur_minmax, hist_scratch);
    hipLaunchKernelGGL(quantile_finalize_kernel, dim3(1), dim3(kThreads), 0, st, hist_scratch, src, threshold, cur_minmax,
                       update_rule, cnt, min_val, max_val, q);
    return check_launch("observe_quantile");
}

namespace osq {
__global__ void selftest_division_kernel(const float* __restrict__ x, const float* __restrict__ sc, int64_t n, int32_t* __restrict__ mismatches) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[i], b = sc[i];
    if (!div_fast_divisor(b) || !div_fast_dividend(a)) return;      // operands the kernels send down the division anyway
    const float want = a / b, got = div_by_reciprocal(a, b, 1.0f / b);
    if (__float_as_uint(want) != __float_as_uint(got)) atomicAdd(mismatches, 1);
}
}  // namespace osq

/* Test aid: counts the pairs (x[i], s[i]) -- among those both guards admit -- for which the reciprocal sequence of the MSE
 * grid (div_by_reciprocal) and the IEEE division differ in any bit.  mismatches: device int32, added to. */
extern "C" int osq_selftest_division(const float* x, const float* s, int64_t n, int32_t* mismatches, osq_stream stream) {
    OSQ_REQUIRE(x && s && mismatches && n >= 0, "selftest_division: bad argument");
    if (n == 0) return OSQ_OK;
    hipLaunchKernelGGL(selftest_division_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, s, n, mismatches);
    return check_launch("selftest_division");
}

OSQ_AB_KNOB(int, g_mse_grid_all, 1);       // osq_set_tuning("mse_grid_all", 0): the per-tensor MSE grid as one launch per 32 candidates (A/B, tests)
namespace osq {
bool set_extra_tuning(const char* key, int value) {
#ifdef OSQ_TUNABLE
    if (std::string(key) == "mse_grid_all") { g_mse_grid_all = value != 0; return true; }
#endif
    return false;
}
}  // namespace osq

extern "C" size_t osq_mse_grid_scratch_bytes(int quant_min, int quant_max, int two_d) {
    const int n_cand = two_d ? 100 * (quant_max - quant_min + 1) : 100;
    const size_t n_pad = static_cast<size_t>((n_cand + kCandBatch - 1) / kCandBatch * kCandBatch);
    return n_pad * sizeof(double) + static_cast<size_t>(kGridAllMaxBlocks) * n_pad * sizeof(double);
}

extern "C" int osq_mse_grid_tensor(const float* x, int64_t n, const osq_token_view* view, const int64_t* lengths,
                                   const float* cur_minmax, int quant_min, int quant_max, int symmetric,
                                   int one_side, int two_d, float* loss_scratch, size_t scratch_bytes,
                                   int update_rule, int64_t cnt, float* min_val, float* max_val,
                                   float* scale_out, void* zero_point_out, int zp_type,
                                   void* workspace, osq_stream stream) {
    OSQ_REQUIRE(x && cur_minmax && loss_scratch && min_val && max_val && workspace, "mse_grid_tensor: null pointer");
    OSQ_REQUIRE(update_rule == OSQ_UPDATE_RUNNING || update_rule == OSQ_UPDATE_AVERAGE, "mse_grid_tensor: bad rule");
    ElemSource src;
    int grid = 1;
    OSQ_REQUIRE(make_source(x, n, view, lengths, &src, &grid, 256), "mse_grid_tensor: empty input");   // 256 x 32 partials = 64 KiB
    GridArgs g{quant_min, quant_max, symmetric, one_side, two_d, 100, two_d ? 100 * (quant_max - quant_min + 1) : 100};
    const QOut q{quant_min, quant_max, symmetric, scale_out, zero_point_out, zp_type};
    hipStream_t st = static_cast<hipStream_t>(stream);
    Workspace ws(workspace);
    const int n_pad = (g.n_cand + kCandBatch - 1) / kCandBatch * kCandBatch;
    if (g_mse_grid_all && scratch_bytes >= osq_mse_grid_scratch_bytes(quant_min, quant_max, two_d)) {
        // all candidates in ONE launch: [losses n_pad floats][partials 256 x n_pad doubles] in the caller's scratch
        const int64_t items = view ? src.v.batch * src.v.tokens : (n + 4 * kGridAllThreads - 1) / (4 * kGridAllThreads);
        const int blocks = static_cast<int>(std::min<int64_t>(kGridAllMaxBlocks, std::max<int64_t>(1, view ? (items + kGridAllWaves - 1) / kGridAllWaves : items)));
        double* partials = reinterpret_cast<double*>(reinterpret_cast<char*>(loss_scratch) + static_cast<size_t>(n_pad) * sizeof(double));
        hipLaunchKernelGGL(mse_grid_all_kernel, dim3(blocks), dim3(kGridAllThreads), 0, st, src, cur_minmax, g, n_pad, partials);
        hipLaunchKernelGGL(mse_grid_reduce_kernel, dim3((g.n_cand + kThreads - 1) / kThreads), dim3(kThreads), 0, st, partials, blocks, n_pad, src, g,
                           loss_scratch);
    } else {
        for (int k0 = 0; k0 < g.n_cand; k0 += kCandBatch)
            hipLaunchKernelGGL(mse_grid_loss_kernel, dim3(grid), dim3(kThreads), 0, st, src, cur_minmax, g, k0, loss_scratch,
                               ws.doubles(kFamHistogram), ws.counter(kFamHistogram));
    }
    hipLaunchKernelGGL(mse_grid_commit_kernel, dim3(1), dim3(kThreads), 0, st, loss_scratch, cur_minmax, g, update_rule, cnt,
                       min_val, max_val, q);
    return check_launch("mse_grid_tensor");
}

extern "C" int osq_mse_grid_candidates(int quant_min, int quant_max, int two_d) {
    return two_d ? 100 * (quant_max - quant_min + 1) : 100;
}

extern "C" int osq_mse_grid_rows(const float* w, int64_t rows, int64_t cols, int quant_min, int quant_max, int symmetric,
                                 int one_side, int two_d, float* best_min, float* best_max, osq_stream stream) {
    OSQ_REQUIRE(w && best_min && best_max && rows > 0 && cols > 0 && cols < (1ll << 31), "mse_grid_rows: bad argument");
    GridArgs g{quant_min, quant_max, symmetric, one_side, two_d, 100, two_d ? 100 * (quant_max - quant_min + 1) : 100};
    const int grid = static_cast<int>((rows + kWaves - 1) / kWaves);
    hipLaunchKernelGGL(mse_grid_rows_kernel, dim3(grid), dim3(kThreads), 0, static_cast<hipStream_t>(stream), w, rows,
                       static_cast<int>(cols), g, best_min, best_max);
    return check_launch("mse_grid_rows");
}
