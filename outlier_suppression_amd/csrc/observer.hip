// Observer reductions for gfx950 (MI355X): global / per-channel / per-token min-max,
// the token-wise percentile clipping of AvgPruneMinMaxObserver, running statistics and
// calculate_qparams.  Replaces quant_transformer/quantization/observer.py:50-237.
//
// One HBM read of the observed tensor (4 B/elem); padded tokens are never read.
// Wave64 shuffle reductions, LDS across waves, and a last-workgroup finaliser so
// that a whole observer call (reduce -> running statistic -> scale/zero_point) is
// one launch for flat / per-channel tensors and two for masked activations.
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

__device__ __forceinline__ void finish_entry(const Finish& f, int64_t idx, float cur_min, float cur_max) {
    if (f.cur) { f.cur[2 * idx] = cur_min; f.cur[2 * idx + 1] = cur_max; }
    float mn = cur_min, mx = cur_max;
    if (f.rule != OSQ_UPDATE_NONE && f.min_val && f.max_val) {
        apply_update(f.rule, f.cnt, cur_min, cur_max, &f.min_val[idx], &f.max_val[idx]);
        mn = f.min_val[idx];
        mx = f.max_val[idx];
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
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
        acc.add4(a); acc.add4(b); acc.add4(c); acc.add4(d);
    }
    for (; i < n4; i += stride) acc.add4(x[i]);
    if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < tail) acc.add(xt[threadIdx.x]);
    acc = block_reduce(acc);
    if (threadIdx.x == 0) {
        partials[3 * blockIdx.x] = acc.mn;
        partials[3 * blockIdx.x + 1] = acc.mx;
        partials[3 * blockIdx.x + 2] = acc.bad ? 1.0f : 0.0f;
    }
    if (grid_last_block(counter, gridDim.x)) {
        MinMax t;
        t.init();
        for (unsigned int k = threadIdx.x; k < gridDim.x; k += kThreads) {
            t.mn = fminf(t.mn, partials[3 * k]);
            t.mx = fmaxf(t.mx, partials[3 * k + 1]);
            t.bad |= partials[3 * k + 2] != 0.0f;
        }
        t = block_reduce(t);
        if (threadIdx.x == 0) {
            t.poison();
            finish_entry(fin, 0, t.mn, t.mx);
            *counter = 0u;
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

// Fast path: feat_inner contiguous (stride 1), everything 16-byte aligned.
// One wave per token.  The wave is split into 64/G lane groups of G lanes; group g
// walks feature segments g, g + 64/G, ...; inside a segment a lane reads 16 bytes at
// a time.  G = 64 when a segment has >= 64 float4 (e.g. [B,T,768]); G = 16 for
// head_dim 64 ([B,h,T,64] and its views), so that no lane idles on short segments.
__global__ __launch_bounds__(kThreads) void token_minmax_vec_kernel(const float* __restrict__ x, osq_token_view v,
                                                                    const int64_t* __restrict__ lengths,
                                                                    float* __restrict__ tok_min,
                                                                    float* __restrict__ tok_max, int lgG, int inner4) {
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    const int G = 1 << lgG;
    const int grp = lane >> lgG, li = lane & (G - 1), ngrp = OSQ_WAVE >> lgG;
    const int64_t ntok = v.batch * v.tokens;
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) / OSQ_WAVE;
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
    for (int64_t tok = wave0; tok < ntok; tok += nwaves) {
        const int64_t b = tok / v.tokens, t = tok - b * v.tokens;
        if (lengths && t >= lengths[b]) continue;   // padded token: never read
        const float* base = x + b * v.stride_batch + t * v.stride_token;
        MinMax acc;
        acc.init();
        if (v.feat_outer == 1) {
            const float4* p = reinterpret_cast<const float4*>(base);
            int j = lane;
            for (; j + 3 * OSQ_WAVE < inner4; j += 4 * OSQ_WAVE) {
                const float4 a = p[j], bb = p[j + OSQ_WAVE], c = p[j + 2 * OSQ_WAVE], d = p[j + 3 * OSQ_WAVE];
                acc.add4(a); acc.add4(bb); acc.add4(c); acc.add4(d);
            }
            for (; j < inner4; j += OSQ_WAVE) acc.add4(p[j]);
        } else {
            int64_t o = grp;
            for (; o + ngrp < v.feat_outer; o += 2 * ngrp) {      // two segments in flight
                const float4* p0 = reinterpret_cast<const float4*>(base + o * v.stride_outer);
                const float4* p1 = reinterpret_cast<const float4*>(base + (o + ngrp) * v.stride_outer);
                for (int j = li; j < inner4; j += G) {
                    const float4 a = p0[j], bb = p1[j];
                    acc.add4(a); acc.add4(bb);
                }
            }
            for (; o < v.feat_outer; o += ngrp) {
                const float4* p0 = reinterpret_cast<const float4*>(base + o * v.stride_outer);
                for (int j = li; j < inner4; j += G) acc.add4(p0[j]);
            }
        }
        acc.wave_reduce();
        if (lane == 0) {
            acc.poison();
            tok_min[tok] = acc.mn;
            tok_max[tok] = acc.mx;
        }
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

// ---------------------------------------------------------------- token range finaliser (K7b + K8 + K9)

// Single workgroup.  Valid slots: b*T + t with t < lengths[b] (all if lengths == NULL).
//
// prune: torch.quantile(|token_max|, p) needs the order statistics floor(rank) and
// ceil(rank), rank = fp32(p) * fp32(N-1).  They are found by a 4-pass, 8-bit radix
// select on the fp32 bit patterns of the absolute values (non-negative floats order
// like their bit patterns); the value at ceil(rank) is the same value when duplicates
// cover it and otherwise the smallest value above.  Interpolation is torch's lerp
// (one fused multiply-add per branch, pinned in tests/test_oracle_pinning.py).
struct Select {
    unsigned int prefix;     // bits fixed so far (high to low)
    unsigned int below;      // how many keys are smaller than every key matching the prefix
};

__device__ __forceinline__ bool slot_valid(int64_t slot, int64_t T, const int64_t* lengths, int64_t* b_cache,
                                           int64_t* len_cache) {
    if (!lengths) return true;
    const int64_t b = slot / T, t = slot - b * T;
    if (b != *b_cache) { *b_cache = b; *len_cache = lengths[b]; }
    return t < *len_cache;
}

__device__ __forceinline__ unsigned int abs_key(float v) { return __float_as_uint(v) & 0x7fffffffu; }

__global__ __launch_bounds__(kFinalThreads) void token_finalize_kernel(const float* __restrict__ tok_min,
                                                                       const float* __restrict__ tok_max, int64_t B,
                                                                       int64_t T, const int64_t* __restrict__ lengths,
                                                                       int prune, float q, Finish fin) {
    __shared__ unsigned int hist[2][256];
    __shared__ Select sel[2];
    __shared__ unsigned int s_count[2];        // keys <= selected value
    __shared__ unsigned int s_next[2];         // smallest key above the selected value
    __shared__ long long s_n;
    __shared__ int s_nan;
    __shared__ float s_res[2];

    const int64_t slots = B * T;
    const int tid = threadIdx.x;

    // ---- pass 0: N (valid tokens), NaN check, plain extrema
    MinMax plain;
    plain.init();
    long long my_n = 0;
    {
        int64_t bc = -1, lc = 0;
        for (int64_t s = tid; s < slots; s += kFinalThreads) {
            if (!slot_valid(s, T, lengths, &bc, &lc)) continue;
            ++my_n;
            const float a = tok_min[s], b = tok_max[s];
            plain.mn = fminf(plain.mn, a);
            plain.mx = fmaxf(plain.mx, b);
            plain.bad |= (a != a) || (b != b);
        }
    }
    if (tid == 0) { s_n = 0; s_nan = 0; }
    __syncthreads();
    {
        long long wn = my_n;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wn += __shfl_xor(wn, o, OSQ_WAVE);
        if ((tid & (OSQ_WAVE - 1)) == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&s_n), static_cast<unsigned long long>(wn));
    }
    plain = block_reduce(plain);
    if (tid == 0) { s_nan = plain.bad; s_res[0] = plain.mn; s_res[1] = plain.mx; }
    __syncthreads();
    const long long N = s_n;
    float cur_min = s_res[0], cur_max = s_res[1];

    if (N > 0 && prune && !s_nan) {
        const float rank = q * static_cast<float>(N - 1);
        const float rlo = floorf(rank);
        const unsigned int k_lo = static_cast<unsigned int>(rlo);
        const unsigned int k_hi = static_cast<unsigned int>(ceilf(rank));
        const float w = rank - rlo;

        if (tid < 2) { sel[tid].prefix = 0u; sel[tid].below = 0u; }
        // ---- 4 radix passes, both arrays at once: [0] = |token_max|, [1] = |token_min|
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            const unsigned int himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
            for (int k = tid; k < 512; k += kFinalThreads) hist[k >> 8][k & 255] = 0u;
            __syncthreads();
            const unsigned int p0 = sel[0].prefix, p1 = sel[1].prefix;
            int64_t bc = -1, lc = 0;
            for (int64_t s = tid; s < slots; s += kFinalThreads) {
                if (!slot_valid(s, T, lengths, &bc, &lc)) continue;
                const unsigned int k0 = abs_key(tok_max[s]), k1 = abs_key(tok_min[s]);
                if ((k0 & himask) == p0) atomicAdd(&hist[0][(k0 >> shift) & 255u], 1u);
                if ((k1 & himask) == p1) atomicAdd(&hist[1][(k1 >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < 2) {
                unsigned int below = sel[tid].below;
                unsigned int bin = 0;
                for (; bin < 256; ++bin) {
                    const unsigned int c = hist[tid][bin];
                    if (below + c > k_lo) break;
                    below += c;
                }
                sel[tid].below = below;
                sel[tid].prefix |= (bin << shift);
            }
            __syncthreads();
        }
        // ---- neighbour above the selected key, and how many keys are <= it
        if (tid < 2) { s_count[tid] = 0u; s_next[tid] = 0xffffffffu; }
        __syncthreads();
        const unsigned int v0 = sel[0].prefix, v1 = sel[1].prefix;
        {
            unsigned int c0 = 0, c1 = 0, n0 = 0xffffffffu, n1 = 0xffffffffu;
            int64_t bc = -1, lc = 0;
            for (int64_t s = tid; s < slots; s += kFinalThreads) {
                if (!slot_valid(s, T, lengths, &bc, &lc)) continue;
                const unsigned int k0 = abs_key(tok_max[s]), k1 = abs_key(tok_min[s]);
                if (k0 <= v0) ++c0; else n0 = min(n0, k0);
                if (k1 <= v1) ++c1; else n1 = min(n1, k1);
            }
            atomicAdd(&s_count[0], c0);
            atomicAdd(&s_count[1], c1);
            atomicMin(&s_next[0], n0);
            atomicMin(&s_next[1], n1);
        }
        __syncthreads();
        float thr[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const unsigned int vk = a == 0 ? v0 : v1;
            const float lo_v = __uint_as_float(vk);
            const float hi_v = (k_hi == k_lo || s_count[a] > k_hi) ? lo_v : __uint_as_float(s_next[a]);
            const float diff = hi_v - lo_v;
            thr[a] = (w < 0.5f) ? __builtin_fmaf(w, diff, lo_v) : __builtin_fmaf(w - 1.0f, diff, hi_v);
        }
        const float upper = thr[0], lower = -thr[1];
        // ---- up = max(token_max[token_max <= upper]) ; lo = min(token_min[token_min >= lower])
        MinMax pr;
        pr.init();
        {
            int64_t bc = -1, lc = 0;
            for (int64_t s = tid; s < slots; s += kFinalThreads) {
                if (!slot_valid(s, T, lengths, &bc, &lc)) continue;
                const float a = tok_min[s], b = tok_max[s];
                if (a >= lower) pr.mn = fminf(pr.mn, a);
                if (b <= upper) pr.mx = fmaxf(pr.mx, b);
            }
        }
        pr = block_reduce(pr);
        if (tid == 0) {
            // aminmax(clip(value, lo, up)) (observer.py:68,227): (lo, up), or (up, up) if lo > up
            s_res[0] = (pr.mn > pr.mx) ? pr.mx : pr.mn;
            s_res[1] = pr.mx;
        }
        __syncthreads();
        cur_min = s_res[0];
        cur_max = s_res[1];
    }
    if (tid == 0 && N > 0) {
        if (s_nan) { cur_min = __builtin_nanf(""); cur_max = __builtin_nanf(""); }
        finish_entry(fin, 0, cur_min, cur_max);
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
        const int grid = grid_for(n4, kThreads * 4, kMaxBlocks);
        hipLaunchKernelGGL(observe_flat_kernel, dim3(grid), dim3(kThreads), 0, st, reinterpret_cast<const float4*>(x), n4,
                           x + n4 * 4, static_cast<int>(n - n4 * 4), ws.floats(), ws.counter(1), fin);
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
        hipLaunchKernelGGL(observe_rows_kernel, dim3(grid), dim3(kThreads), 0, st, reinterpret_cast<const float4*>(x),
                           channels, static_cast<int>(inner / 4), fin);
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
        hipLaunchKernelGGL(token_minmax_vec_kernel, dim3(grid), dim3(kThreads), 0, st, x, v, lengths, token_min, token_max,
                           lgG, inner4);
    } else {
        hipLaunchKernelGGL(token_minmax_generic_kernel, dim3(grid), dim3(kThreads), 0, st, x, v, lengths, token_min,
                           token_max);
    }
    return check_launch("token_minmax");
}

extern "C" int osq_token_range_finalize(const float* token_min, const float* token_max,
                                        int64_t batch, int64_t tokens, const int64_t* lengths,
                                        int prune, double percentile,
                                        int update_rule, int64_t cnt, float* min_val, float* max_val,
                                        float* cur_minmax,
                                        int quant_min, int quant_max, int symmetric,
                                        float* scale_out, void* zero_point_out, int zp_type,
                                        osq_stream stream) {
    const char* why = "";
    OSQ_REQUIRE(token_min && token_max && batch > 0 && tokens > 0, "token_range_finalize: empty or null input");
    OSQ_REQUIRE(!prune || (percentile >= 0.0 && percentile <= 1.0), "token_range_finalize: percentile outside [0, 1]");
    OSQ_REQUIRE(check_finish_args(update_rule, min_val, max_val, &why), why);
    const Finish fin{update_rule, cnt, min_val, max_val, cur_minmax, quant_min, quant_max, symmetric, scale_out,
                     zero_point_out, zp_type};
    hipLaunchKernelGGL(token_finalize_kernel, dim3(1), dim3(kFinalThreads), 0, static_cast<hipStream_t>(stream), token_min,
                       token_max, batch, tokens, lengths, prune, static_cast<float>(percentile), fin);
    return check_launch("token_range_finalize");
}
