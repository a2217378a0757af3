// Device-side helpers shared by the gfx950 kernels.  Compiled with -ffp-contract=off:
// every fp32 operation below is individually rounded, as in the reference's eager
// PyTorch CPU path (SURVEY.md 8a "numerics contract").  The only fused operation is
// the explicit __builtin_fmaf of the quantile interpolation (torch's CPU lerp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/osq_hip.h"

#define OSQ_WAVE 64

namespace osq {

// A pointer the compiler cannot trace to a kernel argument (read from a table in memory) is GENERIC: its loads are flat_load_*,
// which tick lgkmcnt as well as vmcnt and so serialise against every LDS wait.  Device memory by contract -> address space 1.
typedef const float __attribute__((address_space(1)))* GlobalF32;
__device__ __forceinline__ GlobalF32 as_global(const float* p) { return (GlobalF32)p; }

struct QParams {   // effective parameters that reach the quantiser
    float scale;
    float zp;
};

__device__ __forceinline__ float load_zp(const void* zp, int zp_type, int64_t idx = 0) {
    return zp_type == OSQ_ZP_FLOAT32 ? static_cast<const float*>(zp)[idx]
                                     : static_cast<float>(static_cast<const int32_t*>(zp)[idx]);
}

// util_quant.py:70-71  grad_scale forward value: (t - t*g) + t*g
__device__ __forceinline__ float grad_scale_value(float t, float g) {
    const float tg = t * g;
    return (t - tg) + tg;
}

// util_quant.py:49-51 / 30: what the learnable variants hand to the quantiser
__device__ __forceinline__ QParams effective_params(float scale, float zp, int mode, float g) {
    if (mode == OSQ_PARAM_LSQPLUS) {
        zp = (rintf(zp) - zp) + zp;          // round_ste value
        scale = grad_scale_value(scale, g);
        zp = grad_scale_value(zp, g);
    } else if (mode == OSQ_PARAM_LSQ) {
        scale = grad_scale_value(scale, g);
    }
    return {scale, zp};
}

// LSQFakeQuantize / LSQPlusFakeQuantize.forward with the observer off first repair their parameters in
// place -- scale.abs_(); scale.clamp_(min=eps); zero_point.clamp_(qmin, qmax) (fake_quant.py:152-153,
// 188-191) -- and then quantise with them.  With OSQ_PARAM_SANITIZE in `mode` the consuming launch does
// both: every thread derives the repaired values itself (the repair is idempotent, so it does not matter
// whether a thread sees the raw or the already repaired word) and thread 0 of workgroup 0 writes them back.
constexpr float kLsqEps = 1.1920928955078125e-07f;     // torch.finfo(torch.float32).eps, the modules' `eps` buffer

__device__ __forceinline__ QParams tensor_params(float* scale_p, void* zp_p, int zp_type, int mode, float g,
                                                 float qmin, float qmax) {
    float s = scale_p[0], z = load_zp(zp_p, zp_type);
    const int base = mode & OSQ_PARAM_MODE_MASK;
    if (mode & OSQ_PARAM_SANITIZE) {
        s = fabsf(s);
        s = (s < kLsqEps) ? kLsqEps : s;                // clamp_(min=eps) keeps NaN
        if (base == OSQ_PARAM_LSQPLUS && zp_type == OSQ_ZP_FLOAT32) {
            z = (z < qmin) ? qmin : z;
            z = (z > qmax) ? qmax : z;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            scale_p[0] = s;
            if (base == OSQ_PARAM_LSQPLUS && zp_type == OSQ_ZP_FLOAT32) static_cast<float*>(zp_p)[0] = z;
        }
    }
    return effective_params(s, z, base, g);
}

// util_quant.py:12-13: x_int = round_ste(x/scale) + zp ; x_quant = clamp(x_int, qmin, qmax).
// True IEEE division; (rint(u) - u) + u reproduces round_ste's value for every input
// (== rint(u) for finite u, NaN for +-inf); the compare/select clamp lets NaN through
// like torch.clamp.
__device__ __forceinline__ float quantize_value(float x, float scale, float zp, float qmin, float qmax,
                                                float* x_int_out = nullptr) {
    const float u = x / scale;
    const float r = rintf(u);
    const float x_int = ((r - u) + u) + zp;
    if (x_int_out) *x_int_out = x_int;
    float q = x_int;
    q = (x_int < qmin) ? qmin : q;
    q = (x_int > qmax) ? qmax : q;
    return q;
}

// util_quant.py:14
__device__ __forceinline__ float dequantize_value(float q, float scale, float zp) { return (q - zp) * scale; }

// four elements of the scale-clip-round-dequant chain (util_quant.py:12-14)
__device__ __forceinline__ void fq4_plain(const float4& v, float4& y, float4& q, float s, float z, float qmin, float qmax) {
    q.x = quantize_value(v.x, s, z, qmin, qmax);
    q.y = quantize_value(v.y, s, z, qmin, qmax);
    q.z = quantize_value(v.z, s, z, qmin, qmax);
    q.w = quantize_value(v.w, s, z, qmin, qmax);
    y.x = dequantize_value(q.x, s, z);
    y.y = dequantize_value(q.y, s, z);
    y.z = dequantize_value(q.z, s, z);
    y.w = dequantize_value(q.w, s, z);
}

// torch.min / torch.max / torch.clamp propagate NaN; fminf / fmaxf drop it
__device__ __forceinline__ float min_nan(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fminf(a, b); }
__device__ __forceinline__ float max_nan(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }

// observer.py:101-119 for one entry (fp32).  A NaN statistic (a NaN among the observed values) yields NaN
// parameters, as in the reference: every step there is a NaN-propagating torch op.
__device__ __forceinline__ void qparams_from_range(float mn, float mx, int quant_min, int quant_max, int symmetric,
                                                   float* scale_out, float* zp_out) {
    const float min_neg = min_nan(mn, 0.0f);
    const float max_pos = max_nan(mx, 0.0f);
    const float eps = 1e-8f;
    float scale, zp;
    if (symmetric) {
        const float m = max_nan(-min_neg, max_pos);
        const float half = static_cast<float>(static_cast<double>(quant_max - quant_min) / 2.0);
        scale = max_nan(m / half, eps);
        zp = 0.0f;
    } else {
        scale = max_nan((max_pos - min_neg) / static_cast<float>(quant_max - quant_min), eps);
        zp = static_cast<float>(quant_min) - rintf(min_neg / scale);
        zp = min_nan(max_nan(zp, static_cast<float>(quant_min)), static_cast<float>(quant_max));
    }
    *scale_out = scale;
    *zp_out = zp;
}

// Minimum of clip(value, lo, up) (observer.py:68,227: aminmax of the clipped tensor) from the two selected bounds: lo, or up when
// lo > up.  Bounds that are zeros of DIFFERENT sign clip every element to a zero whose sign the reference leaves to the SIMD lane
// (oracle/observer_oracle.py, "zero extrema"): by the IEEE minimum rule, -0 < +0, the minimum is -0.0 -- the OR of the two words.
__device__ __forceinline__ float clipped_min(const float lo, const float up) {
    if (lo > up) return up;
    if (lo == up) return __uint_as_float(__float_as_uint(lo) | __float_as_uint(up));
    return lo;
}

__device__ __forceinline__ void store_zp(void* zp_out, int zp_type, int64_t idx, float zp) {
    if (zp_type == OSQ_ZP_FLOAT32) static_cast<float*>(zp_out)[idx] = zp;
    else static_cast<int32_t*>(zp_out)[idx] = static_cast<int32_t>(zp);
}

// observer.py:143-144 / 194-202 for one entry.  `cnt` = batches seen before this one.
__device__ __forceinline__ void apply_update(int rule, int64_t cnt, float cur_min, float cur_max,
                                             float* min_val, float* max_val) {
    if (rule == OSQ_UPDATE_RUNNING) {
        const float a = *min_val, b = *max_val;
        // torch.min/torch.max propagate NaN
        *min_val = (a != a || cur_min != cur_min) ? __builtin_nanf("") : fminf(a, cur_min);
        *max_val = (b != b || cur_max != cur_max) ? __builtin_nanf("") : fmaxf(b, cur_max);
    } else if (rule == OSQ_UPDATE_AVERAGE) {
        const float b = *max_val;
        float mn, mx;
        if (__builtin_isinf(b)) {                     // first batch: observer.py:194-196
            mn = cur_min; mx = cur_max;
        } else {
            const float c = static_cast<float>(cnt);  // tensor * python int
            mn = (*min_val) * c + cur_min;
            mx = b * c + cur_max;
        }
        const float d = static_cast<float>(cnt + 1);
        *min_val = mn / d;
        *max_val = mx / d;
    }
}

// ---- streaming (non-temporal) 16-byte accesses: data touched once should not displace
// lines that a later pass re-reads from L2 / Infinity Cache
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_stream(const float4* p) {
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_stream(float4* p, const float4& o) {
    v4f v;
    v.x = o.x; v.y = o.y; v.z = o.z; v.w = o.w;
    __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
}

// 16-byte WRITE-THROUGH (sc1) stores into a tensor of less than 4 GiB, addressed base (SGPR descriptor) + 32-bit byte offset.
// A kernel that streams a large output with plain or nt stores leaves dirty lines in the XCDs' L2s and pays for their
// write-back at its end (MI355X_MICROARCH.md, boundary row: + B / 6 TB/s for B dirty bytes); write-through stores leave
// nothing behind.  Measured: the fused observe + fake-quant step 40.9 -> 38.9 us.  The offset goes into the VGPR operand
// (see store_row16 in fused_step.h for why not into soffset).
typedef unsigned int osq_v4u32 __attribute__((ext_vector_type(4)));
struct WtStore {
    __amdgpu_buffer_rsrc_t rs;
    __device__ __forceinline__ WtStore(float4* base, int64_t n4)
        : rs(__builtin_amdgcn_make_buffer_rsrc(base, 0, static_cast<int>(static_cast<unsigned int>(n4 * 16)), 0x00020000)) {}
    __device__ __forceinline__ void put(int64_t i, const float4& o) const {
        osq_v4u32 w;
        w.x = __float_as_uint(o.x); w.y = __float_as_uint(o.y); w.z = __float_as_uint(o.z); w.w = __float_as_uint(o.w);
        __builtin_amdgcn_raw_buffer_store_b128(w, rs, static_cast<unsigned int>(i) * 16u, 0, 16 /* sc1 */);
    }
};
constexpr int64_t kWtMaxFloat4 = (1ll << 28) - 1;      // 4 GiB - 16 B

// ---- wave64 / block reductions -------------------------------------------------------

// DPP cross-lane moves run at VALU rate (no LDS crossbar round trip like ds_bpermute).
// For an idempotent, commutative op (min / max) four DPP steps leave every lane of a
// 16-lane row holding the row's result; the four row results are then read with
// v_readlane and combined, giving a wave-uniform value.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_value(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
constexpr int kDppQuadXor1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141;
constexpr int kDppRowMirror = 0x140;

__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, dpp_move<kDppQuadXor1>(v));
    v = fminf(v, dpp_move<kDppQuadXor2>(v));
    v = fminf(v, dpp_move<kDppRowHalfMirror>(v));
    v = fminf(v, dpp_move<kDppRowMirror>(v));
    return fminf(fminf(lane_value(v, 0), lane_value(v, 16)), fminf(lane_value(v, 32), lane_value(v, 48)));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<kDppQuadXor1>(v));
    v = fmaxf(v, dpp_move<kDppQuadXor2>(v));
    v = fmaxf(v, dpp_move<kDppRowHalfMirror>(v));
    v = fmaxf(v, dpp_move<kDppRowMirror>(v));
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}
template <int CTRL>
__device__ __forceinline__ unsigned int dpp_move_u32(unsigned int v) {
    return static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ unsigned int lane_value_u32(unsigned int v, int lane) {
    return static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(v), lane));
}
__device__ __forceinline__ unsigned int wave_min_u32(unsigned int v) {
    v = min(v, dpp_move_u32<kDppQuadXor1>(v));
    v = min(v, dpp_move_u32<kDppQuadXor2>(v));
    v = min(v, dpp_move_u32<kDppRowHalfMirror>(v));
    v = min(v, dpp_move_u32<kDppRowMirror>(v));
    return min(min(lane_value_u32(v, 0), lane_value_u32(v, 16)), min(lane_value_u32(v, 32), lane_value_u32(v, 48)));
}
__device__ __forceinline__ unsigned int wave_max_u32(unsigned int v) {
    v = max(v, dpp_move_u32<kDppQuadXor1>(v));
    v = max(v, dpp_move_u32<kDppQuadXor2>(v));
    v = max(v, dpp_move_u32<kDppRowHalfMirror>(v));
    v = max(v, dpp_move_u32<kDppRowMirror>(v));
    return max(max(lane_value_u32(v, 0), lane_value_u32(v, 16)), max(lane_value_u32(v, 32), lane_value_u32(v, 48)));
}
// wave64 inclusive add-scan with DPP (row_shr 1/2/4/8, then row_bcast15 / row_bcast31 into the
// following rows): six VALU-rate steps, no LDS traffic.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned int dpp_add_step(unsigned int v) {
    return v + static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ unsigned int wave_inclusive_scan_u32(unsigned int v) {
    v = dpp_add_step<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add_step<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add_step<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add_step<0x118, 0xf>(v);   // row_shr:8
    v = dpp_add_step<0x142, 0xa>(v);   // row_bcast:15 -> rows 1, 3
    v = dpp_add_step<0x143, 0xc>(v);   // row_bcast:31 -> rows 2, 3
    return v;
}

// order-preserving map fp32 -> u32 (for LDS atomicMin/atomicMax on floats; NaN handled separately)
__device__ __forceinline__ unsigned int ordered_bits(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned int u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
// double add-reduction with DPP: the xor / mirror patterns are symmetric, so after each step
// both partners hold the same partial sum; four row results are combined through readlane.
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}
__device__ __forceinline__ double lane_value_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane(static_cast<int>(b >> 32), lane);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move_f64<0xB1>(v);     // quad_perm:[1,0,3,2]
    v += dpp_move_f64<0x4E>(v);     // quad_perm:[2,3,0,1]
    v += dpp_move_f64<0x141>(v);    // row_half_mirror
    v += dpp_move_f64<0x140>(v);    // row_mirror
    return (lane_value_f64(v, 0) + lane_value_f64(v, 16)) + (lane_value_f64(v, 32) + lane_value_f64(v, 48));
}
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

// ---- ATen's CPU summation order (test modes: osq_set_tuning("mse_sum_order" / "bwd_sum_order", 8)) ----------------
// torch's CPU `sum` of a contiguous vector adds in the order of cascade_sum / vectorized_inner_sum
// (aten/src/ATen/native/cpu/SumKernel.cpp; restated and pinned against torch.sum in oracle/aten_sum.py): W SIMD lanes,
// lane l owning elements l, l + W, ...; per lane four interleaved accumulators, each a 4-level cascade; then the n % W
// trailing scalars and the W lanes, in order, onto a scalar.  W = 8 for fp32 and 4 for float64 (256-bit vectors, also on
// AVX-512 machines).  Lanes 0..W-1 of a wave play the SIMD lanes.
__device__ __forceinline__ int ceil_log2_i(int x) { return x <= 1 ? 0 : 32 - __builtin_clz(static_cast<unsigned int>(x - 1)); }

// lane < W: that SIMD lane's partial sum over vectors 0 .. n_vec-1 of sq (vector i = sq[i*W .. i*W + W-1])
template <typename T>
__device__ __forceinline__ T aten_lane_partial(const T* sq, int n_vec, int W, int lane) {
    constexpr int kLevels = 4, kIlp = 4;
    const int size = n_vec / kIlp;
    int level_power = ceil_log2_i(size) / kLevels;
    level_power = level_power < 4 ? 4 : level_power;
    const int level_step = 1 << level_power, level_mask = level_step - 1;
    T acc[kLevels][kIlp];
#pragma unroll
    for (int j = 0; j < kLevels; ++j)
#pragma unroll
        for (int k = 0; k < kIlp; ++k) acc[j][k] = T(0);
    int i = 0;
    while (i + level_step <= size) {
        for (int j = 0; j < level_step; ++j, ++i)
#pragma unroll
            for (int k = 0; k < kIlp; ++k) acc[0][k] = acc[0][k] + sq[(i * kIlp + k) * W + lane];
#pragma unroll
        for (int j = 1; j < kLevels; ++j) {
#pragma unroll
            for (int k = 0; k < kIlp; ++k) { acc[j][k] = acc[j][k] + acc[j - 1][k]; acc[j - 1][k] = T(0); }
            if ((i & (level_mask << (j * level_power))) != 0) break;
        }
    }
    for (; i < size; ++i)
#pragma unroll
        for (int k = 0; k < kIlp; ++k) acc[0][k] = acc[0][k] + sq[(i * kIlp + k) * W + lane];
#pragma unroll
    for (int j = 1; j < kLevels; ++j)
#pragma unroll
        for (int k = 0; k < kIlp; ++k) acc[0][k] = acc[0][k] + acc[j][k];
    for (int v = size * kIlp; v < n_vec; ++v) acc[0][0] = acc[0][0] + sq[v * W + lane];
#pragma unroll
    for (int k = 1; k < kIlp; ++k) acc[0][0] = acc[0][0] + acc[0][k];
    return acc[0][0];
}

// the whole wave calls; returns torch's sum (in T) of sq[0..n-1] (n >= W) in every lane
template <typename T>
__device__ __forceinline__ T aten_sum_wave(const T* sq, int n, int W) {
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    const int n_vec = n / W;
    const T part = lane < W ? aten_lane_partial<T>(sq, n_vec, W, lane) : T(0);
    T fin = T(0);
    for (int k = n_vec * W; k < n; ++k) fin = fin + sq[k];
    for (int l = 0; l < W; ++l) fin = fin + __shfl(part, l, OSQ_WAVE);
    return fin;
}
// n < W: scalar_inner_sum -- four interleaved scalar accumulators (the cascade never flushes below 16 x 4 elements), the
// left-overs onto the first, then the four in order.  Every lane computes it.
template <typename T>
__device__ __forceinline__ T aten_sum_short(const T* sq, int n) {
    T acc[4] = {T(0), T(0), T(0), T(0)};
    const int size = n / 4;
    for (int i = 0; i < size; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = acc[k] + sq[i * 4 + k];
    for (int i = size * 4; i < n; ++i) acc[0] = acc[0] + sq[i];
    return ((acc[0] + acc[1]) + acc[2]) + acc[3];
}
// ... and torch's mean: sum_out(...).div_(n) in T
template <typename T>
__device__ __forceinline__ T aten_mean_wave(const T* sq, int n, int W) { return aten_sum_wave<T>(sq, n, W) / static_cast<T>(n); }


// ---- last-workgroup-finishes pattern, without release fences ---------------------------
// A release fence at agent scope is `buffer_wbl2` = write back the whole XCD L2; issued by
// every workgroup of a 2048-block grid it serialises (measured: 108 us for a 96 MiB min/max
// that streams in 17 us).  Instead (cdna_hip_programming.md G16, "atomics both sides"): each
// workgroup publishes its partial with agent-scope relaxed atomic STORES (write-through, sc1),
// drains them (s_waitcnt vmcnt(0)), then takes a ticket with an agent-scope atomic; the last
// workgroup reads all partials with agent-scope atomic LOADS (served by L2/fabric, never a
// stale L1 line).  No fence, no L2 write-back.  *counter is 0 at launch and reset by the last
// workgroup.
__device__ __forceinline__ void publish_f32(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned int*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float consume_f32(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void publish_f64(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(__double_as_longlong(v)),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double consume_f64(const double* p) {
    return __longlong_as_double(static_cast<long long>(
        __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
}
// Arrival tickets are sharded: one device-scope counter sustains only ~80 arrivals/us
// (MI355X_MICROARCH.md "fanin": 11-13 ns per atomic), so 2048 workgroups finishing together
// would queue for ~25 us on a single word.  32 shard counters (different cache lines would be
// better still; different words already spread over L2 channels by address hash) take the
// arrivals in parallel; the last arriver of a shard takes a ticket on the top counter.
// counters[0] = top, counters[1 + s] = shard s; all zero at launch, reset by the last workgroup.
constexpr unsigned int kTicketShards = 32;
constexpr unsigned int kTicketStride = 16;   // one 64-byte line per counter

// Call from thread 0 AFTER it has published the workgroup's partial; every thread gets the answer.
// bid / nblocks: this workgroup's index among the workgroups that share `counters` (a launch that serves several
// independent reductions gives each its own range of workgroups and its own counters).
__device__ __forceinline__ bool grid_last_block(unsigned int* counters, unsigned int nblocks, unsigned int bid) {
    __shared__ unsigned int s_last;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the published partial has left this CU
        const unsigned int shards = nblocks < kTicketShards ? nblocks : kTicketShards;
        const unsigned int shard = bid % shards;
        const unsigned int members = (nblocks - shard + shards - 1) / shards;
        unsigned int last = 0u;
        const unsigned int t = __hip_atomic_fetch_add(&counters[(1 + shard) * kTicketStride], 1u, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
        if (t == members - 1) {
            const unsigned int top = __hip_atomic_fetch_add(&counters[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (top == shards - 1) ? 1u : 0u;
        }
        s_last = last;
    }
    __syncthreads();
    return s_last != 0u;
}
__device__ __forceinline__ bool grid_last_block(unsigned int* counters, unsigned int nblocks) {
    return grid_last_block(counters, nblocks, blockIdx.x);
}
// thread 0 of the last workgroup
__device__ __forceinline__ void grid_reset(unsigned int* counters, unsigned int nblocks) {
    const unsigned int shards = nblocks < kTicketShards ? nblocks : kTicketShards;
    __hip_atomic_store(&counters[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned int s = 0; s < shards; ++s)
        __hip_atomic_store(&counters[(1 + s) * kTicketStride], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace osq
