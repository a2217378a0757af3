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

// observer.py:101-119 for one entry (fp32)
__device__ __forceinline__ void qparams_from_range(float mn, float mx, int quant_min, int quant_max, int symmetric,
                                                   float* scale_out, float* zp_out) {
    const float min_neg = fminf(mn, 0.0f);
    const float max_pos = fmaxf(mx, 0.0f);
    const float eps = 1e-8f;
    float scale, zp;
    if (symmetric) {
        const float m = fmaxf(-min_neg, max_pos);
        const float half = static_cast<float>(static_cast<double>(quant_max - quant_min) / 2.0);
        scale = fmaxf(m / half, eps);
        zp = 0.0f;
    } else {
        scale = fmaxf((max_pos - min_neg) / static_cast<float>(quant_max - quant_min), eps);
        zp = static_cast<float>(quant_min) - rintf(min_neg / scale);
        zp = fminf(fmaxf(zp, static_cast<float>(quant_min)), static_cast<float>(quant_max));
    }
    *scale_out = scale;
    *zp_out = zp;
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

// ---- wave64 / block reductions -------------------------------------------------------

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, OSQ_WAVE));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, OSQ_WAVE));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, OSQ_WAVE);
    return v;
}
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

// Release this workgroup's global stores to the whole device, take a ticket, and tell
// the caller whether it is the last workgroup of the grid.  The last one has acquired
// every other workgroup's stores when this returns (cdna_hip_programming.md G16).
// *counter must be 0 at launch; the caller resets it when it is the last block.
__device__ __forceinline__ bool grid_last_block(unsigned int* counter, unsigned int nblocks) {
    __shared__ unsigned int s_ticket;
    __syncthreads();                      // every wave's partial stores are issued
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const bool last = (s_ticket == nblocks - 1);
    if (last) {
        if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
    }
    return last;
}

}  // namespace osq
