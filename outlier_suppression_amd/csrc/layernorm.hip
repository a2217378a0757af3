// Residual + LayerNorm + fake-quant in ONE pass over the activation.
//
// The reference runs, per LayerNorm site of a quantized model (model/quant_bert.py:211-216, 298-303;
// model/util_layernorm.py:14-18, 32-37, 49-52):
//     r = input * gamma + hidden          GammaResidual               2 reads + 1 write
//     n = layer_norm(r) [* w + b]         LayerNorm                   1 read  + 1 write
//     n += beta / gamma                   split form only             1 read  + 1 write
//     y = fake_quantize(n)                8 eager ops                 1 read  + 1 write
// = 24..32 B per element.  Here: read input (+ hidden), write y: 8 (12) B per element.
//
// One wave per row; the row lives in registers (R float4 per lane, cols <= 4096); mean and variance are
// two wave reductions (sum, then sum of squared deviations -- the two-pass form, no cancellation);
// every arithmetic step is individually rounded (-ffp-contract=off) in the order of the eager ops:
//     a = input*gamma; r = a + hidden; t = (r - mean) * rstd; t = t*w; t = t + b; then quantize_value.
// The normalisation is not bit-comparable with torch's own kernel (Welford on the GPU, a different
// blocked summation on the CPU): callers that need the eager sequence keep using it (autograd passes do).
#include <string>
#include <hip/hip_ext.h>
#include "osq_device.h"
#include "osq_host.h"

namespace osq {

constexpr int kLnThreads = 256;
constexpr int kLnWaves = kLnThreads / OSQ_WAVE;
OSQ_AB_KNOB(int, g_ln_blocks, 1024);      // grid cap (osq_set_tuning("ln_blocks", n)); rows are grid-strided above it.  tools/bwd_ab.py on MI355X,
                                    // [256,128,768]: one row per wave (8192 workgroups) 58.2 us, 2048: 51.6, 1024: 51.2, 768: 51.1, 512: 60.0; [32,384,768]: 23.6 -> 21.1

template <int CTRL>
__device__ __forceinline__ float dpp_add_f32(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// symmetric xor / mirror patterns: after each step both partners hold the same partial sum, so every
// lane ends with the same value bit for bit
__device__ __forceinline__ float wave_sum_f32(float v) {
    v = dpp_add_f32<kDppQuadXor1>(v);
    v = dpp_add_f32<kDppQuadXor2>(v);
    v = dpp_add_f32<kDppRowHalfMirror>(v);
    v = dpp_add_f32<kDppRowMirror>(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

struct LnArgs {
    const float4* x;
    const float4* hidden;     // nullable: no residual
    const float4* gamma;      // nullable: plain residual add (only read when hidden != NULL)
    const float4* weight;     // nullable: no elementwise scale
    const float4* bias;       // nullable
    float4* y;
    int64_t rows;
    int cols4;
    float inv_cols, eps;
    float* scale;             // nullable: no fake-quant (written only under OSQ_PARAM_SANITIZE)
    void* zero_point;
    int zp_type, mode;
    float grad_factor, qmin, qmax;
};

template <int R>
__global__ __launch_bounds__(kLnThreads) void residual_layernorm_fq_kernel(LnArgs a) {
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * kLnThreads + threadIdx.x) / OSQ_WAVE;
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kLnWaves;
    QParams p{1.f, 0.f};
    if (a.scale) p = tensor_params(a.scale, a.zero_point, a.zp_type, a.mode, a.grad_factor, a.qmin, a.qmax);
    // per-column operands stay in registers across the rows of this wave while the row is short (<= 1024
    // columns: every LayerNorm of BERT / RoBERTa / BART); longer rows re-read them (L1/L2 hits) per row
    constexpr bool KEEP = R <= 4;
    constexpr int RK = KEEP ? R : 1;
    float4 g[RK], w[RK], b[RK];
    if (KEEP) {
#pragma unroll
        for (int k = 0; k < RK; ++k) {
            const int c = lane + k * OSQ_WAVE, cc = c < a.cols4 ? c : a.cols4 - 1;
            g[k] = (a.hidden && a.gamma) ? a.gamma[cc] : make_float4(1.f, 1.f, 1.f, 1.f);
            w[k] = a.weight ? a.weight[cc] : make_float4(1.f, 1.f, 1.f, 1.f);
            b[k] = a.bias ? a.bias[cc] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (int64_t row = wave0; row < a.rows; row += nwaves) {
        const float4* xr = a.x + row * a.cols4;
        const float4* hr = a.hidden ? a.hidden + row * a.cols4 : nullptr;
        float4 v[R], h[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int c = lane + k * OSQ_WAVE, cc = c < a.cols4 ? c : a.cols4 - 1;
            v[k] = load_stream(xr + cc);
            if (hr) h[k] = load_stream(hr + cc);
        }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if (hr) {
                if (a.gamma) {
                    const int c = lane + k * OSQ_WAVE;
                    const float4 gk = KEEP ? g[KEEP ? k : 0] : a.gamma[c < a.cols4 ? c : a.cols4 - 1];
                    v[k].x *= gk.x; v[k].y *= gk.y; v[k].z *= gk.z; v[k].w *= gk.w;
                }
                v[k].x += h[k].x; v[k].y += h[k].y; v[k].z += h[k].z; v[k].w += h[k].w;
            }
            if (lane + k * OSQ_WAVE < a.cols4) sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
        const float mean = wave_sum_f32(sum) * a.inv_cols;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if (lane + k * OSQ_WAVE < a.cols4) {
                const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
                sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
        const float var = wave_sum_f32(sq) * a.inv_cols;
        const float rstd = 1.0f / sqrtf(var + a.eps);
        float4* yr = a.y + row * a.cols4;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int c = lane + k * OSQ_WAVE;
            if (c < a.cols4) {
                float t[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                const float4 wk = KEEP ? w[KEEP ? k : 0] : (a.weight ? a.weight[c] : make_float4(1.f, 1.f, 1.f, 1.f));
                const float4 bk = KEEP ? b[KEEP ? k : 0] : (a.bias ? a.bias[c] : make_float4(0.f, 0.f, 0.f, 0.f));
                const float ww[4] = {wk.x, wk.y, wk.z, wk.w}, bb[4] = {bk.x, bk.y, bk.z, bk.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float u = (t[e] - mean) * rstd;
                    if (a.weight) u = u * ww[e];
                    if (a.bias) u = u + bb[e];
                    if (a.scale) u = dequantize_value(quantize_value(u, p.scale, p.zp, a.qmin, a.qmax), p.scale, p.zp);
                    t[e] = u;
                }
                store_stream(yr + c, make_float4(t[0], t[1], t[2], t[3]));
            }
        }
    }
}

bool set_layernorm_tuning(const char* key, int value) {
#ifdef OSQ_TUNABLE
    if (std::string(key) == "ln_blocks" && value >= 1) { g_ln_blocks = value; return true; }
#endif
    return false;
}

}  // namespace osq

using namespace osq;

extern "C" int osq_residual_layernorm_fake_quant(const float* x, const float* hidden, const float* gamma,
                                                 const float* weight, const float* bias, double eps,
                                                 float* y, int64_t rows, int64_t cols,
                                                 float* scale, void* zero_point, int zp_type,
                                                 int mode, float grad_factor, int quant_min, int quant_max,
                                                 osq_stream stream) {
    OSQ_REQUIRE(rows >= 0 && cols > 0, "residual_layernorm_fake_quant: bad shape");
    if (rows == 0) return OSQ_OK;
    OSQ_REQUIRE(x && y, "residual_layernorm_fake_quant: null tensor");
    OSQ_REQUIRE(!scale || zero_point, "residual_layernorm_fake_quant: scale without zero_point");
    OSQ_REQUIRE(!scale || ((mode & ~(OSQ_PARAM_MODE_MASK | OSQ_PARAM_SANITIZE)) == 0 && (mode & OSQ_PARAM_MODE_MASK) <= OSQ_PARAM_LSQPLUS),
                "residual_layernorm_fake_quant: bad mode");
    if (cols % 4 != 0 || cols > 4 * 16 * OSQ_WAVE || !aligned16(x) || !aligned16(y) || (hidden && !aligned16(hidden)) ||
        (gamma && !aligned16(gamma)) || (weight && !aligned16(weight)) || (bias && !aligned16(bias))) {
        set_error("residual_layernorm_fake_quant: needs cols %% 4 == 0, cols <= 4096 and 16-byte aligned operands");
        return OSQ_ERR_UNSUPPORTED;
    }
    LnArgs a{reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(hidden),
             reinterpret_cast<const float4*>(gamma), reinterpret_cast<const float4*>(weight),
             reinterpret_cast<const float4*>(bias), reinterpret_cast<float4*>(y), rows, static_cast<int>(cols / 4),
             static_cast<float>(1.0 / static_cast<double>(cols)), static_cast<float>(eps), scale, zero_point, zp_type, mode,
             grad_factor, static_cast<float>(quant_min), static_cast<float>(quant_max)};
    int64_t blocks = (rows + kLnWaves - 1) / kLnWaves;
    if (blocks > g_ln_blocks) blocks = g_ln_blocks;
    const dim3 grid(static_cast<unsigned>(blocks));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int per_lane = (a.cols4 + OSQ_WAVE - 1) / OSQ_WAVE;
    const TimingHook th = take_timing_hook(OSQ_TIME_LAYERNORM);
#define OSQ_LN(R) hipExtLaunchKernelGGL(residual_layernorm_fq_kernel<R>, grid, dim3(kLnThreads), 0, st, th.start, th.stop, 0, a)
    if (per_lane <= 1) OSQ_LN(1);
    else if (per_lane <= 2) OSQ_LN(2);
    else if (per_lane <= 3) OSQ_LN(3);
    else if (per_lane <= 4) OSQ_LN(4);
    else if (per_lane <= 8) OSQ_LN(8);
    else OSQ_LN(16);
#undef OSQ_LN
    return check_launch("residual_layernorm_fake_quant");
}
