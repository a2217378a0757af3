// Gamma-Migration arithmetic for gfx950 (MI355X).
// Replaces quant_transformer/solver/gamma_migration.py:70-71 (fold LayerNorm gamma into
// the next linear's weight columns) and quant_transformer/model/util_layernorm.py:27,49-52.
#include "osq_device.h"
#include "osq_host.h"

namespace osq {

constexpr int kThreads = 256;

// W[r, c] *= gamma[c]; one 16-byte access per lane when cols % 4 == 0
__global__ __launch_bounds__(kThreads) void gamma_fold_vec_kernel(float4* __restrict__ w, const float4* __restrict__ gamma,
                                                                  int64_t n4, int cols4) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4; i += stride) {
        const float4 g = gamma[i % cols4];
        float4 v = w[i];
        v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
        w[i] = v;
    }
}

__global__ __launch_bounds__(kThreads) void gamma_fold_kernel(float* __restrict__ w, const float* __restrict__ gamma,
                                                              int64_t n, int64_t cols) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) w[i] *= gamma[i % cols];
}

__global__ void split_bias_kernel(const float* __restrict__ beta, const float* __restrict__ gamma, float* __restrict__ out,
                                  int64_t n) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) out[i] = beta[i] / gamma[i];
}

// out = input * gamma + hidden (two roundings, as the eager reference: util_layernorm.py:50-52)
__global__ __launch_bounds__(kThreads) void gamma_residual_kernel(const float* __restrict__ input,
                                                                  const float* __restrict__ hidden,
                                                                  const float* __restrict__ gamma, float* __restrict__ out,
                                                                  int64_t n, int64_t cols) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
        float a = input[i];
        if (gamma) a = a * gamma[i % cols];
        out[i] = a + hidden[i];
    }
}

__global__ __launch_bounds__(kThreads) void gamma_residual_vec_kernel(const float4* __restrict__ input,
                                                                      const float4* __restrict__ hidden,
                                                                      const float4* __restrict__ gamma,
                                                                      float4* __restrict__ out, int64_t n4, int cols4) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4; i += stride) {
        float4 a = input[i];
        const float4 h = hidden[i];
        if (gamma) {
            const float4 g = gamma[i % cols4];
            a.x *= g.x; a.y *= g.y; a.z *= g.z; a.w *= g.w;
        }
        a.x += h.x; a.y += h.y; a.z += h.z; a.w += h.w;
        out[i] = a;
    }
}

static inline int grid_for(int64_t items, int per_block) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return static_cast<int>(b);
}

}  // namespace osq

using namespace osq;

extern "C" int osq_gamma_fold(float* weight, const float* gamma, int64_t rows, int64_t cols, osq_stream stream) {
    OSQ_REQUIRE(rows >= 0 && cols >= 0, "gamma_fold: negative size");
    const int64_t n = rows * cols;
    if (n == 0) return OSQ_OK;
    OSQ_REQUIRE(weight && gamma, "gamma_fold: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cols % 4 == 0 && aligned16(weight) && aligned16(gamma) && cols / 4 < (1ll << 31))
        hipLaunchKernelGGL(gamma_fold_vec_kernel, dim3(grid_for(n / 4, kThreads)), dim3(kThreads), 0, st,
                           reinterpret_cast<float4*>(weight), reinterpret_cast<const float4*>(gamma), n / 4,
                           static_cast<int>(cols / 4));
    else
        hipLaunchKernelGGL(gamma_fold_kernel, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, st, weight, gamma, n, cols);
    return check_launch("gamma_fold");
}

extern "C" int osq_gamma_split_bias(const float* beta, const float* gamma, float* bias_out, int64_t n, osq_stream stream) {
    OSQ_REQUIRE(n >= 0, "gamma_split_bias: negative size");
    if (n == 0) return OSQ_OK;
    OSQ_REQUIRE(beta && gamma && bias_out, "gamma_split_bias: null pointer");
    hipLaunchKernelGGL(split_bias_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), beta, gamma, bias_out, n);
    return check_launch("gamma_split_bias");
}

extern "C" int osq_gamma_residual(const float* input, const float* hidden, const float* gamma, float* out, int64_t rows,
                                  int64_t cols, osq_stream stream) {
    OSQ_REQUIRE(rows >= 0 && cols >= 0, "gamma_residual: negative size");
    const int64_t n = rows * cols;
    if (n == 0) return OSQ_OK;
    OSQ_REQUIRE(input && hidden && out, "gamma_residual: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = cols % 4 == 0 && aligned16(input) && aligned16(hidden) && aligned16(out) &&
                     (!gamma || aligned16(gamma)) && cols / 4 < (1ll << 31);
    if (vec)
        hipLaunchKernelGGL(gamma_residual_vec_kernel, dim3(grid_for(n / 4, kThreads)), dim3(kThreads), 0, st,
                           reinterpret_cast<const float4*>(input), reinterpret_cast<const float4*>(hidden),
                           reinterpret_cast<const float4*>(gamma), reinterpret_cast<float4*>(out), n / 4,
                           static_cast<int>(cols / 4));
    else
        hipLaunchKernelGGL(gamma_residual_kernel, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, st, input, hidden, gamma,
                           out, n, cols);
    return check_launch("gamma_residual");
}
