// Read-only streaming of 604 MB with the two lane -> address patterns a chunk of the ordered sum can be read with (16 KB chunk =
// 16 blocks x 16 rows x 16 fp32 columns), 8 x 16-byte loads per lane in flight as a ring, nothing but an fp32 add per element:
//   PATTERN 0  "rows": lane (block, quad) reads ITS rows one after the other -- a wave-load is 16 x 64 B, 1 KB apart (half lines)
//   PATTERN 1  "coalesced": a wave-load is 1 KB contiguous (lane l: 16 B at l * 16), 8 full lines
//   PATTERN 2  "slices": lanes 0-15 -> 256 B of block b (4 rows), lanes 16-31 -> block b + 1 ...: 4 x 256 B, 1 KB apart (full lines)
// hipcc -O3 --offload-arch=gfx950 tools/micro/stream_pattern.hip -o tools/micro/stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(256, 4) void stream_kernel(const float* __restrict__ x, long chunks, float* __restrict__ out, int cpw) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long waves = (long)gridDim.x * 4;
    const long w0 = (long)blockIdx.x * 4 + wv;
    float acc = 0.f;
    v4 A[8];
    // step s of a chunk: 8 wave-loads (8 KB); a chunk is 2 steps
    auto addr = [&](long m, int h, int k) -> const v4* {
        const float* base = x + m * 4096;
        if (PATTERN == 0) return reinterpret_cast<const v4*>(base + ((lane >> 2) * 256 + (h * 8 + k) * 16 + (lane & 3) * 4));
        if (PATTERN == 1) return reinterpret_cast<const v4*>(base + ((h * 8 + k) * 256 + lane * 4));
        return reinterpret_cast<const v4*>(base + (((h * 8 + k) & 3) * 4 + (lane >> 4)) * 256 + ((h * 8 + k) >> 2) * 64 + (lane & 15) * 4);
    };
    long m = w0;
    if (m >= chunks) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) { A[k] = *addr(m, 0, k); __builtin_amdgcn_sched_barrier(0); }
    for (;;) {
        long mn = m + waves;
        const bool more = mn < chunks;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc += (A[k].x + A[k].y) + (A[k].z + A[k].w);
                asm volatile("" : "+v"(acc));
                A[k] = h == 0 ? *addr(m, 1, k) : *addr(more ? mn : m, 0, k);
            }
        }
        if (!more) break;
        m = mn;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int PATTERN>
static int run(const char* name, const float* x, long chunks, float* out, int grid) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(stream_kernel<PATTERN>, dim3(grid), dim3(256), 0, 0, x, chunks, out, 0);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream_kernel<PATTERN>, dim3(grid), dim3(256), 0, 0, x, chunks, out, 0);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    printf("%-10s grid %5d: %7.1f us  %6.3f TB/s\n", name, grid, best * 1e3, chunks * 16384.0 / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    const long chunks = 36864;                       // 604 MB
    float *x, *out;
    CHECK(hipMalloc(&x, chunks * 16384));
    CHECK(hipMalloc(&out, 4 * 65536 * 256));
    CHECK(hipMemset(x, 0, chunks * 16384));
    for (int grid : {1024, 2304, 4608, 9216}) {
        run<0>("rows", x, chunks, out, grid);
        run<1>("coalesced", x, chunks, out, grid);
        run<2>("slices", x, chunks, out, grid);
    }
    return 0;
}
