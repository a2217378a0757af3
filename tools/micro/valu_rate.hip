// Issue rate of the VALU instructions the MSEFast float64 term is made of, on one MI355X: every SIMD of the chip runs WAVES
// waves that each issue N independent copies of one instruction per loop turn; prints clocks per wave-instruction per SIMD.
// hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ void rate_kernel(double* out, int iters, double seed_d, float seed_f) {
    double a0 = seed_d + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0 = seed_f + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    const double kd = seed_d * 0.5 + 1.0;
    const float kf = seed_f * 0.5f + 1.0f;
    for (int i = 0; i < iters; ++i) {
#define EIGHT(INS, T0, T1, T2, T3, T4, T5, T6, T7) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
        : "+v"(T0), "+v"(T1), "+v"(T2), "+v"(T3), "+v"(T4), "+v"(T5), "+v"(T6), "+v"(T7) : "v"(kd), "v"(kf))
        if (OP == 0) {
#define I0(n) "v_mul_f64 %" #n ", %" #n ", %8\n"
            EIGHT(I0, a0, a1, a2, a3, a4, a5, a6, a7);
        } else if (OP == 1) {
#define I1(n) "v_add_f64 %" #n ", %" #n ", %8\n"
            EIGHT(I1, a0, a1, a2, a3, a4, a5, a6, a7);
        } else if (OP == 2) {
#define I2(n) "v_fma_f64 %" #n ", %" #n ", %8, %8\n"
            EIGHT(I2, a0, a1, a2, a3, a4, a5, a6, a7);
        } else if (OP == 3) {
#define I3(n) "v_mul_f32 %" #n ", %" #n ", %9\n"
            EIGHT(I3, f0, f1, f2, f3, f4, f5, f6, f7);
        } else if (OP == 4) {
#define I4(n) "v_rndne_f32 %" #n ", %" #n "\n"
            EIGHT(I4, f0, f1, f2, f3, f4, f5, f6, f7);
        } else if (OP == 5) {
#define I5(n) "v_med3_f32 %" #n ", %" #n ", %9, %9\n"
            EIGHT(I5, f0, f1, f2, f3, f4, f5, f6, f7);
        } else if (OP == 6) {           // f32 -> f64: the source is the low word of the destination pair
#define I6(n) "v_cvt_f64_f32 %" #n ", %" #n "\n"
            asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\nv_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
        } else if (OP == 7) {
#define I7(n) "v_rndne_f64 %" #n ", %" #n "\n"
            EIGHT(I7, a0, a1, a2, a3, a4, a5, a6, a7);
        } else if (OP == 8) {
#define I8(n) "v_sub_f32 %" #n ", %" #n ", %9\n"
            EIGHT(I8, f0, f1, f2, f3, f4, f5, f6, f7);
        } else if (OP == 9) {
            asm volatile("v_cmp_ge_f32 vcc, |%0|, %1\nv_cmp_ge_f32 vcc, |%2|, %1\nv_cmp_ge_f32 vcc, |%3|, %1\nv_cmp_ge_f32 vcc, |%4|, %1\n"
                         "v_cmp_ge_f32 vcc, |%5|, %1\nv_cmp_ge_f32 vcc, |%6|, %1\nv_cmp_ge_f32 vcc, |%7|, %1\nv_cmp_ge_f32 vcc, |%8|, %1\n"
                         :: "v"(f0), "v"(kf), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7) : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

template <int OP>
static int run(const char* name, double* out, int cus, int waves_per_simd) {
    const int iters = 20000;
    const int block = 256;                         // one wave per SIMD of a CU
    const int grid = cus * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(block), 0, 0, out, 2000, 1.5, 1.25f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(block), 0, 0, out, iters, 1.5, 1.25f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = static_cast<double>(iters) * 8 * waves_per_simd;
    printf("%-16s %d waves/SIMD: %8.3f ms  -> %6.2f ns per wave-instruction per SIMD (= %5.2f clocks at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    double* out;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * cus * 8));
    for (int w : {1, 4}) {
        run<0>("v_mul_f64", out, cus, w);
        run<1>("v_add_f64", out, cus, w);
        run<2>("v_fma_f64", out, cus, w);
        run<3>("v_mul_f32", out, cus, w);
        run<4>("v_rndne_f32", out, cus, w);
        run<5>("v_med3_f32", out, cus, w);
        run<6>("v_cvt_f64_f32", out, cus, w);
        run<7>("v_rndne_f64", out, cus, w);
        run<8>("v_sub_f32", out, cus, w);
        run<9>("v_cmp_ge_f32", out, cus, w);
    }
    return 0;
}
