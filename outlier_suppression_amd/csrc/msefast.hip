// MSEFastObserver / AvgMSEFastObserver on gfx950 (MI355X).
// Replaces quant_transformer/quantization/observer.py:412-567: a host-driven
// scipy.optimize.minimize_scalar(method='Bounded') whose every loss evaluation is ~10 eager
// kernels plus a .cpu() sync (and, per-channel, a Python loop over rows: ~15 evaluations per
// row, 134 K rows for RoBERTa-base).
//
// Here the bounded Brent search (Forsythe/Malcolm/Moler fmin; restated and pinned against
// scipy in oracle/brent.py + tests/test_oracle_pinning.py) is a device-side state machine:
//   * per-channel weights: ONE launch; one wave per row, the row lives in registers, the whole
//     search (1-D, or nested 2-D for asymmetric two-sided rows) runs inside the wave;
//   * per-tensor activations: ONE persistent launch per search when the tensor fits the grid's
//     registers (msefast_resident_kernel below); otherwise the search state sits in device
//     memory and each loss evaluation is one streaming launch (4 B/elem, padded tokens skipped)
//     whose last workgroup feeds the loss to the state machine and publishes the next candidate
//     -- launches are enqueued in chunks, a finished search turns the rest of a chunk into no-ops.
// Numerics follow the reference: candidates and qparams in float64 (scipy hands float64,
// observer.py:423-428), scale applied as fp32, zero-point truncated to int, squared error in
// fp32, summed in float64 from the first addition on, the mean rounded to fp32 once (the
// reference's torch mean is an fp32 sum whose order depends on the machine's vector width).
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>
#include <hip/hip_ext.h>
#include "osq_device.h"
#include "aten_order.h"
#include "osq_host.h"

namespace osq {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / OSQ_WAVE;
typedef unsigned int v4u32_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- bounded Brent, ask/tell

struct Brent {
    double a, b, v, w, xf, d, e, fx, fv, fw, xm, tol1, tol2, pending;
    int nfev, done, first;
    int f32_values;      // the objective returns np.float32 (fp32 tensors): scipy subtracts such values in float32

    __device__ double start(double lo, double hi) {
        const double gold = 0.5 * (3.0 - 2.23606797749978969641);   // 0.5*(3 - sqrt(5))
        a = lo; b = hi;
        v = w = xf = a + gold * (b - a);
        d = e = 0.0;
        fx = fv = fw = 0.0;
        nfev = 0; done = 0; first = 1;
        pending = xf;
        return xf;
    }
    __device__ void tolerances() {
        const double sqrt_eps = 1.4832396974191326e-08;               // sqrt(2.2e-16)
        xm = 0.5 * (a + b);
        tol1 = sqrt_eps * fabs(xf) + 1e-5 / 3.0;                      // xatol = 1e-5
        tol2 = 2.0 * tol1;
    }
    __device__ double propose() {
        const double gold = 0.5 * (3.0 - 2.23606797749978969641);
        bool golden = true;
        if (fabs(e) > tol1) {
            golden = false;
            // The function values are fp32 numbers (np.float32 from loss_fx, observer.py:431-432), so scipy's
            // `fx - ffulc` is an fp32 subtraction -- rounded when the two differ by more than a factor of two -- and
            // only its product with the float64 abscissa difference is float64.  (float64 tensors -- the reference's
            // per-tensor observers from their second call on, see Search::f64 -- give np.float64 values: plain float64.)
            double r = (xf - w) * (f32_values ? static_cast<double>(static_cast<float>(fx) - static_cast<float>(fv)) : fx - fv);
            double q = (xf - v) * (f32_values ? static_cast<double>(static_cast<float>(fx) - static_cast<float>(fw)) : fx - fw);
            double p = (xf - v) * q - (xf - w) * r;
            q = 2.0 * (q - r);
            if (q > 0.0) p = -p;
            q = fabs(q);
            r = e;
            e = d;
            if (fabs(p) < fabs(0.5 * q * r) && p > q * (a - xf) && p < q * (b - xf)) {
                d = p / q;
                const double x = xf + d;
                if ((x - a) < tol2 || (b - x) < tol2) {
                    const double sg = (xm > xf ? 1.0 : (xm < xf ? -1.0 : 0.0)) + (xm == xf ? 1.0 : 0.0);
                    d = tol1 * sg;
                }
            } else {
                golden = true;
            }
        }
        if (golden) {
            e = (xf >= xm) ? (a - xf) : (b - xf);
            d = gold * e;
        }
        const double sg = (d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0)) + (d == 0.0 ? 1.0 : 0.0);
        const double ad = fabs(d);
        return xf + sg * (ad > tol1 ? ad : tol1);
    }
    // returns the next abscissa; sets done when converged (the caller must not evaluate again)
    __device__ double tell(double fu) {
        ++nfev;
        const double x = pending;
        if (first) {
            first = 0;
            fx = fv = fw = fu;
        } else {
            if (fu <= fx) {
                if (x >= xf) a = xf; else b = xf;
                v = w; fv = fw;
                w = xf; fw = fx;
                xf = x; fx = fu;
            } else {
                if (x < xf) a = x; else b = x;
                if (fu <= fw || w == xf) {
                    v = w; fv = fw;
                    w = x; fw = fu;
                } else if (fu <= fv || v == xf || v == w) {
                    v = x; fv = fu;
                }
            }
            if (nfev >= 500) { done = 1; return xf; }                 // maxiter
        }
        tolerances();
        if (!(fabs(xf - xm) > (tol2 - 0.5 * (b - a)))) { done = 1; return xf; }
        pending = propose();
        return pending;
    }
};

// ---------------------------------------------------------------- search driver (observer.py:434-494)

enum { SIDE_NO = 0, SIDE_POS = 1, SIDE_NEG = 2 };

struct Search {
    Brent outer, inner;
    double x_min, x_max;        // observed extrema (exact fp32 values)
    double cand_min, cand_max;  // candidate range whose loss is wanted next
    double cur_range, best_min, best_max;
    int quant_min, quant_max, symmetric, side, two_d, phase, done, nfev;
    // MSEFastObserver.forward casts its input to min_val's dtype (observer.py:524 / 549), and after the first
    // per-tensor call min_val is the float64 tensor scipy's result was wrapped in (observer.py:481,494): from the
    // second call on the reference runs the whole search on a float64 copy of x -- float64 fake-quant, float64 mean,
    // np.float64 function values, float64 range x_max - x_min.  f64 = 1 selects that arithmetic.
    int f64;

    __device__ void shift_bounds(double r, double* lo, double* hi) const {
        const double delta = r / static_cast<double>(quant_max - quant_min);
        *lo = delta * quant_min;
        *hi = delta * quant_max;
    }
    __device__ void candidate_1d(double r) {
        cand_min = side == SIDE_POS ? 0.0 : -r;
        cand_max = side == SIDE_NEG ? 0.0 : r;
    }
    __device__ void candidate_2d(double shift) {
        const double lo = 0.0 - shift, hi = cur_range - shift;
        cand_min = lo > x_min ? lo : x_min;        // max(tmp_min - shift, x_min)
        cand_max = hi < x_max ? hi : x_max;        // min(tmp_max - shift, x_max)
    }
    __device__ void begin_inner(double r) {
        cur_range = r;
        double lo, hi;
        shift_bounds(r, &lo, &hi);
        candidate_2d(inner.start(lo, hi));
    }
    __device__ void init(float xmin_f, float xmax_f, int qmin, int qmax, int sym, int side_, int two_d_, int f64_ = 0) {
        x_min = xmin_f; x_max = xmax_f;
        quant_min = qmin; quant_max = qmax; symmetric = sym; side = side_; two_d = two_d_;
        f64 = f64_;
        outer.f32_values = inner.f32_values = f64_ ? 0 : 1;
        phase = 0; done = 0; nfev = 0;
        best_min = x_min; best_max = x_max;
        if (!two_d) {
            const float xr_f = fmaxf(fabsf(xmin_f), xmax_f);          // torch.max(x_min.abs(), x_max), fp32
            const double xr = xr_f;
            const double lo = 0.01 * xr < 0.1 ? 0.01 * xr : 0.1;
            candidate_1d(outer.start(lo, xr));
        } else {
            const float xr_f = xmax_f - xmin_f;                        // fp32 subtraction (observer.py:459) of fp32 extrema
            const double xr = f64_ ? static_cast<double>(xmax_f) - static_cast<double>(xmin_f) : static_cast<double>(xr_f);
            const double lo = 0.01 * xr < 0.1 ? 0.01 * xr : 0.1;
            begin_inner(outer.start(lo, xr));
        }
    }
    __device__ void tell(double f) {       // f: the loss as the reference's objective returns it (an fp32 value unless f64)
        ++nfev;
        if (!two_d) {
            const double r = outer.tell(f);
            if (outer.done) {
                const double rr = outer.xf;
                best_min = side == SIDE_POS ? 0.0 : -rr;
                best_max = side == SIDE_NEG ? 0.0 : rr;
                done = 1;
            } else {
                candidate_1d(r);
            }
            return;
        }
        const double s = inner.tell(f);
        if (!inner.done) { candidate_2d(s); return; }
        if (phase == 0) {
            const double r = outer.tell(inner.fx);                     // golden_asym_range_loss returns result.fun
            if (outer.done) { phase = 1; begin_inner(outer.xf); }
            else begin_inner(r);
        } else {
            const double shift = inner.xf, lo = 0.0 - shift, hi = cur_range - shift;
            best_min = lo > x_min ? lo : x_min;
            best_max = hi < x_max ? hi : x_max;
            done = 1;
        }
    }
};

// observer.py:101-119 in float64, then what loss_fx hands to the fake-quant (observer.py:426-429)
__device__ __forceinline__ void loss_qparams(double mn, double mx, int qmin, int qmax, int sym, float* scale_f, float* zp_f,
                                             double* scale_d = nullptr) {
    const double min_neg = mn < 0.0 ? mn : 0.0, max_pos = mx > 0.0 ? mx : 0.0;
    const double eps = static_cast<double>(1e-8f);
    double scale, zp = 0.0;
    if (sym) {
        const double m = -min_neg > max_pos ? -min_neg : max_pos;
        scale = m / (static_cast<double>(qmax - qmin) / 2.0);
        scale = scale > eps ? scale : eps;
    } else {
        scale = (max_pos - min_neg) / static_cast<double>(qmax - qmin);
        scale = scale > eps ? scale : eps;
        zp = static_cast<double>(qmin) - rint(min_neg / scale);
        zp = zp < qmin ? qmin : (zp > qmax ? qmax : zp);
    }
    *scale_f = static_cast<float>(scale);
    *zp_f = static_cast<float>(static_cast<int>(zp));
    if (scale_d) *scale_d = scale;        // scale.item() applied to a float64 tensor stays float64
}

__device__ __forceinline__ float sq_err(float x, float s, float z, float qmin, float qmax) {
    const float y = dequantize_value(quantize_value(x, s, z, qmin, qmax), s, z);
    const float d = fabsf(y - x);
    return d * d;
}

// the same chain on a float64 copy of x (util_quant.py:11-15 with float64 operands)
__device__ __forceinline__ double sq_err_f64(float xf, double s, double z, double qmin, double qmax) {
    const double x = xf;
    const double u = x / s;
    const double r = rint(u);
    const double x_int = ((r - u) + u) + z;
    double q = x_int;
    q = (x_int < qmin) ? qmin : q;
    q = (x_int > qmax) ? qmax : q;
    const double y = (q - z) * s;
    const double d = fabs(y - x);
    return d * d;
}
// The same with x / s computed as Markstein's sequence for a divisor shared by all elements: y = RN(1 / s) once, then
// q0 = x*y, two rounds of (r = x - q*s exactly, by fma; q += r*y).  The second round's result is the correctly rounded
// quotient -- the same bits as the division -- whenever no intermediate over/underflows and the significand of s is
// not all ones (Markstein 1990; Muller et al., Handbook of Floating-Point Arithmetic, 2nd ed., Theorem 4.9 / section
// 4.7.2).  x holds finite fp32 values (|x| in [1.4e-45, 3.4e38] or 0) and s lies in [1e-8, 1.1e37], so quotients stay
// within [1e-82, 3.4e46]: nothing leaves the normal range.  5 full-rate fma-class instructions instead of
// v_rcp_f64 (quarter rate) + its Newton steps + div_scale / div_fmas / div_fixup.  The caller checks the two conditions.
__device__ __forceinline__ double sq_err_f64_rcp(float xf, double s, double y, double z, double qmin, double qmax) {
    const double x = xf;
    const double q0 = x * y;
    const double q1 = __builtin_fma(__builtin_fma(-q0, s, x), y, q0);
    const double u = __builtin_fma(__builtin_fma(-q1, s, x), y, q1);
    // u is finite here, so (rint(u) - u) + u == rint(u) exactly (the difference is exact by Sterbenz's lemma, or 0), and
    // the clamp never meets a NaN: fmax / fmin give what the two compare-and-selects give
    const double x_int = rint(u) + z;
    const double q = __builtin_fmin(__builtin_fmax(x_int, qmin), qmax);
    const double yv = (q - z) * s;
    const double d = fabs(yv - x);
    return d * d;
}
// LEAN form of the same term (round 5; profiles/r05_mse_rounds_pmc.txt: a round's VALU pipes are busy 62 % of its time, 15
// float64 operations per element at 4 clocks each).  Only the INTEGER LEVEL depends on the quotient: q - z =
// clamp(rint(x / s), qmin - z, qmax - z) for an integer z (exact small-integer arithmetic in the reference's float64 chain
// too).  The level is taken from an fp32 quotient u32 = x * RN32(1 / s), whose relative error is below 2^-23: for |u| <= 513
// (the levels lie within +-512, checked by the caller) it is within 6.2e-5 of the float64 quotient, so rint agrees unless u32
// lies within 1e-4 of a tie (0.5 - 0.4999 > 6.2e-5) -- those elements, one in five thousand, take the exact float64 chain above
// (a 5e-4 guard measured the same: the rounds are not bound by their VALU work); beyond the clamp range
// either rounding saturates to the same level (+-inf included).  The dequantised value, the difference and the square are
// the reference's float64 operations: c * s, - x, squared ((-d)^2 = d^2 exactly).  6 fp32 + 6 float64 operations instead of 15.
__device__ __forceinline__ double sq_err_f64_lean(float xf, double s, double y, float y32, float lo32, float hi32, double z, double qmin,
                                                  double qmax) {
    const float u = xf * y32;
    const float r = rintf(u);
    if (fabsf(u - r) >= 0.4999f) return sq_err_f64_rcp(xf, s, y, z, qmin, qmax);      // false for NaN (u = +-inf): saturates below
    const float c = __builtin_amdgcn_fmed3f(r, lo32, hi32);       // the clamp as ONE instruction (lo32 <= hi32, r is not NaN here: fminf(fmaxf()) is three, two of them canonicalising moves)
    const double d = static_cast<double>(c) * s - static_cast<double>(xf);
    return d * d;
}
// uniform: may the lean form stand in for sq_err_f64_rcp?  (z an integer, levels within +-512 of it; the caller has checked
// rcp_division_exact: finite data, a scale in [1e-9, 1e38])
__device__ __forceinline__ bool lean_level_exact(float z, float qmin, float qmax) {
    return z == rintf(z) && fabsf(qmin - z) <= 512.0f && fabsf(qmax - z) <= 512.0f && qmin <= qmax;
}
__device__ __forceinline__ double sq_err4_f64_rcp(const float4& a, double s, double y, double z, double qmin, double qmax) {
    return (sq_err_f64_rcp(a.x, s, y, z, qmin, qmax) + sq_err_f64_rcp(a.y, s, y, z, qmin, qmax)) +
           (sq_err_f64_rcp(a.z, s, y, z, qmin, qmax) + sq_err_f64_rcp(a.w, s, y, z, qmin, qmax));
}
__device__ __forceinline__ bool rcp_division_exact(double s, double x_min, double x_max) {
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(s));
    const bool finite = fabs(x_min) <= 3.5e38 && fabs(x_max) <= 3.5e38;        // false for NaN / inf extrema
    return finite && (b & 0xfffffffffffffull) != 0xfffffffffffffull && s >= 1e-9 && s <= 1e38;
}
__device__ __attribute__((noinline)) double sq_err_f64_outofline(float xf, double s, double z, double qmin, double qmax) {
    return sq_err_f64(xf, s, z, qmin, qmax);
}
__device__ __forceinline__ double sq_err4_f64(const float4& a, double s, double z, double qmin, double qmax) {
    return (sq_err_f64(a.x, s, z, qmin, qmax) + sq_err_f64(a.y, s, z, qmin, qmax)) +
           (sq_err_f64(a.z, s, z, qmin, qmax) + sq_err_f64(a.w, s, z, qmin, qmax));
}

// four squared errors, summed in float64 (see msefast_rows_kernel on why every addition is a float64 one)
__device__ __forceinline__ double sq_err4(const float4& a, float s, float z, float qmin, float qmax) {
    return (static_cast<double>(sq_err(a.x, s, z, qmin, qmax)) + static_cast<double>(sq_err(a.y, s, z, qmin, qmax))) +
           (static_cast<double>(sq_err(a.z, s, z, qmin, qmax)) + static_cast<double>(sq_err(a.w, s, z, qmin, qmax)));
}

// ---------------------------------------------------------------- the reference machine's summation order
//
// The reference's loss is torch's CPU `mean` of n fp32 squared errors: sum_out(...).div_(n), and the sum adds in the
// order of ATen's cascade_sum / vectorized_inner_sum (aten/src/ATen/native/cpu/SumKernel.cpp; restated and pinned in
// oracle/aten_sum.py): W = 8 SIMD lanes (the sum kernel is dispatched at AVX2 width also on AVX-512 machines), lane l
// owning elements l, l + 8, ...; per lane four interleaved accumulators, each a 4-level cascade; then the n % 8
// trailing scalars and the 8 lanes, in order, onto a scalar.  For a per-channel ROW (< 32768 elements: below ATen's
// parallel grain) that order is what every x86 host computes, so the per-row kernel adds in exactly that order BY DEFAULT
// (osq_set_tuning("mse_rows_order", 8); fp32 additions, lanes 0..7 of the wave playing the SIMD lanes, squared errors
// staged in LDS): the searches are the reference's own, iterate for iterate, and the ranges equal the reference-generated
// fixtures BIT FOR BIT (tests/test_gpu_parity.py::test_msefast_rows_equal_reference_in_its_summation_order).  Per-TENSOR
// searches keep the order-free sum -- the correctly rounded loss: beyond 32768 elements torch's order depends on the host's
// thread count -- and take the reference's order only in the test mode osq_set_tuning("mse_sum_order", 8).
// (ceil_log2_i, aten_lane_partial, aten_sum_wave, aten_mean_wave: osq_device.h -- the LSQ backward has the same test mode)

// ---------------------------------------------------------------- per-channel: one wave per row

template <int MAXV, bool ATEN = false>   // cached floats per lane (row length <= 64*MAXV); MAXV == 0: re-read the row
__global__ __launch_bounds__(kThreads) void msefast_rows_kernel(const float* __restrict__ w, int64_t rows, int cols,
                                                                int quant_min, int quant_max, int symmetric, int side,
                                                                int two_d, float* __restrict__ best_min,
                                                                float* __restrict__ best_max, int* __restrict__ nfev_out,
                                                                int sum_width = 0) {
    extern __shared__ float sq_stage[];          // ATEN: cols floats per wave
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    const int64_t row = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) / OSQ_WAVE;   // blockDim.x: 256, or 64 for long rows in ATEN mode
    if (row >= rows) return;
    const float* xr = w + row * cols;
    constexpr int R = MAXV > 0 ? MAXV : 1;
    float cache[R];
    float mn = __builtin_inff(), mx = -__builtin_inff();
    if (MAXV > 0) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int j = lane + k * OSQ_WAVE;
            cache[k] = j < cols ? xr[j] : 0.0f;
            if (j < cols) { mn = fminf(mn, cache[k]); mx = fmaxf(mx, cache[k]); }
        }
    } else {
        for (int j = lane; j < cols; j += OSQ_WAVE) { const float v = xr[j]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    Search S;                                    // every lane runs the identical state machine
    S.init(mn, mx, quant_min, quant_max, symmetric, side, two_d);   // observer.py:500-517: the row's own extrema
    const float qmin_f = static_cast<float>(quant_min), qmax_f = static_cast<float>(quant_max);
    while (!S.done) {
        float s, z;
        loss_qparams(S.cand_min, S.cand_max, quant_min, quant_max, symmetric, &s, &z);
        // every squared error (an fp32 value) is added in float64 from the first addition on: the total is then the
        // exact sum to ~1e-16 whatever the order, and its fp32-rounded mean is THE correctly rounded loss -- the same
        // number oracle/observer_oracle.py::mse_loss computes, so the search is iterate-for-iterate the oracle's
        if (ATEN) {
            float* sq = sq_stage + (threadIdx.x / OSQ_WAVE) * cols;
            if (MAXV > 0) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int j = lane + k * OSQ_WAVE;
                    if (j < cols) sq[j] = sq_err(cache[k], s, z, qmin_f, qmax_f);
                }
            } else {
                for (int j = lane; j < cols; j += OSQ_WAVE) sq[j] = sq_err(xr[j], s, z, qmin_f, qmax_f);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // rows shorter than one SIMD vector: ATen's scalar_inner_sum (four interleaved scalar accumulators)
            S.tell(static_cast<double>(cols >= sum_width ? aten_mean_wave(sq, cols, sum_width) : aten_sum_short(sq, cols) / static_cast<float>(cols)));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            continue;
        }
        double part = 0.0;
        if (MAXV > 0) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int j = lane + k * OSQ_WAVE;
                if (j < cols) part += static_cast<double>(sq_err(cache[k], s, z, qmin_f, qmax_f));
            }
        } else {
            for (int j = lane; j < cols; j += OSQ_WAVE) part += static_cast<double>(sq_err(xr[j], s, z, qmin_f, qmax_f));
        }
        const double tot = wave_sum(part);
        S.tell(static_cast<double>(static_cast<float>(tot / static_cast<double>(cols))));
    }
    if (lane == 0) {
        best_min[row] = static_cast<float>(S.best_min);    // assignment into fp32 tensors (observer.py:504,516)
        best_max[row] = static_cast<float>(S.best_max);
        if (nfev_out) nfev_out[row] = S.nfev;
    }
}

// ---------------------------------------------------------------- per-tensor: state in device memory

// LOSS MEMO (round 6).  loss_fx (observer.py:423-432) is a pure function of the tensor and of the fake-quant parameters it is
// handed -- scale.item() and int(zero_point.item()) -- and scipy's bounded search asks for the same pair again and again: the
// inner search of the nested form (observer.py:434-446) moves the shift of a FIXED range, so the scale stays and the integer
// zero point changes once per quantisation step; after ~8 evaluations the bracket is narrower than a step and the remaining
// ~15 (xatol = 1e-5) repeat one pair, and the final inner search (observer.py:469-475) repeats the best range's from its first
// evaluation to its last.  On two-sided data ~100 of a search's ~450 evaluations are distinct.  Every per-tensor search keeps
// the (scale word, zero point) -> loss pairs of the evaluations it has streamed; the step that follows an evaluation keeps
// advancing the state machine for as long as the next candidate's pair is in the table.  Same function value, bit for bit
// (it IS the earlier evaluation's), same iterates, same nfev -- the pass over the tensor is what is skipped.
// osq_set_tuning("mse_memo", 0): every evaluation streams (tests: equal results; A/B).
constexpr int kMemoCap = 512;                    // pairs a search keeps (more distinct pairs than that: the rest stream as before)
struct TensorSearch {
    Search S;
    float scale, zp;          // fake-quant parameters of the pending candidate
    int evals_launched;
    int pad;
    double scale_d;           // the same scale before its fp32 rounding (float64 arithmetic, Search::f64)
    int memo_count, memo_hits;
    unsigned long long memo_scale[kMemoCap];     // the scale as the evaluation uses it: the float64 word (Search::f64) or the fp32 one
    double memo_loss[kMemoCap];                  // what the evaluation handed to tell()
    unsigned int memo_zp[kMemoCap];
};
static_assert(sizeof(Search) % 8 == 0 && sizeof(Search) <= 1024, "the staged copy of the state: 8-byte words, at most 1 KiB");
constexpr int kMemoLdsBytes = 1024 + kMemoCap * 20;          // the state's copy + the table, staged by the advancing wave

__global__ void msefast_tensor_init_kernel(TensorSearch* __restrict__ ts, const float* __restrict__ cur_minmax,
                                           int quant_min, int quant_max, int symmetric, int side, int two_d, int f64) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ts->S.init(cur_minmax[0], cur_minmax[1], quant_min, quant_max, symmetric, side, two_d, f64);
    loss_qparams(ts->S.cand_min, ts->S.cand_max, quant_min, quant_max, symmetric, &ts->scale, &ts->zp, &ts->scale_d);
    ts->evals_launched = 0;
    ts->memo_count = 0;
    ts->memo_hits = 0;
}

// The step after a streamed evaluation: loss -> state machine -> next candidate, repeated while the memo knows the candidate.
// Called by ONE WHOLE WAVE (64 converged lanes) of the workgroup that finished the evaluation; `loss` (what the reference's
// objective returns: an fp32 value unless Search::f64) is lane 0's.  lds: kMemoLdsBytes the wave may overwrite.  The state and
// the table are staged in LDS for the walk -- a hit then costs the serial Brent step on LDS operands (~0.3 us) instead of a
// memory round trip per field -- and written back once.
__device__ __forceinline__ void tensor_search_advance(TensorSearch* ts, double loss, unsigned char* lds, const bool memo_on) {
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    constexpr int kWords = static_cast<int>(sizeof(Search) / 8);
    Search* const S = reinterpret_cast<Search*>(lds);
    unsigned long long* const mk = reinterpret_cast<unsigned long long*>(lds + 1024);
    double* const ml = reinterpret_cast<double*>(mk + kMemoCap);
    unsigned int* const mz = reinterpret_cast<unsigned int*>(ml + kMemoCap);
    int count = memo_on ? __builtin_amdgcn_readfirstlane(ts->memo_count) : 0;
    if (count > kMemoCap) count = kMemoCap;
    float sc = ts->scale, zp = ts->zp;                     // the pair the streamed evaluation used
    double scd = ts->scale_d;
    {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&ts->S);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(S);
        for (int i = lane; i < kWords; i += OSQ_WAVE) dst[i] = src[i];
        for (int i = lane; i < count; i += OSQ_WAVE) { mk[i] = ts->memo_scale[i]; ml[i] = ts->memo_loss[i]; mz[i] = ts->memo_zp[i]; }
    }
    cascade_wave_sync();                                      // lane 0 reads what the other lanes staged
    const bool f64 = __builtin_amdgcn_readfirstlane(S->f64) != 0;
    auto key_of = [&](unsigned long long& ks, unsigned int& kz) {
        const unsigned long long w = f64 ? static_cast<unsigned long long>(__double_as_longlong(scd)) : static_cast<unsigned long long>(__float_as_uint(sc));
        ks = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(w >> 32))) << 32) |
             static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(w & 0xffffffffull)));
        kz = static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(__float_as_uint(zp))));
    };
    unsigned long long ks;
    unsigned int kz;
    key_of(ks, kz);
    int n = count, hits = 0, done = 0;
    const bool inserted = memo_on && n < kMemoCap;
    if (inserted) {
        if (lane == 0) { mk[n] = ks; mz[n] = kz; ml[n] = loss; }
        ++n;
    }
    for (;;) {
        cascade_wave_sync();
        if (lane == 0) {
            S->tell(loss);
            if (!S->done) loss_qparams(S->cand_min, S->cand_max, S->quant_min, S->quant_max, S->symmetric, &sc, &zp, &scd);
        }
        cascade_wave_sync();
        done = __builtin_amdgcn_readfirstlane(S->done);
        if (done || !memo_on || hits >= 65536) break;
        key_of(ks, kz);
        int found = -1;
        for (int i = lane; i < n; i += OSQ_WAVE)
            if (mk[i] == ks && mz[i] == kz) found = i;
        const unsigned long long hit = __ballot(found >= 0);
        if (!hit) break;
        const int idx = __shfl(found, __ffsll(static_cast<long long>(hit)) - 1);
        loss = ml[idx];
        ++hits;
    }
    cascade_wave_sync();
    {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(&ts->S);
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(S);
        for (int i = lane; i < kWords; i += OSQ_WAVE) dst[i] = src[i];
    }
    if (lane == 0) {
        if (!done) { ts->scale = sc; ts->zp = zp; ts->scale_d = scd; }
        if (inserted) {
            ts->memo_scale[count] = mk[count]; ts->memo_loss[count] = ml[count]; ts->memo_zp[count] = mz[count];
            ts->memo_count = n;
        }
        if (hits) ts->memo_hits += hits;
    }
}

// ---- test mode osq_set_tuning("mse_sum_order", 64): the loss summed as a double-double (error ~1e-32 relative), so that
// its rounding to float64 does not depend on the order of the additions.  In float64 arithmetic (a per-tensor observer's
// second call on, observer.py:524,549) a plain float64 sum carries its order in its last bits, and near the minimum of a
// staircase loss that is enough to send Brent down another path: with this mode kernel and oracle (whose mean is then
// math.fsum, exactly rounded) agree bit for bit on float64 batches too -- what is left against the reference is the
// rounding noise of ITS sum.  One launch per evaluation only; production sums in plain float64.
struct DD {
    double hi, lo;
};
__device__ __forceinline__ void dd_add(DD& a, double x) {
    const double s = a.hi + x, bb = s - a.hi;
    const double e = (a.hi - (s - bb)) + (x - bb);             // two-sum: a.hi + x = s + e exactly
    a.hi = s;
    a.lo += e;
}
__device__ __forceinline__ DD dd_join(const DD& a, const DD& b) {
    const double s = a.hi + b.hi, bb = s - a.hi;
    const double e = (a.hi - (s - bb)) + (b.hi - bb);
    const double lo = (a.lo + b.lo) + e;
    const double hi = s + lo;
    return DD{hi, lo - (hi - s)};
}
__device__ __forceinline__ DD dd_wave_sum(DD v) {
#pragma unroll
    for (int off = OSQ_WAVE / 2; off > 0; off >>= 1) {
        const DD o{__shfl_xor(v.hi, off), __shfl_xor(v.lo, off)};
        v = dd_join(v, o);
    }
    return v;
}
constexpr int kDdLoOffset = kMaxBlocks + 8;                  // low words of the published partials (behind the count of the token form)

__device__ __forceinline__ void block_sum_publish_finish_exact(DD part, double* partials, unsigned int* counters,
                                                               TensorSearch* ts, double count, const bool memo_on) {
    __shared__ double sh_hi[kWavesPerBlock], sh_lo[kWavesPerBlock];
    __shared__ double memo_lds[kMemoLdsBytes / 8];
    part = dd_wave_sum(part);
    const int lane = threadIdx.x & (OSQ_WAVE - 1), wv = threadIdx.x / OSQ_WAVE;
    if (lane == 0) { sh_hi[wv] = part.hi; sh_lo[wv] = part.lo; }
    __syncthreads();
    if (threadIdx.x == 0) {
        DD a{0.0, 0.0};
        for (int k = 0; k < kWavesPerBlock; ++k) a = dd_join(a, DD{sh_hi[k], sh_lo[k]});
        publish_f64(&partials[blockIdx.x], a.hi);
        publish_f64(&partials[kDdLoOffset + blockIdx.x], a.lo);
    }
    if (grid_last_block(counters, gridDim.x)) {
        constexpr int kPer = kMaxBlocks / kThreads;
        DD a{0.0, 0.0};
        for (int j = 0; j < kPer; ++j) {
            const unsigned int k = threadIdx.x + j * kThreads;
            if (k < gridDim.x) a = dd_join(a, DD{consume_f64(&partials[k]), consume_f64(&partials[kDdLoOffset + k])});
        }
        a = dd_wave_sum(a);
        __syncthreads();
        if (lane == 0) { sh_hi[wv] = a.hi; sh_lo[wv] = a.lo; }
        __syncthreads();
        if (wv == 0) {                                  // the first wave: lane 0 folds, the wave advances the search (memo walk)
            DD tot{0.0, 0.0};
            if (lane == 0)
                for (int k = 0; k < kWavesPerBlock; ++k) tot = dd_join(tot, DD{sh_hi[k], sh_lo[k]});
            const double mean = (tot.hi + tot.lo) / count;
            tensor_search_advance(ts, ts->S.f64 ? mean : static_cast<double>(static_cast<float>(mean)), reinterpret_cast<unsigned char*>(memo_lds), memo_on);
            if (lane == 0) grid_reset(counters, gridDim.x);
        }
    }
}

__device__ __forceinline__ void block_sum_publish_finish(double part, double* partials, unsigned int* counters,
                                                         TensorSearch* ts, double count, const bool memo_on) {
    __shared__ double sh[kWavesPerBlock];
    __shared__ double memo_lds[kMemoLdsBytes / 8];
    part = wave_sum(part);
    const int lane = threadIdx.x & (OSQ_WAVE - 1), wv = threadIdx.x / OSQ_WAVE;
    if (lane == 0) sh[wv] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int k = 0; k < kWavesPerBlock; ++k) a += sh[k];
        publish_f64(&partials[blockIdx.x], a);
    }
    if (grid_last_block(counters, gridDim.x)) {
        // all loads before any use (consumed one by one, the ordered agent-scope loads cost a memory round
        // trip per iteration); same lane-strided summation order as a plain loop
        constexpr int kPer = kMaxBlocks / kThreads;
        double pa[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const unsigned int k = threadIdx.x + j * kThreads;
            pa[j] = consume_f64(&partials[k < gridDim.x ? k : gridDim.x - 1]);
        }
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) a += (threadIdx.x + j * kThreads < gridDim.x) ? pa[j] : 0.0;
        a = wave_sum(a);
        __syncthreads();
        if (lane == 0) sh[wv] = a;
        __syncthreads();
        if (wv == 0) {
            double tot = 0.0;
            if (lane == 0)
                for (int k = 0; k < kWavesPerBlock; ++k) tot += sh[k];
            const double mean = tot / count;
            tensor_search_advance(ts, ts->S.f64 ? mean : static_cast<double>(static_cast<float>(mean)), reinterpret_cast<unsigned char*>(memo_lds), memo_on);
            if (lane == 0) grid_reset(counters, gridDim.x);
        }
    }
}

__global__ __launch_bounds__(kThreads) void msefast_flat_loss_kernel(const float4* __restrict__ x, int64_t n4,
                                                                     const float* __restrict__ xt, int tail, int64_t n,
                                                                     TensorSearch* __restrict__ ts,
                                                                     double* __restrict__ partials,
                                                                     unsigned int* __restrict__ counters, int exact) {
    if (ts->S.done) return;                       // uniform: state only changes between launches
    const float s = ts->scale, z = ts->zp;
    const double sd = ts->scale_d;
    const bool f64 = ts->S.f64 != 0;
    const float qmin = static_cast<float>(ts->S.quant_min), qmax = static_cast<float>(ts->S.quant_max);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
    if (exact & 1) {                              // test mode: every squared error joins a double-double on its own
        DD dd{0.0, 0.0};
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4; i += stride) {
            const float4 a = x[i];
            const float e[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) dd_add(dd, f64 ? sq_err_f64(e[k], sd, z, qmin, qmax) : static_cast<double>(sq_err(e[k], s, z, qmin, qmax)));
        }
        if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < tail)
            dd_add(dd, f64 ? sq_err_f64(xt[threadIdx.x], sd, z, qmin, qmax) : static_cast<double>(sq_err(xt[threadIdx.x], s, z, qmin, qmax)));
        block_sum_publish_finish_exact(dd, partials, counters, ts, static_cast<double>(n), (exact & 2) != 0);
        return;
    }
    double acc = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4; i += 2 * stride) {
        const float4 a = x[i];
        const bool two = (i + stride) < n4;
        const float4 b = two ? x[i + stride] : a;
        if (f64) {
            acc += sq_err4_f64(a, sd, z, qmin, qmax);
            if (two) acc += sq_err4_f64(b, sd, z, qmin, qmax);
        } else {
            acc += sq_err4(a, s, z, qmin, qmax);
            if (two) acc += sq_err4(b, s, z, qmin, qmax);
        }
    }
    if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < tail)
        acc += f64 ? sq_err_f64(xt[threadIdx.x], sd, z, qmin, qmax) : static_cast<double>(sq_err(xt[threadIdx.x], s, z, qmin, qmax));
    block_sum_publish_finish(acc, partials, counters, ts, static_cast<double>(n), (exact & 2) != 0);
}

// masked / strided activations: one wave per token (valid tokens only), any strides
__global__ __launch_bounds__(kThreads) void msefast_token_loss_kernel(const float* __restrict__ x, osq_token_view v,
                                                                      const int64_t* __restrict__ lengths, int vec,
                                                                      TensorSearch* __restrict__ ts,
                                                                      double* __restrict__ partials,
                                                                      unsigned int* __restrict__ counters,
                                                                      const double* __restrict__ valid_count, int exact) {
    if (ts->S.done) return;
    const float s = ts->scale, z = ts->zp;
    const double sd = ts->scale_d;
    const bool f64 = ts->S.f64 != 0;
    const float qmin = static_cast<float>(ts->S.quant_min), qmax = static_cast<float>(ts->S.quant_max);
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    const int64_t ntok = v.batch * v.tokens;
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) / OSQ_WAVE;
    const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
    const int64_t F = v.feat_outer * v.feat_inner;
    if (exact & 1) {                              // test mode: every squared error joins a double-double on its own
        DD dd{0.0, 0.0};
        for (int64_t tok = wave0; tok < ntok; tok += nwaves) {
            const int64_t b = tok / v.tokens, t = tok - b * v.tokens;
            if (lengths && t >= lengths[b]) continue;
            const float* base = x + b * v.stride_batch + t * v.stride_token;
            for (int64_t j = lane; j < F; j += OSQ_WAVE) {
                const int64_t o = j / v.feat_inner, i = j - o * v.feat_inner;
                const float xv = base[o * v.stride_outer + i * v.stride_inner];
                dd_add(dd, f64 ? sq_err_f64(xv, sd, z, qmin, qmax) : static_cast<double>(sq_err(xv, s, z, qmin, qmax)));
            }
        }
        block_sum_publish_finish_exact(dd, partials, counters, ts, valid_count[0], (exact & 2) != 0);
        return;
    }
    double acc = 0.0;
    for (int64_t tok = wave0; tok < ntok; tok += nwaves) {
        const int64_t b = tok / v.tokens, t = tok - b * v.tokens;
        if (lengths && t >= lengths[b]) continue;
        const float* base = x + b * v.stride_batch + t * v.stride_token;
        double p = 0.0;
        if (vec) {           // feat_inner contiguous, 16-byte aligned segments
            const int inner4 = static_cast<int>(v.feat_inner / 4);
            const int64_t F4 = v.feat_outer * inner4;
            for (int64_t j = lane; j < F4; j += OSQ_WAVE) {
                const int64_t o = j / inner4, i = j - o * inner4;
                const float4 a = reinterpret_cast<const float4*>(base + o * v.stride_outer)[i];
                p += f64 ? sq_err4_f64(a, sd, z, qmin, qmax) : sq_err4(a, s, z, qmin, qmax);
            }
        } else {
            for (int64_t j = lane; j < F; j += OSQ_WAVE) {
                const int64_t o = j / v.feat_inner, i = j - o * v.feat_inner;
                const float xv = base[o * v.stride_outer + i * v.stride_inner];
                p += f64 ? sq_err_f64(xv, sd, z, qmin, qmax) : static_cast<double>(sq_err(xv, s, z, qmin, qmax));
            }
        }
        acc += p;
    }
    block_sum_publish_finish(acc, partials, counters, ts, valid_count[0], (exact & 2) != 0);
}

// osq_set_tuning("mse_sum_order", 8 | 16) for the PER-TENSOR searches -- the strict switch of the package
// (outlier_suppression_amd.set_strict): the squared errors of one evaluation are added the way ATen's CPU kernel adds a
// contiguous vector on one thread (aten_order.h: W lanes for fp32, W / 2 for float64 -- 256-bit vectors), in the order
// remove_padding / flatten lays the elements out (observer.py:72-84), the sum is divided in that type and the search
// advances.  Any length: every workgroup adds level-1 chunks of the cascade (8192 fp32 elements), the workgroup that
// arrives last adds the upper levels and plays scipy's step -- one launch per evaluation.  The searches are then the
// reference's own on a one-thread host, and the ranges of every call equal tests/golden/msefast*.npz BIT FOR BIT, at
// BERT-base site sizes included (tests/test_gpu_parity.py::test_msefast_*_equals_reference_in_its_summation_order,
// ::test_msefast_site_size_equals_reference_in_its_summation_order).  A masked / strided site is first gathered into
// the flat layout the reference's remove_padding builds (gather_valid_tokens_kernel: once per search, not per evaluation).
#ifdef OSQ_MSE_DBG
// development build: thread 0 of every workgroup of a round adds its phase durations (100 MHz ticks) here; osq_mse_dbg_read
__device__ unsigned long long g_mse_phase[8];
#define OSQ_MSE_STAMP(var) long long var = 0; do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); var = wall_clock64(); } while (0)
#define OSQ_MSE_PHASE(slot, d) do { if (threadIdx.x == 0) atomicAdd(&g_mse_phase[slot], static_cast<unsigned long long>(d)); } while (0)
#else
#define OSQ_MSE_STAMP(var) do { } while (0)
#define OSQ_MSE_PHASE(slot, d) do { } while (0)
#endif
constexpr int kOrdThreads = 512;
static_assert(kMemoLdsBytes <= 16 * 1024, "the memo walk of a strict evaluation's finisher reuses its 16 KiB of LDS");
constexpr int kOrdLdsBytes = 16 * 1024;                       // stage 1: S * NC values (<= 32 x 64 x 4 B, 32 x 32 x 8 B); stage 2: columns + a tile of level-2 units
// one loss evaluation of one search: workgroup `bid` of the `nblk` that serve it; `counters` are the search's own
template <int THREADS>
__device__ __forceinline__ void ordered_evaluation(const float* __restrict__ x_generic, const int64_t n, TensorSearch* __restrict__ ts,
                                                   void* __restrict__ scratch, unsigned int* __restrict__ counters, const int W_and_flags,
                                                   const unsigned int bid, const unsigned int nblk, double* lds_raw) {
    // The rounds kernel takes x from a TABLE in memory: to the compiler a generic pointer, whose loads are flat_load_dword (they
    // tick lgkmcnt as well as vmcnt, so every LDS wait also waits for data loads in flight).  x is device memory by the entry
    // points' contract: say so, and the loads are global_load_dword.
    const GlobalF32 x = as_global(x_generic);
    OSQ_MSE_STAMP(t_entry);
    const int W = W_and_flags & 0xff;                             // 8 | 16; bit 8: the lean float64 term (osq_set_tuning("mse_lean"))
    const bool lean_ok = (W_and_flags >> 8) & 1;
    const bool memo_on = (W_and_flags >> 12) & 1;                  // bit 12: the loss memo (tensor_search_advance); 9-10: probe modes of -DOSQ_MSE_DBG builds
    // the state's fields travel together with its `done` flag: one round trip, not two, before the first data load
    const float s = ts->scale, z = ts->zp;
    const double sd = ts->scale_d;
    const bool f64 = ts->S.f64 != 0;
    const float qmin = static_cast<float>(ts->S.quant_min), qmax = static_cast<float>(ts->S.quant_max);
    const double x_min = ts->S.x_min, x_max = ts->S.x_max;
    if (ts->S.done) return;                                       // uniform: a converged search costs its workgroups one look
    OSQ_MSE_STAMP(t_state);
    if (f64) {
        const CascadeGeom g = cascade_geom(n, W / 2);
        // the float64 chain is VALU-bound on its division: the exact reciprocal sequence of the resident search (same bits,
        // sq_err_f64_rcp) whenever its two conditions hold -- a uniform branch, the division out of line
        const double rcp = 1.0 / sd;
        // uniform facts, and the compiler is told so: they come from vector loads of the search's state, and as per-lane values
        // every term sat behind three levels of exec masks (~12 scalar instructions per element next to ~14 vector ones)
        const bool fast = __builtin_amdgcn_readfirstlane(rcp_division_exact(sd, x_min, x_max) ? 1 : 0) != 0;
        const bool lean = __builtin_amdgcn_readfirstlane((fast && lean_ok && lean_level_exact(z, qmin, qmax)) ? 1 : 0) != 0;
        const float rcp32 = static_cast<float>(rcp), lo32 = qmin - z, hi32 = qmax - z;
        auto term = [=](int64_t e, double (&t)[1]) {
            if (lean) t[0] = sq_err_f64_lean(x[e], sd, rcp, rcp32, lo32, hi32, z, qmin, qmax);
            else if (fast) t[0] = sq_err_f64_rcp(x[e], sd, rcp, z, qmin, qmax);
            else t[0] = sq_err_f64_outofline(x[e], sd, z, qmin, qmax);
        };
        double* part = static_cast<double*>(scratch);
        double* lds = lds_raw;
        if ((g.S * g.NC) <= THREADS && g.chunks > 0) {
            // full chunks: loads of the next chunk under the arithmetic of this one (aten_order.h); the open unit below
#ifdef OSQ_MSE_DBG
            const bool dbg_noload = (W_and_flags >> 9) & 1, dbg_trivial = (W_and_flags >> 10) & 1;
            auto load = [=](int64_t e) { return dbg_noload ? __int_as_float(0x3f000000 + static_cast<int>(e & 0xfffff)) : x[e]; };
#else
            constexpr bool dbg_trivial = false;
            auto load = [=](int64_t e) { return x[e]; };
#endif
            if (lean && g.NC == 16) {
                // the common case by far (a finite tensor, an integer zero point, the reference machine's W = 8): its own loop --
                // the lean term alone, log2(NC) a constant (one address per thread and step, 16 immediate offsets)
                // four of a thread's rows side by side; a four with an element within the tie guard (one in ~1200) is redone in the
                // exact chain -- ONE branch per four elements (a diamond per element cost ~6 scalar instructions each)
                auto eval = [=](const float (&xs)[4], double (&t)[4]) {
                    if (dbg_trivial) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) t[i] = static_cast<double>(xs[i]);
                        return;
                    }
                    float r[4];
                    bool tie[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float u = xs[i] * rcp32;
                        r[i] = rintf(u);
                        tie[i] = fabsf(u - r[i]) >= 0.4999f;                  // false for NaN (u = +-inf): saturates below
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float c = __builtin_amdgcn_fmed3f(r[i], lo32, hi32);
                        const double d = static_cast<double>(c) * sd - static_cast<double>(xs[i]);
                        t[i] = d * d;
                    }
                    if (tie[0] | tie[1] | tie[2] | tie[3]) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const double exact = sq_err_f64_rcp(xs[i], sd, rcp, z, qmin, qmax);
                            t[i] = tie[i] ? exact : t[i];
                        }
                    }
                };
                static_assert(cascade_lds_fits<1, 4, THREADS>(kOrdLdsBytes / 8) && cascade_lds_fits<1, 4, THREADS>(kOrdLdsBytes / 4), "kOrdLdsBytes: two tiles + one group's block sums");
                if (g.P == 4) cascade_chunks_pipelined<double, 1, 4, THREADS, float, decltype(load), decltype(eval), 4, 4>(g, part, lds, load, eval, bid, nblk, kOrdLdsBytes / 8);
                else cascade_chunks_pipelined<double, 1, 5, THREADS, float, decltype(load), decltype(eval), 4, 4>(g, part, lds, load, eval, bid, nblk, kOrdLdsBytes / 8);
            } else {
                auto eval = [=](float xf, int64_t, double (&t)[1]) {
                    if (lean) t[0] = sq_err_f64_lean(xf, sd, rcp, rcp32, lo32, hi32, z, qmin, qmax);
                    else t[0] = fast ? sq_err_f64_rcp(xf, sd, rcp, z, qmin, qmax) : sq_err_f64_outofline(xf, sd, z, qmin, qmax);
                };
                if (g.P == 4) cascade_chunks_pipelined<double, 1, 4, THREADS, float>(g, part, lds, load, eval, bid, nblk, kOrdLdsBytes / 8);
                else cascade_chunks_pipelined<double, 1, 5, THREADS, float>(g, part, lds, load, eval, bid, nblk, kOrdLdsBytes / 8);
            }
            OSQ_MSE_STAMP(t_groups);
            if (bid == nblk - 1) cascade_units<double, 1, THREADS>(g, part, lds, term, 0u, 1u, g.chunks);     // the open unit: the workgroup with the fewest chunks
            OSQ_MSE_PHASE(0, t_state - t_entry); OSQ_MSE_PHASE(1, t_groups - t_state); OSQ_MSE_PHASE(4, 1);
        } else {
            cascade_units<double, 1, THREADS>(g, part, lds, term, bid, nblk);
        }
        OSQ_MSE_STAMP(t_before_ticket);
        const bool last_wg = grid_last_block(counters, nblk, bid);
        OSQ_MSE_STAMP(t_after_ticket);
        OSQ_MSE_PHASE(2, t_after_ticket - t_before_ticket);
        if (last_wg) {
            double sum[1] = {0.0};
            cascade_finish<double, 1, THREADS>(g, part, lds, kOrdLdsBytes / 8, term, sum);
            if (threadIdx.x < OSQ_WAVE) {                         // thread 0 holds the sum; its wave walks the memo (the LDS is free now)
                tensor_search_advance(ts, sum[0] / static_cast<double>(n), reinterpret_cast<unsigned char*>(lds_raw), memo_on);
                if (threadIdx.x == 0) grid_reset(counters, nblk);
            }
            OSQ_MSE_STAMP(t_finished);
            OSQ_MSE_PHASE(3, t_finished - t_after_ticket); OSQ_MSE_PHASE(5, 1);
        }
    } else {
        const CascadeGeom g = cascade_geom(n, W);
        auto term = [=](int64_t e, float (&t)[1]) { t[0] = sq_err(x[e], s, z, qmin, qmax); };
        float* part = static_cast<float*>(scratch);
        float* lds = reinterpret_cast<float*>(lds_raw);
        if ((g.S * g.NC) <= THREADS && g.chunks > 0) {
            auto load = [=](int64_t e) { return x[e]; };
            auto eval = [=](float xf, int64_t, float (&t)[1]) { t[0] = sq_err(xf, s, z, qmin, qmax); };
            if (g.NC == 32 && g.P == 4)                         // the reference machine's W = 8 (S * NC <= 512 leaves P = 4 only): log2(NC) a constant, as above
                cascade_chunks_pipelined<float, 1, 4, THREADS, float, decltype(load), decltype(eval), 5>(g, part, lds, load, eval, bid, nblk, kOrdLdsBytes / 4);
            else if (g.P == 4) cascade_chunks_pipelined<float, 1, 4, THREADS, float>(g, part, lds, load, eval, bid, nblk, kOrdLdsBytes / 4);
            else cascade_chunks_pipelined<float, 1, 5, THREADS, float>(g, part, lds, load, eval, bid, nblk, kOrdLdsBytes / 4);
            if (bid == nblk - 1) cascade_units<float, 1, THREADS>(g, part, lds, term, 0u, 1u, g.chunks);
        } else {
            cascade_units<float, 1, THREADS>(g, part, lds, term, bid, nblk);
        }
        if (grid_last_block(counters, nblk, bid)) {
            float sum[1] = {0.0f};
            cascade_finish<float, 1, THREADS>(g, part, lds, kOrdLdsBytes / 4, term, sum);
            if (threadIdx.x < OSQ_WAVE) {
                tensor_search_advance(ts, static_cast<double>(sum[0] / static_cast<float>(n)), reinterpret_cast<unsigned char*>(lds_raw), memo_on);
                if (threadIdx.x == 0) grid_reset(counters, nblk);
            }
        }
    }
}

__global__ __launch_bounds__(kOrdThreads) void msefast_tensor_ordered_kernel(const float* __restrict__ x, int64_t n_host,
                                                                            const int64_t* __restrict__ n_dev,
                                                                            TensorSearch* __restrict__ ts, void* __restrict__ scratch,
                                                                            unsigned int* __restrict__ counters, int W) {
    __shared__ double lds_raw[kOrdLdsBytes / 8];
    ordered_evaluation<kOrdThreads>(x, n_dev ? n_dev[0] : n_host, ts, scratch, counters, W, blockIdx.x, gridDim.x, lds_raw);
}

// The strict form of the searches of a whole forward (the observers of an observer pass are independent): ONE launch per
// ROUND = one loss evaluation of every unfinished search.  A launch-per-evaluation of one site costs ~15 us whatever its
// size (dispatch, the ticket, the serial upper levels of the cascade, scipy's step) on top of ~1.7 us per million
// elements; here that fixed part is paid once per round of up to 128 searches.  Site s owns the workgroups
// block_begin .. block_begin + blocks - 1 of the 1-D grid, its own scratch and its own ticket counters.
struct OrderedSite {
    const float* x;
    const int64_t* n_dev;
    TensorSearch* ts;
    void* scratch;
    unsigned int* counters;
    int64_t n_host;
    unsigned int block_begin, blocks;
    unsigned int pad[2];
};
static_assert(sizeof(OrderedSite) == 64, "OrderedSite is a 64-byte table entry");
constexpr int kOrderedMaxSites = 128;
constexpr size_t kOrderedCounterBytes = (1 + kTicketShards) * kTicketStride * sizeof(unsigned int);

// A workgroup's first loads are a dependent chain -- which site, its table entry, the search's state, only then the data --
// and a round has thousands of short-lived workgroups: the chain is kept at three round trips (block -> site map, entry,
// state; the element count of a masked site is copied into its entry once, by ordered_sites_counts_kernel).
__global__ __launch_bounds__(kOrdThreads, 4) void msefast_tensor_ordered_multi_kernel(const OrderedSite* __restrict__ sites,
                                                                                      const unsigned char* __restrict__ block_site, int n_sites, int W) {
    __shared__ double lds_raw[kOrdLdsBytes / 8];
    const OrderedSite s = sites[block_site[blockIdx.x]];
    ordered_evaluation<kOrdThreads>(s.x, s.n_host, s.ts, s.scratch, s.counters, W, blockIdx.x - s.block_begin, s.blocks, lds_raw);
}

// after the gathers of a group: n_host <- the device-side count of valid elements (same stream, once per group)
__global__ void ordered_sites_counts_kernel(OrderedSite* __restrict__ sites, int n_sites) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sites && sites[i].n_dev) {
        sites[i].n_host = sites[i].n_dev[0];
        sites[i].n_dev = nullptr;
    }
}

__global__ void msefast_done_multi_kernel(const OrderedSite* __restrict__ sites, int n_sites, int* __restrict__ done_out) {
    int all = 1;
    for (int i = threadIdx.x; i < n_sites; i += OSQ_WAVE) all &= sites[i].ts->S.done ? 1 : 0;
    all = __all(all) ? 1 : 0;
    if (threadIdx.x == 0) done_out[0] = all;
}

// remove_padding (observer.py:72-84) as a copy: out[(valid token j) * F + f] for the tokens t < lengths[b], sample by
// sample, features in the view's (outer, inner) order; count_out[0] = elements written.  One wave per token.
__global__ __launch_bounds__(kThreads) void gather_valid_tokens_kernel(const float* __restrict__ x, osq_token_view v,
                                                                       const int64_t* __restrict__ lengths,
                                                                       float* __restrict__ out, int64_t* __restrict__ count_out) {
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    const int64_t tok = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / OSQ_WAVE;
    const int64_t F = v.feat_outer * v.feat_inner;
    auto len_of = [&](int64_t b) -> int64_t {
        int64_t l = lengths ? lengths[b] : v.tokens;
        return l < 0 ? 0 : (l > v.tokens ? v.tokens : l);
    };
    if (tok == 0 && count_out) {                                // wave 0 of workgroup 0: the total
        int64_t tot = 0;
        for (int64_t b = lane; b < v.batch; b += OSQ_WAVE) tot += len_of(b);
        for (int off = OSQ_WAVE / 2; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
        if (lane == 0) count_out[0] = tot * F;
    }
    if (tok >= v.batch * v.tokens) return;
    const int64_t b = tok / v.tokens, t = tok - b * v.tokens;
    if (t >= len_of(b)) return;
    int64_t pre = 0;                                            // valid tokens of the samples before b
    for (int64_t k = lane; k < b; k += OSQ_WAVE) pre += len_of(k);
    for (int off = OSQ_WAVE / 2; off > 0; off >>= 1) pre += __shfl_xor(pre, off);
    const float* base = x + b * v.stride_batch + t * v.stride_token;
    float* dst = out + (pre + t) * F;
    for (int64_t j = lane; j < F; j += OSQ_WAVE) {
        const int64_t o = j / v.feat_inner, i = j - o * v.feat_inner;
        dst[j] = base[o * v.stride_outer + i * v.stride_inner];
    }
}

// number of observed elements = F * sum(min(len, T)) (observer.py:72-84), as a double for the mean
__global__ void msefast_valid_count_kernel(const int64_t* __restrict__ lengths, int64_t B, int64_t T, int64_t F,
                                           double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t tot = 0;
    for (int64_t b = 0; b < B; ++b) {
        int64_t l = lengths ? lengths[b] : T;
        l = l < 0 ? 0 : (l > T ? T : l);
        tot += l;
    }
    out[0] = static_cast<double>(tot * F);
}

// commit a finished per-tensor search: running min/max (observer.py:535-536) or running mean
// (observer.py:559-567) in float64 -- per-tensor results stay float64 in the reference
// (torch.tensor(np.float64), observer.py:481,494) -- then qparams in float64 (observer.py:101-119).
// ref_f64 (nullable: both statistics float64 from the start): int[2] = does the REFERENCE's min_val / max_val hold float64 by
// now.  Its per-tensor results are np.float64 wrapped in tensors -- except where Python hands back a float32 tensor: the
// zeros_like of a one-sided search (observer.py:491-492) and the extremum itself when the nested search's range reaches
// beyond the data (`max(tmp_min - shift, x_min)`, observer.py:479-480).  torch.min / `* cnt + cur` / `/ cnt` then stay
// float32 until a float64 value joins (type promotion of 0-dim tensors), the running mean of such a statistic is fp32
// arithmetic, and calculate_qparams runs in fp32 while BOTH are float32.  Followed here operation by operation; the flags
// are sticky and the host reads [0] to know whether the next batch is cast to float64 (observer.py:524 / 549).
__device__ __forceinline__ double avg_step(double old, bool old_f64, double cur, bool new_f64, int64_t cnt) {
    const double prod = old_f64 ? old * static_cast<double>(cnt)
                                : static_cast<double>(static_cast<float>(old) * static_cast<float>(cnt));
    const double sum = new_f64 ? prod + cur : static_cast<double>(static_cast<float>(prod) + static_cast<float>(cur));
    return new_f64 ? sum / static_cast<double>(cnt + 1)
                   : static_cast<double>(static_cast<float>(sum) / static_cast<float>(cnt + 1));
}

__global__ void msefast_commit_kernel(const TensorSearch* __restrict__ ts, int rule, int64_t cnt,
                                      double* __restrict__ min_val, double* __restrict__ max_val, int quant_min,
                                      int quant_max, int symmetric, float* __restrict__ scale_out,
                                      void* __restrict__ zp_out, int zp_type, int* __restrict__ nfev_out,
                                      int* __restrict__ ref_f64) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Search& S = ts->S;
    const double bmin = S.best_min, bmax = S.best_max;
    bool cur_min_f64 = true, cur_max_f64 = true, old_min_f64 = true, old_max_f64 = true;
    if (ref_f64) {
        old_min_f64 = ref_f64[0] != 0;
        old_max_f64 = ref_f64[1] != 0;
        if (!S.two_d) {
            cur_min_f64 = S.side != SIDE_POS;
            cur_max_f64 = S.side != SIDE_NEG;
        } else if (!S.f64) {
            cur_min_f64 = !(bmin == S.x_min);          // the float32 extremum itself came back
            cur_max_f64 = !(bmax == S.x_max);
        }
    }
    const bool new_min_f64 = old_min_f64 || cur_min_f64, new_max_f64 = old_max_f64 || cur_max_f64;
    double mn = min_val[0], mx = max_val[0];
    if (rule == OSQ_UPDATE_AVERAGE) {
        if (__builtin_isinf(mx)) { mn = bmin; mx = bmax; }
        else {
            mn = avg_step(mn, old_min_f64, bmin, new_min_f64, cnt);
            mx = avg_step(mx, old_max_f64, bmax, new_max_f64, cnt);
        }
    } else {                                   // torch.min / torch.max propagate NaN (a poisoned or NaN-fed search must not vanish)
        mn = (bmin != bmin || mn != mn) ? __builtin_nan("") : (bmin < mn ? bmin : mn);
        mx = (bmax != bmax || mx != mx) ? __builtin_nan("") : (bmax > mx ? bmax : mx);
    }
    min_val[0] = mn;
    max_val[0] = mx;
    if (ref_f64) { ref_f64[0] = new_min_f64 ? 1 : 0; ref_f64[1] = new_max_f64 ? 1 : 0; }
    if (nfev_out) nfev_out[0] = S.nfev;
    if (scale_out) {
        if (!new_min_f64 && !new_max_f64) {           // both statistics float32 in the reference: observer.py:101-119 in fp32
            float s, z;
            qparams_from_range(static_cast<float>(mn), static_cast<float>(mx), quant_min, quant_max, symmetric, &s, &z);
            scale_out[0] = s;
            if (zp_out) store_zp(zp_out, zp_type, 0, z);
            return;
        }
        // the same NaN-propagating min / max / clamp as qparams_f64_kernel below (torch.min, torch.max, torch.clamp)
        const double nan = __builtin_nan("");
        const double min_neg = (mn != mn) ? nan : (mn < 0.0 ? mn : 0.0), max_pos = (mx != mx) ? nan : (mx > 0.0 ? mx : 0.0);
        const double eps = static_cast<double>(1e-8f);
        double scale, zp = 0.0;
        if (symmetric) {
            const double m = (min_neg != min_neg || max_pos != max_pos) ? nan : (-min_neg > max_pos ? -min_neg : max_pos);
            scale = m / (static_cast<double>(quant_max - quant_min) / 2.0);
            scale = (scale != scale) ? nan : (scale > eps ? scale : eps);
        } else {
            scale = (max_pos - min_neg) / static_cast<double>(quant_max - quant_min);
            scale = (scale != scale) ? nan : (scale > eps ? scale : eps);
            zp = static_cast<double>(quant_min) - rint(min_neg / scale);
            zp = (zp != zp) ? nan : (zp < quant_min ? quant_min : (zp > quant_max ? quant_max : zp));
        }
        scale_out[0] = static_cast<float>(scale);
        if (zp_out) store_zp(zp_out, zp_type, 0, static_cast<float>(zp));
    }
}

// calculate_qparams on float64 statistics (a caller handing a per-tensor MSEFast observer's min_val / max_val to
// ObserverBase.calculate_qparams, observer.py:101-119: torch computes in the statistics' dtype and the result is float64)
__global__ void qparams_f64_kernel(const double* __restrict__ mn_p, const double* __restrict__ mx_p, int64_t n, int quant_min,
                                   int quant_max, int symmetric, float* __restrict__ scale_out, void* __restrict__ zp_out,
                                   int zp_type) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double mn = mn_p[i], mx = mx_p[i];
    const double nan = __builtin_nan("");
    const double min_neg = (mn != mn) ? nan : (mn < 0.0 ? mn : 0.0), max_pos = (mx != mx) ? nan : (mx > 0.0 ? mx : 0.0);
    const double eps = static_cast<double>(1e-8f);
    double scale, zp = 0.0;
    if (symmetric) {
        const double m = (min_neg != min_neg || max_pos != max_pos) ? nan : (-min_neg > max_pos ? -min_neg : max_pos);
        scale = m / (static_cast<double>(quant_max - quant_min) / 2.0);
        scale = (scale != scale) ? nan : (scale > eps ? scale : eps);
    } else {
        scale = (max_pos - min_neg) / static_cast<double>(quant_max - quant_min);
        scale = (scale != scale) ? nan : (scale > eps ? scale : eps);
        zp = static_cast<double>(quant_min) - rint(min_neg / scale);
        zp = (zp != zp) ? nan : (zp < quant_min ? quant_min : (zp > quant_max ? quant_max : zp));
    }
    scale_out[i] = static_cast<float>(scale);
    if (zp_out) store_zp(zp_out, zp_type, i, static_cast<float>(zp));
}

__global__ void msefast_done_kernel(const TensorSearch* __restrict__ ts, int* __restrict__ done_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) done_out[0] = ts->S.done;
}

// ---------------------------------------------------------------- per-tensor, resident: the whole search in ONE launch
//
// The launch-per-evaluation form above costs ~12 us per evaluation on a BERT-base site whatever the tensor's size (a
// kernel boundary, the parameters' round trip through memory, the ticket of the last workgroup, and a 6-27 MB tensor
// streamed again from L2 / HBM), and an asymmetric per-tensor search is 300-600 evaluations.  Here the VALID part of
// the tensor is loaded ONCE into the registers of a persistent grid (one 512-thread workgroup per CU; up to 32 float4
// per lane = 67 MB on 256 CUs), and one evaluation is: every thread's squared errors -> one double per workgroup,
// published as two tagged 8-byte granules -> EVERY workgroup collects all partials, adds them in the same order and
// advances its own copy of the state machine (identical arithmetic on identical numbers: no master, no second
// exchange).  One hop through memory per evaluation instead of a kernel boundary.  Padded slots hold 0.0f, whose
// fake-quant is 0 for every candidate (the zero-point is clamped into the quantised range), so the loop carries no mask.
// Tags: epoch + 1 + evaluation; the epoch word lives in the workspace (read by every workgroup at its start, advanced by
// workgroup 0 at its end -- every workgroup has taken part in the last evaluation by then).  Two buffers of granules:
// a workgroup can be at most one evaluation ahead of the slowest reader of its previous partial.
//
// Workgroup shape (round 3): 512 threads = 8 waves = two per SIMD, so a wave may use 256 VGPRs.  Round 2 ran 1024
// threads (128 VGPRs per wave) with 16 float4 per lane: the float64 error chain, the poll and the Brent step did not
// fit beside 64 registers of data and the compiler spilled 102-486 VGPRs to scratch in the instantiations that hold
// 8-16 float4 -- the ones the BERT-base sites use.  Same capacity per CU (512 x 32 float4), same VALU work per SIMD,
// no scratch (tests/test_abi_and_host.py reads vgpr_spill_count of every msefast_resident* kernel from the library).
constexpr int kResThreads = 512;
constexpr int kResWaves = kResThreads / OSQ_WAVE;
constexpr int kResMaxSlots = 32;                     // float4 per lane
constexpr int kResMaxBatch = kResThreads;            // prefix sums of the lengths: one sample per thread
constexpr unsigned int kResSpinLimit = 1u << 22;
OSQ_SWITCH(int, g_mse_rows_order, 8);                     // osq_set_tuning("mse_rows_order", 0 | 8 | 16): the per-channel rows' loss in ATen's CPU order (8 lanes: x86 torch), 0 = order-free
OSQ_AB_KNOB(int, g_mse_dbg, 0);                            // -DOSQ_MSE_DBG builds only (tools/mse_dbg_probe.py): 1 = generated values instead of data loads, 2 = a conversion instead of the term; WRONG results, timing probes
OSQ_AB_KNOB(int, g_mse_lean, 1);                           // osq_set_tuning("mse_lean", 0): the float64 terms of the reference-order evaluations without the guarded fp32 quotient (sq_err_f64_lean; tests, A/B)
OSQ_SWITCH(int, g_mse_sum_order, 0);                      // osq_set_tuning("mse_sum_order", 0 | 8 | 16 | 64): 8 / 16 = per-row losses summed in ATen's CPU order, 64 = per-tensor losses summed as double-doubles (test modes)
OSQ_SWITCH(int, g_mse_memo, 1);                           // osq_set_tuning("mse_memo", 0): every loss evaluation of a per-tensor search streams its tensor, also a pair the search has seen (tests: equal results; A/B)
OSQ_SWITCH(unsigned int, g_res_spin_limit, 0u);            // osq_set_tuning("mse_spin_limit", n): 0 = kResSpinLimit, n > 0 = n - 1 polls (tests: 1 forces the time-out path)

struct ResidentState {                                   // workspace slice, all-zero before the first launch
    unsigned int epoch, pad0[15];                        // tags handed out so far
    unsigned int status, pad1[15];                       // sticky: 1 = a workgroup timed out waiting for a partial
    unsigned long long part[2][kResidentMaxSites][kResidentMaxBlocks][2];   // [evaluation parity][site][workgroup]{tag << 32 | low word, tag << 32 | high word}
};
static_assert(sizeof(ResidentState) == kWsResidentBytes, "ResidentState must fill its slice of the workspace");
static_assert(offsetof(ResidentState, status) == 64, "osq_persistent_status (observer.hip) reads the status word at byte 64");

struct ResidentArgs {
    const float* x;
    int64_t n;                    // flat tensor: n elements (v.batch == 0)
    osq_token_view v;             // masked / strided activation: valid tokens only
    const int64_t* lengths;
    int vec;
    TensorSearch* ts;
    ResidentState* rs;
    unsigned int spin_limit;
};

__device__ __forceinline__ unsigned int uniform(unsigned int v) {
    return static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
}
__device__ __forceinline__ double uniform_f64(double v) {
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
    const unsigned long long u = (static_cast<unsigned long long>(uniform(static_cast<unsigned int>(b >> 32))) << 32) | uniform(static_cast<unsigned int>(b));
    return __longlong_as_double(static_cast<long long>(u));
}
// the true-division form of the float64 chain, out of line: it runs only for a scale whose significand is all ones (or
// non-finite extrema), and inlined once per slot it would cost the loop its registers
__device__ __attribute__((noinline)) double sq_err4_f64_outofline(float4 a, double s, double z, double qmin, double qmax) {
    return sq_err4_f64(a, s, z, qmin, qmax);
}

// prefix sums of the clamped lengths of a masked site into pre[0..B] (B <= kResMaxBatch); all threads call
__device__ __forceinline__ void resident_prefix(const int64_t* lengths, unsigned int Bu, int64_t T, unsigned int* pre, unsigned int* s_wtot) {
    const int tid = threadIdx.x, lane = tid & (OSQ_WAVE - 1), wv = tid / OSQ_WAVE;
    unsigned int len = 0u;
    if (static_cast<unsigned int>(tid) < Bu) {
        int64_t l = lengths ? lengths[tid] : T;
        l = l < 0 ? 0 : (l > T ? T : l);
        len = static_cast<unsigned int>(l);
    }
    const unsigned int incl = wave_inclusive_scan_u32(len);
    if (lane == OSQ_WAVE - 1) s_wtot[wv] = incl;
    __syncthreads();
    unsigned int base = 0u;
#pragma unroll
    for (int k = 0; k < kResWaves; ++k) base += (k < wv) ? s_wtot[k] : 0u;
    if (tid == 0) pre[0] = 0u;
    if (static_cast<unsigned int>(tid) < Bu) pre[tid + 1] = base + incl;
    __syncthreads();
}

// float4 group g of the valid-element stream of a masked site (its tokens listed sample by sample), zero beyond the end
__device__ __forceinline__ float4 resident_load_masked(const float* x, const osq_token_view& v, int vec, const unsigned int* pre,
                                                       unsigned int Bu, unsigned int V, uint64_t g) {
    const unsigned int F = static_cast<unsigned int>(v.feat_outer * v.feat_inner);
    const unsigned int fi = static_cast<unsigned int>(v.feat_inner);
    auto token_base = [&](unsigned int j) -> const float* {       // valid token j -> its first element
        unsigned int lo = 0u, hi = Bu;                             // invariant pre[lo] <= j < pre[hi]
        while (lo + 1u < hi) {
            const unsigned int mid = (lo + hi) >> 1;
            if (pre[mid] <= j) lo = mid; else hi = mid;
        }
        return x + static_cast<int64_t>(lo) * v.stride_batch + static_cast<int64_t>(j - pre[lo]) * v.stride_token;
    };
    float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec) {                                                     // feat_inner contiguous, 16-byte aligned segments
        const unsigned int inner4 = fi / 4u, F4 = F / 4u;
        if (g < static_cast<uint64_t>(V) * F4) {
            const unsigned int j = static_cast<unsigned int>(g / F4), i = static_cast<unsigned int>(g - static_cast<uint64_t>(j) * F4);
            const unsigned int o = i / inner4, ii = i - o * inner4;
            h = reinterpret_cast<const float4*>(token_base(j) + static_cast<int64_t>(o) * v.stride_outer)[ii];
        }
    } else {
        const uint64_t E = static_cast<uint64_t>(V) * F;
        float e4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint64_t q = g * 4u + c;
            if (q < E) {
                const unsigned int j = static_cast<unsigned int>(q / F), f = static_cast<unsigned int>(q - static_cast<uint64_t>(j) * F);
                const unsigned int o = f / fi, i = f - o * fi;
                e4[c] = token_base(j)[static_cast<int64_t>(o) * v.stride_outer + static_cast<int64_t>(i) * v.stride_inner];
            }
        }
        h = make_float4(e4[0], e4[1], e4[2], e4[3]);
    }
    return h;
}
__device__ __forceinline__ float4 resident_load_flat(const float* x, int64_t n, int64_t g) {
    const int64_t n4 = n / 4;
    const int tail = static_cast<int>(n - n4 * 4);
    float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < n4) {
        h = reinterpret_cast<const float4*>(x)[g];
    } else if (g == n4 && tail) {
        const float* t = x + n4 * 4;
        h.x = t[0];
        if (tail > 1) h.y = t[1];
        if (tail > 2) h.z = t[2];
    }
    return h;
}

// Loading goes through a small LDS stage: the addressing of a masked / strided site (a bisection of the prefix sums per
// token, 64-bit strides) is sizeable code, and a lane's data registers can only be named statically -- unrolled once per
// slot it was 120-200 KB per instantiation (and beyond the unroller's budget at 32 slots: the array then lived in
// scratch).  A rolled loop loads kResStage slots into the thread's own LDS cells, a small unrolled loop moves them to the
// registers.  No barrier: every thread reads back its own cells.
// Slots KR .. K-1 of a lane (the multi-site kernel at 32 slots: 24 in registers were the most the compiler kept without
// scratch beside the site tables) stay in LDS cells of their own (`keep`), read back by ds_read_b128 in every evaluation.
constexpr int kResStage = 8;
template <int K, int KR, typename Load>
__device__ __forceinline__ void resident_fill(float4 (&hold)[KR], float4* keep, int slot0, int slots, float4* stage, Load load) {
    float4* const mine = stage + threadIdx.x;
    for (int c0 = 0; c0 < slots; c0 += kResStage) {
#pragma unroll 2
        for (int kk = 0; kk < kResStage; ++kk)
            if (c0 + kk < slots) mine[kk * kResThreads] = load(c0 + kk);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int kk = k - slot0 - c0;                             // uniform
            if (kk >= 0 && kk < kResStage && kk + c0 < slots) {
                if (k < KR) hold[k < KR ? k : 0] = mine[kk * kResThreads];
                else keep[(k - KR) * kResThreads + threadIdx.x] = mine[kk * kResThreads];
            }
        }
    }
}

// One wave: publish this workgroup's partial `p` of one search under `tag`, collect every workgroup's partial of the same
// tag and return their sum, added in the same order by every workgroup (wave-uniform).  *failed: a partial did not
// arrive within spin_limit polls.  Lane l takes workgroups l, l + 64, ...: one 16-byte sc1 load per partial (each half
// carries its own tag), all of a lane's loads in flight together; halves still stale are read again.
__device__ __forceinline__ double resident_exchange(unsigned long long (*part)[2], double p, unsigned int tag, unsigned int spin_limit, bool* failed) {
    const int lane = threadIdx.x & (OSQ_WAVE - 1);
    if (lane == 0) {
        const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(p));
        unsigned long long* slot = part[blockIdx.x];
        __hip_atomic_store(&slot[0], (static_cast<unsigned long long>(tag) << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&slot[1], (static_cast<unsigned long long>(tag) << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    constexpr int kSlots = kResidentMaxBlocks / OSQ_WAVE;
    const auto prs = __builtin_amdgcn_make_buffer_rsrc(&part[0][0], 0, static_cast<int>(gridDim.x * 16u), 0x00020000);
    double got[kSlots];
    unsigned int pending = 0u;
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        got[i] = 0.0;
        if (static_cast<unsigned int>(lane + i * OSQ_WAVE) < gridDim.x) pending |= 1u << i;
    }
    unsigned int spins = 0u;
    while (__any(pending != 0u)) {
        if (++spins > spin_limit) break;
        v4u32_t w[kSlots];
#pragma unroll
        for (int i = 0; i < kSlots; ++i)
            if (pending & (1u << i)) w[i] = __builtin_amdgcn_raw_buffer_load_b128(prs, static_cast<unsigned int>(lane + i * OSQ_WAVE) * 16u, 0, 16);
#pragma unroll
        for (int i = 0; i < kSlots; ++i) {
            if ((pending & (1u << i)) && w[i].y == tag && w[i].w == tag) {
                got[i] = __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(w[i].z) << 32) | w[i].x));
                pending &= ~(1u << i);
            }
        }
        if (pending) __builtin_amdgcn_s_sleep(1);
    }
    *failed = __any(pending != 0u);
    double mine = 0.0;
#pragma unroll
    for (int i = 0; i < kSlots; ++i) mine += got[i];               // the same order in every workgroup
    return wave_sum(mine);
}

// the squared errors of one held float4 under the pending candidate; every parameter wave-uniform (SGPRs)
__device__ __forceinline__ double resident_sq_err4(const float4& h, int f64, int fast, float sc, float zp, double sd, double rcp, float qmin, float qmax) {
    if (!f64) return sq_err4(h, sc, zp, qmin, qmax);
    if (fast) return sq_err4_f64_rcp(h, sd, rcp, zp, qmin, qmax);
    return sq_err4_f64_outofline(h, sd, zp, qmin, qmax);
}

template <int K>
__global__ __launch_bounds__(kResThreads) void msefast_resident_kernel(ResidentArgs a) {
    __shared__ unsigned int pre[kResMaxBatch + 1];
    __shared__ unsigned int s_wtot[kResWaves];
    __shared__ double s_part[kResWaves];
    __shared__ Search S;
    __shared__ float s_scale, s_zp;
    __shared__ double s_scale_d, s_count;
    __shared__ double s_rcp;
    __shared__ unsigned int s_epoch, s_done, s_fast, s_kv, s_failed;

    const int tid = threadIdx.x, lane = tid & (OSQ_WAVE - 1), wv = tid / OSQ_WAVE;
    const unsigned int NT = gridDim.x * kResThreads, gt = blockIdx.x * kResThreads + tid;
    ResidentState* rs = a.rs;
    if (tid == 0) {
        S = a.ts->S;
        s_scale = a.ts->scale;
        s_zp = a.ts->zp;
        s_scale_d = a.ts->scale_d;
        s_epoch = __hip_atomic_load(&rs->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_done = a.ts->S.done ? 1u : 0u;
        s_rcp = 1.0 / a.ts->scale_d;
        s_fast = rcp_division_exact(a.ts->scale_d, a.ts->S.x_min, a.ts->S.x_max) ? 1u : 0u;
        s_failed = 0u;
    }
    // ---- load this thread's share of the valid elements: float4 group g = gt + k * NT of the valid-element stream
    __shared__ float4 stage[kResStage * kResThreads];
    float4 hold[K];
#pragma unroll
    for (int k = 0; k < K; ++k) hold[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.v.batch == 0) {
        resident_fill<K, K>(hold, nullptr, 0, K, stage, [&](int k) { return resident_load_flat(a.x, a.n, static_cast<int64_t>(gt) + static_cast<int64_t>(k) * NT); });
        if (tid == 0) { s_count = static_cast<double>(a.n); s_kv = static_cast<unsigned int>(((a.n + 3) / 4 + NT - 1) / NT); }
    } else {
        const osq_token_view v = a.v;
        const unsigned int Bu = static_cast<unsigned int>(v.batch);
        resident_prefix(a.lengths, Bu, v.tokens, pre, s_wtot);
        const unsigned int V = pre[Bu];                                // valid tokens
        const unsigned int F = static_cast<unsigned int>(v.feat_outer * v.feat_inner);
        if (tid == 0) {
            s_count = static_cast<double>(V) * static_cast<double>(F);   // observer.py:72-84: what remove_padding keeps
            const uint64_t groups = (static_cast<uint64_t>(V) * F + 3u) / 4u;
            s_kv = static_cast<unsigned int>((groups + NT - 1) / NT);
        }
        const int vec = a.vec;
        resident_fill<K, K>(hold, nullptr, 0, K, stage, [&](int k) { return resident_load_masked(a.x, v, vec, pre, Bu, V, static_cast<uint64_t>(gt) + static_cast<uint64_t>(k) * NT); });
    }
    __syncthreads();
    const unsigned int base_tag = s_epoch + 1u;
    const double count = s_count;
    const float qmin = static_cast<float>(static_cast<int>(uniform(static_cast<unsigned int>(S.quant_min))));
    const float qmax = static_cast<float>(static_cast<int>(uniform(static_cast<unsigned int>(S.quant_max))));
    const int f64 = static_cast<int>(uniform(static_cast<unsigned int>(S.f64)));
    const int kv = static_cast<int>(uniform(s_kv));       // float4 slots of every lane that can hold valid data (the rest is padding)
    unsigned int e = 0u;
#ifdef OSQ_FINAL_TIMING
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define OSQ_RSTAMP(i) do { if (blockIdx.x == 0 && tid == 0) { const long long now = wall_clock64(); tacc[i] += now - tprev; tprev = now; } } while (0)
    if (blockIdx.x == 0 && tid == 0) tprev = wall_clock64();
#else
#define OSQ_RSTAMP(i) do { } while (0)
#endif
    // ---- one trip per loss evaluation
    while (!s_done) {                                                  // LDS, uniform: written by thread 0 before the closing barrier
        const float sc = __uint_as_float(uniform(__float_as_uint(s_scale))), zp = __uint_as_float(uniform(__float_as_uint(s_zp)));
        const double sd = uniform_f64(s_scale_d), rcp = uniform_f64(s_rcp);
        const int fast = static_cast<int>(uniform(s_fast));
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (k < kv) acc += resident_sq_err4(hold[k], f64, fast, sc, zp, sd, rcp, qmin, qmax);
            __builtin_amdgcn_sched_barrier(0);                         // one slot's temporaries at a time
        }
        acc = wave_sum(acc);
        if (lane == 0) s_part[wv] = acc;
        __syncthreads();
        OSQ_RSTAMP(0);
        if (wv == 0) {
            // ---- wave 0: publish this workgroup's partial, collect everybody's, advance the state machine
            double p = 0.0;
#pragma unroll
            for (int k = 0; k < kResWaves; ++k) p += s_part[k];
            bool failed = false;
            const double tot = resident_exchange(rs->part[e & 1u][0], p, base_tag + e, a.spin_limit, &failed);
            OSQ_RSTAMP(2);
            if (lane == 0) {
                if (failed) {                                          // poison the search instead of hanging
                    S.best_min = S.best_max = __builtin_nan("");
                    S.done = 1;
                    s_done = 1u;
                    s_failed = 1u;
                    __hip_atomic_fetch_or(&rs->status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    const double mean = tot / count;
                    S.tell(f64 ? mean : static_cast<double>(static_cast<float>(mean)));
                    if (!S.done) {
                        float scn, zpn;
                        double scd;
                        loss_qparams(S.cand_min, S.cand_max, S.quant_min, S.quant_max, S.symmetric, &scn, &zpn, &scd);
                        s_scale = scn; s_zp = zpn; s_scale_d = scd;
                        s_rcp = 1.0 / scd;
                        s_fast = rcp_division_exact(scd, S.x_min, S.x_max) ? 1u : 0u;
                    }
                    s_done = S.done ? 1u : 0u;
                }
            }
            OSQ_RSTAMP(4);
        }
        __syncthreads();
        OSQ_RSTAMP(5);
        ++e;
    }
#ifdef OSQ_FINAL_TIMING
    if (blockIdx.x == 0 && tid == 0 && e > 0)
        printf("resident K=%d kv=%d evals=%u: compute+reduce %.2f (unused %.2f) publish+poll+sum %.2f (unused %.2f) tell %.2f barrier %.2f us per evaluation\n",
               K, kv, e, tacc[0] / 100.0 / e, tacc[1] / 100.0 / e, tacc[2] / 100.0 / e, tacc[3] / 100.0 / e, tacc[4] / 100.0 / e, tacc[5] / 100.0 / e);
#endif
#undef OSQ_RSTAMP
    if (blockIdx.x == 0 && tid == 0) {
        a.ts->S = S;
        // after a time-out the workgroups may have stopped one evaluation apart: leave a gap so that no granule of this
        // launch can carry a tag of the next one
        __hip_atomic_store(&rs->epoch, base_tag + e + (s_failed ? 8u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------- several resident searches in ONE launch
//
// One evaluation of a resident search is ~1.6 us of arithmetic inside ~5.9 us: the exchange of the partial sums and the
// serial Brent step leave the VALUs idle.  The observers of one forward are independent (observer passes run with
// fake-quant off), so up to 16 of them share a launch: a lane's float4 slots are dealt out to the sites (slot_site),
// every round evaluates every unfinished site's loss, wave w of every workgroup publishes / collects the partials of
// sites w, w + 8 and advances their state machines -- the exchange and the Brent steps of the sites overlap.
struct ResidentSite {
    const float* x;
    int64_t n;                    // flat tensor (v.batch == 0)
    osq_token_view v;
    const int64_t* lengths;
    TensorSearch* ts;
    int vec, slot0, slots, pad;   // this site's float4 slots per lane: slot0 .. slot0 + slots - 1
};
struct ResidentMultiArgs {
    ResidentSite site[kResidentMaxSites];
    int n_sites;
    unsigned int spin_limit;
    ResidentState* rs;
};

template <int KT>
__global__ __launch_bounds__(kResThreads) void msefast_resident_multi_kernel(ResidentMultiArgs a) {
    __shared__ unsigned int pre[kResMaxBatch + 1];
    __shared__ unsigned int s_wtot[kResWaves];
    __shared__ double s_part[kResidentMaxSites][kResWaves];
    __shared__ Search S[kResidentMaxSites];
    __shared__ float s_scale[kResidentMaxSites], s_zp[kResidentMaxSites];
    __shared__ double s_scale_d[kResidentMaxSites], s_rcp[kResidentMaxSites], s_count[kResidentMaxSites];
    __shared__ unsigned int s_done[kResidentMaxSites], s_fast[kResidentMaxSites], s_kv[kResidentMaxSites], s_epoch, s_active, s_failed;
    __shared__ ResidentSite s_site[kResidentMaxSites];     // a by-value kernel argument indexed dynamically would be copied to scratch
    __shared__ int s_slot_site[KT];                        // slot -> site (or -1)

    const int tid = threadIdx.x, lane = tid & (OSQ_WAVE - 1), wv = tid / OSQ_WAVE;
    const unsigned int NT = gridDim.x * kResThreads, gt = blockIdx.x * kResThreads + tid;
    ResidentState* rs = a.rs;
    const int ns = a.n_sites;
    {
        // the site table, word by word (static indices into the argument), and the slot map
        constexpr int kWords = static_cast<int>(sizeof(ResidentSite) / 4);
        const unsigned int* src = reinterpret_cast<const unsigned int*>(&a.site[0]);
        unsigned int* dst = reinterpret_cast<unsigned int*>(&s_site[0]);
        for (int i = tid; i < kResidentMaxSites * kWords; i += kResThreads) dst[i] = src[i];
    }
    __syncthreads();
    if (tid < KT) {
        int site = -1;
        for (int q = 0; q < ns; ++q) site = (tid >= s_site[q].slot0 && tid < s_site[q].slot0 + s_site[q].slots) ? q : site;
        s_slot_site[tid] = site;
    }
    if (tid < ns) {
        const TensorSearch* ts = s_site[tid].ts;
        S[tid] = ts->S;
        s_scale[tid] = ts->scale;
        s_zp[tid] = ts->zp;
        s_scale_d[tid] = ts->scale_d;
        s_rcp[tid] = 1.0 / ts->scale_d;
        s_fast[tid] = rcp_division_exact(ts->scale_d, ts->S.x_min, ts->S.x_max) ? 1u : 0u;
        s_done[tid] = ts->S.done ? 1u : 0u;
    }
    if (tid == 0) { s_epoch = __hip_atomic_load(&rs->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_failed = 0u; }
    // ---- load: site after site (the prefix sums of a masked site's lengths take their turn in `pre`)
    constexpr int KR = KT > 24 ? 24 : KT, KL = KT - KR;      // register slots, LDS slots
    __shared__ float4 stage[kResStage * kResThreads];
    __shared__ float4 keep[(KL > 0 ? KL : 1) * (KL > 0 ? kResThreads : 1)];
    float4 hold[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) hold[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < KL; ++k) keep[k * kResThreads + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int si = 0; si < ns; ++si) {
        __syncthreads();                                               // `pre` of the previous site is no longer read
        const int slot0 = s_site[si].slot0, slots = s_site[si].slots;
        const float* x = s_site[si].x;
        if (s_site[si].v.batch == 0) {
            const int64_t n = s_site[si].n;
            resident_fill<KT, KR>(hold, keep, slot0, slots, stage, [&](int k) { return resident_load_flat(x, n, static_cast<int64_t>(gt) + static_cast<int64_t>(k) * NT); });
            if (tid == 0) { s_count[si] = static_cast<double>(n); s_kv[si] = static_cast<unsigned int>(((n + 3) / 4 + NT - 1) / NT); }
            continue;
        }
        const osq_token_view v = s_site[si].v;
        const unsigned int Bu = static_cast<unsigned int>(v.batch);
        resident_prefix(s_site[si].lengths, Bu, v.tokens, pre, s_wtot);
        const unsigned int V = pre[Bu];
        const unsigned int F = static_cast<unsigned int>(v.feat_outer * v.feat_inner);
        if (tid == 0) {
            s_count[si] = static_cast<double>(V) * static_cast<double>(F);
            const uint64_t groups = (static_cast<uint64_t>(V) * F + 3u) / 4u;
            s_kv[si] = static_cast<unsigned int>((groups + NT - 1) / NT);
        }
        const int vec = s_site[si].vec;
        resident_fill<KT, KR>(hold, keep, slot0, slots, stage, [&](int k) { return resident_load_masked(x, v, vec, pre, Bu, V, static_cast<uint64_t>(gt) + static_cast<uint64_t>(k) * NT); });
    }
    __syncthreads();
    if (tid == 0) {
        unsigned int act = 0u;
        for (int si = 0; si < ns; ++si) act += s_done[si] ? 0u : 1u;
        s_active = act;
    }
    __syncthreads();
    const unsigned int base_tag = s_epoch + 1u;
    unsigned int e = 0u;
    // ---- one trip per round: one loss evaluation of every unfinished site
    while (s_active != 0u) {
        int si = -1;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            // slots of a site are contiguous: a change of site closes the previous site's sum (uniform control flow)
            const int site_k = static_cast<int>(uniform(static_cast<unsigned int>(s_slot_site[k])));
            if (site_k != si) {
                if (si >= 0 && !s_done[si]) {
                    acc = wave_sum(acc);
                    if (lane == 0) s_part[si][wv] = acc;
                }
                si = site_k;
                acc = 0.0;
            }
            if (si < 0 || s_done[si] || static_cast<unsigned int>(k - s_site[si].slot0) >= s_kv[si]) continue;
            // the site's parameters go to SGPRs: as vector registers, hoisted above the arithmetic, the parameters of all
            // slots would be alive together
            const float qmin = static_cast<float>(static_cast<int>(uniform(static_cast<unsigned int>(S[si].quant_min))));
            const float qmax = static_cast<float>(static_cast<int>(uniform(static_cast<unsigned int>(S[si].quant_max))));
            const float zp = __uint_as_float(uniform(__float_as_uint(s_zp[si])));
            const float sc = __uint_as_float(uniform(__float_as_uint(s_scale[si])));
            const int f64 = static_cast<int>(uniform(static_cast<unsigned int>(S[si].f64)));
            const int fast = static_cast<int>(uniform(s_fast[si]));
            const double sd = uniform_f64(s_scale_d[si]), rcp = uniform_f64(s_rcp[si]);
            const float4 h = k < KR ? hold[k < KR ? k : 0] : keep[(k - KR) * kResThreads + tid];
            acc += resident_sq_err4(h, f64, fast, sc, zp, sd, rcp, qmin, qmax);
            __builtin_amdgcn_sched_barrier(0);                         // one slot's temporaries at a time
        }
        if (si >= 0 && !s_done[si]) {
            acc = wave_sum(acc);
            if (lane == 0) s_part[si][wv] = acc;
        }
        __syncthreads();
        for (int w = wv; w < ns; w += kResWaves) {
            if (s_done[w]) continue;
            // ---- this wave: site w's partial out, everybody's in, site w's state machine one step on
            double p = 0.0;
#pragma unroll
            for (int k = 0; k < kResWaves; ++k) p += s_part[w][k];
            bool failed = false;
            const double tot = resident_exchange(rs->part[e & 1u][w], p, base_tag + e, a.spin_limit, &failed);
            if (lane == 0) {
                if (failed) {
                    S[w].best_min = S[w].best_max = __builtin_nan("");
                    S[w].done = 1;
                    s_failed = 1u;
                    __hip_atomic_fetch_or(&rs->status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    const double mean = tot / s_count[w];
                    S[w].tell(S[w].f64 ? mean : static_cast<double>(static_cast<float>(mean)));
                    if (!S[w].done) {
                        float sc, zp;
                        double scd;
                        loss_qparams(S[w].cand_min, S[w].cand_max, S[w].quant_min, S[w].quant_max, S[w].symmetric, &sc, &zp, &scd);
                        s_scale[w] = sc; s_zp[w] = zp; s_scale_d[w] = scd;
                        s_rcp[w] = 1.0 / scd;
                        s_fast[w] = rcp_division_exact(scd, S[w].x_min, S[w].x_max) ? 1u : 0u;
                    }
                }
                if (S[w].done) {
                    s_done[w] = 1u;
                    atomicSub(&s_active, 1u);
                }
            }
        }
        __syncthreads();
        ++e;
    }
    if (blockIdx.x == 0) {
        if (tid < ns) s_site[tid].ts->S = S[tid];
        if (tid == 0) __hip_atomic_store(&rs->epoch, base_tag + e + (s_failed ? 8u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static inline int grid_for(int64_t items, int per_block, int max_blocks) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return static_cast<int>(b);
}

}  // namespace osq

using namespace osq;

extern "C" size_t osq_msefast_state_bytes(void) { return sizeof(TensorSearch) + 64; }

extern "C" int osq_msefast_rows(const float* w, int64_t rows, int64_t cols, int quant_min, int quant_max, int symmetric,
                                int one_side, int two_d, float* best_min, float* best_max, int32_t* nfev,
                                osq_stream stream) {
    OSQ_REQUIRE(w && best_min && best_max && rows > 0 && cols > 0, "msefast_rows: empty or null input");
    OSQ_REQUIRE(cols < (1ll << 31), "msefast_rows: row too long");
    OSQ_REQUIRE(one_side >= SIDE_NO && one_side <= SIDE_NEG, "msefast_rows: bad one_side");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = static_cast<int>((rows + kWavesPerBlock - 1) / kWavesPerBlock);
    const int c = static_cast<int>(cols);
    const TimingHook th = take_timing_hook(OSQ_TIME_MSEFAST_ROWS);
    // Per-channel rows are summed in the REFERENCE's order by default (osq_set_tuning("mse_rows_order", 8)): a row is shorter
    // than ATen's 32768-element grain, so torch adds it serially in an order fixed by its 256-bit vectors whatever the
    // host's thread count -- the searches are then the reference's own, iterate for iterate, and the weight ranges, hence
    // the integer weights, come out bit-equal to the reference CPU path (BASELINE north_star).  1.4-2x the time of the
    // order-free kernel (134 495 rows of RoBERTa-base: 37 ms instead of 21).  Rows of 1 .. 3072 columns: registers + LDS
    // (ATen's scalar path below one SIMD vector); 3073 .. 32767 columns: the long-row form below; rows of 32768 columns
    // and more (beyond ATen's serial grain), and "mse_rows_order" 0, take the order-free sum.
    const bool forced = g_mse_sum_order == 8 || g_mse_sum_order == 16;
    const int order = forced ? g_mse_sum_order : g_mse_rows_order;
    if (order && cols > 64 * 48 && cols < 32768) {
        // long rows (BART-large's 4096-column fc2): one wave per workgroup, the row re-read, its squared errors in up to 128 KiB of LDS
        // the attribute is per DEVICE (a process may drive several): set before every such launch -- a host-side call of
        // microseconds in front of a search of milliseconds; on failure the order-free kernel below takes the rows
        bool big_lds = true;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&msefast_rows_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4) != hipSuccess) {
            (void)hipGetLastError();
            OSQ_REQUIRE(!forced, "msefast_rows: cannot reserve the LDS of the summation-order mode for rows this long");
            big_lds = false;
        }
        if (big_lds) {
            hipExtLaunchKernelGGL((msefast_rows_kernel<0, true>), dim3(static_cast<unsigned int>(rows)), dim3(OSQ_WAVE), static_cast<size_t>(c) * sizeof(float), st,
                                  th.start, th.stop, 0, w, rows, c, quant_min, quant_max, symmetric, one_side, two_d, best_min, best_max, nfev, order);
            return check_launch("msefast_rows(reference summation order, long rows)");
        }
    }
    if (forced && cols >= 32768) return OSQ_ERR_UNSUPPORTED;
    if (order && cols <= 64 * 48) {
        const size_t lds = static_cast<size_t>(kWavesPerBlock) * c * sizeof(float);
#define OSQ_ROWS_ATEN(M) hipExtLaunchKernelGGL((msefast_rows_kernel<M, true>), dim3(grid), dim3(kThreads), lds, st, th.start, th.stop, 0, w, rows, c, \
                                            quant_min, quant_max, symmetric, one_side, two_d, best_min, best_max, nfev, order)
        if (cols <= 64 * 4) OSQ_ROWS_ATEN(4);
        else if (cols <= 64 * 16) OSQ_ROWS_ATEN(16);
        else OSQ_ROWS_ATEN(48);
#undef OSQ_ROWS_ATEN
        return check_launch("msefast_rows(reference summation order)");
    }
#define OSQ_ROWS(M) hipExtLaunchKernelGGL(msefast_rows_kernel<M>, dim3(grid), dim3(kThreads), 0, st, th.start, th.stop, 0, w, rows, c, \
                                          quant_min, quant_max, symmetric, one_side, two_d, best_min, best_max, nfev, 0)
    if (cols <= 64 * 4) OSQ_ROWS(4);
    else if (cols <= 64 * 16) OSQ_ROWS(16);
    else if (cols <= 64 * 48) OSQ_ROWS(48);
    else OSQ_ROWS(0);
#undef OSQ_ROWS
    return check_launch("msefast_rows");
}

extern "C" int osq_msefast_tensor_begin(void* state, const float* cur_minmax, int quant_min, int quant_max,
                                        int symmetric, int one_side, int two_d, int float64_input, osq_stream stream) {
    OSQ_REQUIRE(state && cur_minmax, "msefast_tensor_begin: null pointer");
    hipLaunchKernelGGL(msefast_tensor_init_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<TensorSearch*>(state), cur_minmax, quant_min, quant_max, symmetric, one_side, two_d,
                       float64_input ? 1 : 0);
    return check_launch("msefast_tensor_begin");
}

extern "C" int osq_msefast_tensor_evals_flat(void* state, const float* x, int64_t n, int n_evals, void* workspace,
                                             osq_stream stream) {
    OSQ_REQUIRE(state && x && n > 0 && workspace && n_evals >= 0, "msefast_tensor_evals_flat: bad argument");
    OSQ_REQUIRE(aligned16(x), "msefast_tensor_evals_flat: x must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Workspace ws(workspace);
    const int64_t n4 = n / 4;
    const int grid = grid_for(n4, kThreads * 2, kMaxBlocks);
    for (int e = 0; e < n_evals; ++e)
        hipLaunchKernelGGL(msefast_flat_loss_kernel, dim3(grid), dim3(kThreads), 0, st, reinterpret_cast<const float4*>(x),
                           n4, x + n4 * 4, static_cast<int>(n - n4 * 4), n, static_cast<TensorSearch*>(state),
                           ws.doubles(kFamMseFlat), ws.counter(kFamMseFlat), (g_mse_sum_order == 64 ? 1 : 0) | (g_mse_memo ? 2 : 0));
    return check_launch("msefast_tensor_evals_flat");
}

extern "C" int osq_msefast_tensor_evals_tokens(void* state, const float* x, const osq_token_view* view,
                                               const int64_t* lengths, int n_evals, void* workspace,
                                               osq_stream stream) {
    OSQ_REQUIRE(state && x && view && workspace && n_evals >= 0, "msefast_tensor_evals_tokens: bad argument");
    const osq_token_view v = *view;
    OSQ_REQUIRE(v.batch > 0 && v.tokens > 0 && v.feat_outer > 0 && v.feat_inner > 0, "msefast_tensor_evals_tokens: empty view");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Workspace ws(workspace);
    double* count = ws.doubles(kFamMseTokens) + kMaxBlocks;        // behind the partials
    hipLaunchKernelGGL(msefast_valid_count_kernel, dim3(1), dim3(64), 0, st, lengths, v.batch, v.tokens,
                       v.feat_outer * v.feat_inner, count);
    const int vec = v.stride_inner == 1 && v.feat_inner % 4 == 0 && aligned16(x) && v.stride_batch % 4 == 0 &&
                    v.stride_token % 4 == 0 && (v.feat_outer == 1 || v.stride_outer % 4 == 0);
    const int grid = grid_for(v.batch * v.tokens, kWavesPerBlock, kMaxBlocks);
    for (int e = 0; e < n_evals; ++e)
        hipLaunchKernelGGL(msefast_token_loss_kernel, dim3(grid), dim3(kThreads), 0, st, x, v, lengths, vec,
                           static_cast<TensorSearch*>(state), ws.doubles(kFamMseTokens), ws.counter(kFamMseTokens), count,
                           (g_mse_sum_order == 64 ? 1 : 0) | (g_mse_memo ? 2 : 0));
    return check_launch("msefast_tensor_evals_tokens");
}

// ---- the reference's summation order at any length (aten_order.h)
extern "C" size_t osq_ordered_sum_scratch_bytes(int64_t n, int n_sums) {
    if (n <= 0 || n_sums <= 0) return 0;
    // the widest of the forms a caller may run: fp32 on 8 / 16 lanes, float64 on 4 / 8
    size_t b = cascade_scratch_bytes(n, 8, n_sums, 4);
    b = std::max(b, cascade_scratch_bytes(n, 16, n_sums, 4));
    b = std::max(b, cascade_scratch_bytes(n, 4, n_sums, 8));
    b = std::max(b, cascade_scratch_bytes(n, 8, n_sums, 8));
    return b + 256;
}

extern "C" int osq_gather_valid_tokens(const float* x, const osq_token_view* view, const int64_t* lengths, float* out,
                                       int64_t* count_out, osq_stream stream) {
    OSQ_REQUIRE(x && view && out, "gather_valid_tokens: null pointer");
    const osq_token_view v = *view;
    OSQ_REQUIRE(v.batch > 0 && v.tokens > 0 && v.feat_outer > 0 && v.feat_inner > 0, "gather_valid_tokens: empty view");
    const int64_t tokens = v.batch * v.tokens;
    OSQ_REQUIRE(tokens < (1ll << 31) * kWavesPerBlock, "gather_valid_tokens: too many tokens");
    hipLaunchKernelGGL(gather_valid_tokens_kernel, dim3(static_cast<unsigned>((tokens + kWavesPerBlock - 1) / kWavesPerBlock)),
                       dim3(kThreads), 0, static_cast<hipStream_t>(stream), x, v, lengths, out, count_out);
    return check_launch("gather_valid_tokens");
}

extern "C" int osq_msefast_tensor_evals_ordered(void* state, const float* x_flat, int64_t n, const int64_t* n_device, int n_evals,
                                                void* scratch, size_t scratch_bytes, void* workspace, osq_stream stream) {
    OSQ_REQUIRE(state && x_flat && n > 0 && scratch && workspace && n_evals >= 0, "msefast_tensor_evals_ordered: bad argument");
    OSQ_REQUIRE(g_mse_sum_order == 8 || g_mse_sum_order == 16,
                "msefast_tensor_evals_ordered: set \"mse_sum_order\" to the reference machine's SIMD width (8 or 16) first");
    OSQ_REQUIRE(scratch_bytes >= osq_ordered_sum_scratch_bytes(n, 1), "msefast_tensor_evals_ordered: scratch smaller than osq_ordered_sum_scratch_bytes(n, 1)");
    OSQ_REQUIRE(cascade_geom(n, g_mse_sum_order / 2).P <= kCascadeMaxP, "msefast_tensor_evals_ordered: tensor too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Workspace ws(workspace);
    // one workgroup per level-1 chunk of the cascade in its finest form (float64: 256 rows x W / 2 x 4 columns), at most 2048
    OSQ_REQUIRE(aligned16(x_flat), "msefast_tensor_evals_ordered: x_flat must be 16-byte aligned");
    const CascadeGeom g = cascade_geom(n, g_mse_sum_order / 2);
    const int grid = static_cast<int>(std::min<int64_t>(g.chunks + 1, kMaxBlocks));
    for (int e = 0; e < n_evals; ++e)
        hipLaunchKernelGGL(msefast_tensor_ordered_kernel, dim3(grid), dim3(kOrdThreads), 0, st, x_flat, n, n_device,
                           static_cast<TensorSearch*>(state), scratch, ws.counter(kFamMseFlat), g_mse_sum_order | (g_mse_lean ? 256 : 0) | (g_mse_memo ? 4096 : 0));
    return check_launch("msefast_tensor_evals_ordered");
}

OSQ_AB_KNOB(int, g_ord_groups, 8);          // osq_set_tuning("mse_round_groups", n): chunk groups per workgroup of a strict round
extern "C" size_t osq_msefast_ordered_multi_bytes(int n_sites) {
    if (n_sites <= 0) return 0;
    // site table, ticket counters, block -> site map (one byte per workgroup of a round, at most kMaxBlocks per site)
    return static_cast<size_t>(n_sites) * (sizeof(OrderedSite) + kOrderedCounterBytes + static_cast<size_t>(kMaxBlocks));
}

extern "C" int osq_msefast_ordered_multi_prepare(void* table, size_t table_bytes, void* const* states, const float* const* x_flat,
                                                 const int64_t* n, const int64_t* const* n_device, void* const* scratch,
                                                 const size_t* scratch_bytes, int n_sites, int* total_blocks_out, osq_stream stream) {
    OSQ_REQUIRE(table && states && x_flat && n && scratch && scratch_bytes && total_blocks_out, "msefast_ordered_multi_prepare: null pointer");
    OSQ_REQUIRE(n_sites > 0 && n_sites <= kOrderedMaxSites, "msefast_ordered_multi_prepare: 1 .. 128 searches per table");
    OSQ_REQUIRE(table_bytes >= osq_msefast_ordered_multi_bytes(n_sites), "msefast_ordered_multi_prepare: table smaller than osq_msefast_ordered_multi_bytes(n_sites)");
    OSQ_REQUIRE(g_mse_sum_order == 8 || g_mse_sum_order == 16,
                "msefast_ordered_multi_prepare: set \"mse_sum_order\" to the reference machine's SIMD width (8 or 16) first");
    std::vector<OrderedSite> host(static_cast<size_t>(n_sites));
    char* const counters0 = static_cast<char*>(table) + static_cast<size_t>(n_sites) * sizeof(OrderedSite);
    int64_t total = 0;
    for (int i = 0; i < n_sites; ++i) {
        OSQ_REQUIRE(states[i] && x_flat[i] && n[i] > 0 && scratch[i], "msefast_ordered_multi_prepare: bad site");
        OSQ_REQUIRE(aligned16(x_flat[i]), "msefast_ordered_multi_prepare: x_flat must be 16-byte aligned");
        OSQ_REQUIRE(scratch_bytes[i] >= osq_ordered_sum_scratch_bytes(n[i], 1), "msefast_ordered_multi_prepare: scratch smaller than osq_ordered_sum_scratch_bytes(n, 1)");
        const CascadeGeom g = cascade_geom(n[i], g_mse_sum_order / 2);
        OSQ_REQUIRE(g.P <= kCascadeMaxP, "msefast_ordered_multi_prepare: tensor too large");
        OrderedSite& s = host[static_cast<size_t>(i)];
        s.x = x_flat[i];
        s.n_dev = n_device ? n_device[i] : nullptr;
        s.ts = static_cast<TensorSearch*>(states[i]);
        s.scratch = scratch[i];
        s.counters = reinterpret_cast<unsigned int*>(counters0 + static_cast<size_t>(i) * kOrderedCounterBytes);
        s.n_host = n[i];
        s.block_begin = static_cast<unsigned int>(total);
        // a workgroup of a round takes g_ord_groups chunk groups (8192 elements each, fp32 and float64 alike: 64 K elements at
        // the default) so that the loads of its next group travel under the arithmetic of the current one (aten_order.h,
        // cascade_chunks_pipelined); S = 32 (beyond 8.4 M elements): chunks of 16384 elements, 8 to a workgroup (measured best)
        const int64_t chunk_elems = static_cast<int64_t>(g.S) * g.S * g.NC;
        const int64_t per_wg = std::max<int64_t>((static_cast<int64_t>(g_ord_groups) * 8192 * (g.P > 4 ? 2 : 1)) / chunk_elems, 1);
        const int64_t groups = (g.chunks + per_wg - 1) / per_wg + 1;
        s.blocks = static_cast<unsigned int>(std::min<int64_t>(std::max<int64_t>(groups, 1), kMaxBlocks));
        s.pad[0] = s.pad[1] = 0u;
        total += s.blocks;
    }
    OSQ_REQUIRE(total < (1ll << 31), "msefast_ordered_multi_prepare: too many workgroups");
    hipStream_t st = static_cast<hipStream_t>(stream);
    std::vector<unsigned char> map(static_cast<size_t>(total));
    for (int i = 0; i < n_sites; ++i)
        std::fill(map.begin() + host[static_cast<size_t>(i)].block_begin, map.begin() + host[static_cast<size_t>(i)].block_begin + host[static_cast<size_t>(i)].blocks,
                  static_cast<unsigned char>(i));
    char* const map_dev = counters0 + static_cast<size_t>(n_sites) * kOrderedCounterBytes;
    // the counters between the table and the map stay as the caller zeroed them; table and map are in place when this returns
    OSQ_REQUIRE(hipMemcpyAsync(table, host.data(), host.size() * sizeof(OrderedSite), hipMemcpyHostToDevice, st) == hipSuccess &&
                hipMemcpyAsync(map_dev, map.data(), map.size(), hipMemcpyHostToDevice, st) == hipSuccess &&
                hipStreamSynchronize(st) == hipSuccess, "msefast_ordered_multi_prepare: copying the table failed");
    hipLaunchKernelGGL(ordered_sites_counts_kernel, dim3(1), dim3(kOrderedMaxSites), 0, st, static_cast<OrderedSite*>(table), n_sites);
    *total_blocks_out = static_cast<int>(total);
    return check_launch("msefast_ordered_multi_prepare");
}

extern "C" int osq_msefast_ordered_multi_evals(const void* table, int n_sites, int total_blocks, int n_evals, int32_t* done_out,
                                               osq_stream stream) {
    OSQ_REQUIRE(table && n_sites > 0 && n_sites <= kOrderedMaxSites && total_blocks > 0 && n_evals >= 0, "msefast_ordered_multi_evals: bad argument");
    OSQ_REQUIRE(g_mse_sum_order == 8 || g_mse_sum_order == 16,
                "msefast_ordered_multi_evals: set \"mse_sum_order\" to the reference machine's SIMD width (8 or 16) first");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const OrderedSite* sites = static_cast<const OrderedSite*>(table);
    const unsigned char* block_site = static_cast<const unsigned char*>(table) + static_cast<size_t>(n_sites) * (sizeof(OrderedSite) + kOrderedCounterBytes);
    for (int e = 0; e < n_evals; ++e)
        hipLaunchKernelGGL(msefast_tensor_ordered_multi_kernel, dim3(static_cast<unsigned>(total_blocks)), dim3(kOrdThreads), 0, st, sites, block_site, n_sites,
                           g_mse_sum_order | (g_mse_lean ? 256 : 0) | (g_mse_dbg << 9) | (g_mse_memo ? 4096 : 0));
    if (done_out) hipLaunchKernelGGL(msefast_done_multi_kernel, dim3(1), dim3(OSQ_WAVE), 0, st, sites, n_sites, done_out);
    return check_launch("msefast_ordered_multi_evals");
}

// osq_set_tuning("mse_resident", 0), or OSQ_FUSED_STEP=0 in the environment (the switch for processes that SHARE a GPU:
// persistent grids of two processes cannot be ordered against each other): per-tensor searches run one launch per evaluation
static std::atomic<int> g_mse_resident{[] { const char* e = getenv("OSQ_FUSED_STEP"); return (e && e[0] == '0') ? 0 : 1; }()};
namespace osq { bool set_msefast_tuning(const char* key, int value) {
    if (std::string(key) == "mse_resident") { g_mse_resident = value != 0; return true; }
    if (std::string(key) == "mse_memo") { g_mse_memo = value != 0; return true; }
    if (std::string(key) == "mse_rows_order") { if (value != 0 && value != 8 && value != 16) return false; g_mse_rows_order = value; return true; }
    if (std::string(key) == "mse_sum_order") { if (value != 0 && value != 8 && value != 16 && value != 64) return false; g_mse_sum_order = value; return true; }
#ifdef OSQ_TUNABLE
    if (std::string(key) == "mse_round_groups") { if (value < 1 || value > 64) return false; g_ord_groups = value; return true; }
    if (std::string(key) == "mse_lean") { g_mse_lean = value != 0; return true; }
#ifdef OSQ_MSE_DBG
    if (std::string(key) == "mse_dbg") { g_mse_dbg = value & 3; return true; }
#endif
#endif
    if (std::string(key) == "mse_spin_limit") { if (value < 0) return false; g_res_spin_limit = static_cast<unsigned int>(value); return true; }
    return false;
} }

/* The whole per-tensor search in one persistent launch (between osq_msefast_tensor_begin and _commit).
 * OSQ_ERR_UNSUPPORTED: the tensor does not fit the grid's registers (or the layout / device does not qualify) and
 * nothing was launched -- the caller runs osq_msefast_tensor_evals_* instead. */
extern "C" int osq_msefast_tensor_search(void* state, const float* x, int64_t n, const osq_token_view* view,
                                         const int64_t* lengths, void* workspace, osq_stream stream) {
    OSQ_REQUIRE(state && x && workspace, "msefast_tensor_search: null pointer");
    if (!g_mse_resident || g_mse_sum_order != 0) return OSQ_ERR_UNSUPPORTED;     // the exact-sum test mode is one launch per evaluation
    ResidentArgs a{};
    a.x = x;
    a.ts = static_cast<TensorSearch*>(state);
    a.rs = static_cast<ResidentState*>(Workspace(workspace).resident());
    int64_t elems;
    if (view) {
        const osq_token_view v = *view;
        OSQ_REQUIRE(v.batch > 0 && v.tokens > 0 && v.feat_outer > 0 && v.feat_inner > 0, "msefast_tensor_search: empty view");
        if (v.batch > kResMaxBatch) return OSQ_ERR_UNSUPPORTED;
        elems = v.batch * v.tokens * v.feat_outer * v.feat_inner;
        a.v = v;
        a.lengths = lengths;
        a.vec = v.stride_inner == 1 && v.feat_inner % 4 == 0 && aligned16(x) && v.stride_batch % 4 == 0 &&
                v.stride_token % 4 == 0 && (v.feat_outer == 1 || v.stride_outer % 4 == 0);
    } else {
        OSQ_REQUIRE(n > 0, "msefast_tensor_search: empty tensor");
        if (!aligned16(x)) return OSQ_ERR_UNSUPPORTED;
        elems = n;
        a.n = n;
    }
    a.spin_limit = g_res_spin_limit ? g_res_spin_limit - 1u : kResSpinLimit;
    static int grid = -1;       // per process; devices of one node are identical
    if (grid < 0) grid = persistent_grid_for(reinterpret_cast<const void*>(&msefast_resident_kernel<kResMaxSlots>), kResThreads);
    if (grid < 1 || grid > kResidentMaxBlocks) return OSQ_ERR_UNSUPPORTED;
    const int64_t per_k = static_cast<int64_t>(grid) * kResThreads * 4;      // elements one float4 per lane holds
    const int64_t need = (elems + per_k - 1) / per_k;
    if (need > kResMaxSlots) return OSQ_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!persistent_serialize(st)) return OSQ_ERR_UNSUPPORTED;
#define OSQ_RESIDENT(KK) hipLaunchKernelGGL(msefast_resident_kernel<KK>, dim3(grid), dim3(kResThreads), 0, st, a)
    if (need <= 1) OSQ_RESIDENT(1);
    else if (need <= 2) OSQ_RESIDENT(2);
    else if (need <= 4) OSQ_RESIDENT(4);
    else if (need <= 8) OSQ_RESIDENT(8);
    else if (need <= 16) OSQ_RESIDENT(16);
    else if (need <= 24) OSQ_RESIDENT(24);
    else OSQ_RESIDENT(32);
#undef OSQ_RESIDENT
    return check_launch("msefast_tensor_search");
}

/* Float4 slots per lane of the resident grid that a per-tensor search over `elems` elements occupies (1..32), or 0 when
 * it cannot be resident (too large, no persistent grid on this device, osq_set_tuning("mse_resident", 0)).  A group of
 * searches fits one osq_msefast_tensor_search_multi launch when there are at most osq_msefast_resident_limits' max_sites
 * of them and their slots add up to at most its max_slots. */
extern "C" int osq_msefast_resident_slots(int64_t elems) {
    if (!g_mse_resident || g_mse_sum_order != 0 || elems <= 0) return 0;
    static int grid = -1;
    if (grid < 0) grid = persistent_grid_for(reinterpret_cast<const void*>(&msefast_resident_multi_kernel<kResMaxSlots>), kResThreads);
    if (grid < 1 || grid > kResidentMaxBlocks) return 0;
    const int64_t per_k = static_cast<int64_t>(grid) * kResThreads * 4;
    const int64_t need = (elems + per_k - 1) / per_k;
    return need > kResMaxSlots ? 0 : static_cast<int>(need);
}

extern "C" int osq_msefast_resident_limits(int* max_slots, int* max_sites) {
    OSQ_REQUIRE(max_slots && max_sites, "msefast_resident_limits: null pointer");
    *max_slots = kResMaxSlots;
    *max_sites = kResidentMaxSites;
    return OSQ_OK;
}

/* Up to 16 per-tensor searches whose slots add up to at most 32 (each between its own osq_msefast_tensor_begin and _commit) in ONE persistent launch:
 * every round evaluates the pending candidate of every unfinished search.  states[i] / xs[i] / ns[i] / views[i] /
 * lengths[i] as in osq_msefast_tensor_search (views[i].batch == 0: flat, ns[i] elements).  The caller groups with
 * osq_msefast_resident_slots; a group that does not fit returns OSQ_ERR_UNSUPPORTED and launches nothing. */
extern "C" int osq_msefast_tensor_search_multi(void* const* states, const float* const* xs, const int64_t* ns,
                                               const osq_token_view* views, const int64_t* const* lengths, int n_sites,
                                               void* workspace, osq_stream stream) {
    OSQ_REQUIRE(states && xs && ns && views && lengths && workspace && n_sites > 0, "msefast_tensor_search_multi: bad argument");
    if (!g_mse_resident || g_mse_sum_order != 0 || n_sites > kResidentMaxSites) return OSQ_ERR_UNSUPPORTED;
    ResidentMultiArgs a{};
    a.n_sites = n_sites;
    a.spin_limit = g_res_spin_limit ? g_res_spin_limit - 1u : kResSpinLimit;
    a.rs = static_cast<ResidentState*>(Workspace(workspace).resident());
    int slot = 0;
    for (int i = 0; i < n_sites; ++i) {
        ResidentSite& s = a.site[i];
        OSQ_REQUIRE(states[i] && xs[i], "msefast_tensor_search_multi: null site");
        s.x = xs[i];
        s.ts = static_cast<TensorSearch*>(states[i]);
        int64_t elems;
        if (views[i].batch > 0) {
            const osq_token_view v = views[i];
            OSQ_REQUIRE(v.tokens > 0 && v.feat_outer > 0 && v.feat_inner > 0, "msefast_tensor_search_multi: empty view");
            if (v.batch > kResMaxBatch) return OSQ_ERR_UNSUPPORTED;
            elems = v.batch * v.tokens * v.feat_outer * v.feat_inner;
            s.v = v;
            s.lengths = lengths[i];
            s.vec = v.stride_inner == 1 && v.feat_inner % 4 == 0 && aligned16(s.x) && v.stride_batch % 4 == 0 &&
                    v.stride_token % 4 == 0 && (v.feat_outer == 1 || v.stride_outer % 4 == 0);
        } else {
            OSQ_REQUIRE(ns[i] > 0, "msefast_tensor_search_multi: empty tensor");
            if (!aligned16(s.x)) return OSQ_ERR_UNSUPPORTED;
            elems = ns[i];
            s.n = ns[i];
        }
        const int k = osq_msefast_resident_slots(elems);
        if (k == 0) return OSQ_ERR_UNSUPPORTED;
        s.slot0 = slot;
        s.slots = k;
        slot += k;
    }
    if (slot > kResMaxSlots) return OSQ_ERR_UNSUPPORTED;
    static int grid = -1;
    if (grid < 0) grid = persistent_grid_for(reinterpret_cast<const void*>(&msefast_resident_multi_kernel<kResMaxSlots>), kResThreads);
    if (grid < 1) return OSQ_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!persistent_serialize(st)) return OSQ_ERR_UNSUPPORTED;
#define OSQ_RESIDENT_MULTI(KK) hipLaunchKernelGGL(msefast_resident_multi_kernel<KK>, dim3(grid), dim3(kResThreads), 0, st, a)
    if (slot <= 8) OSQ_RESIDENT_MULTI(8);
    else if (slot <= 16) OSQ_RESIDENT_MULTI(16);
    else if (slot <= 24) OSQ_RESIDENT_MULTI(24);
    else OSQ_RESIDENT_MULTI(32);
#undef OSQ_RESIDENT_MULTI
    return check_launch("msefast_tensor_search_multi");
}

extern "C" int osq_msefast_tensor_done(const void* state, int32_t* done_out, osq_stream stream) {
    OSQ_REQUIRE(state && done_out, "msefast_tensor_done: null pointer");
    hipLaunchKernelGGL(msefast_done_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const TensorSearch*>(state), done_out);
    return check_launch("msefast_tensor_done");
}

__global__ void msefast_stats_kernel(const TensorSearch* __restrict__ ts, int32_t* __restrict__ out) {
    if (threadIdx.x == 0) { out[0] = ts->S.nfev; out[1] = ts->memo_count; out[2] = ts->memo_hits; out[3] = ts->S.done; }
}
extern "C" int osq_msefast_tensor_stats(const void* state, int32_t* stats_out, osq_stream stream) {
    OSQ_REQUIRE(state && stats_out, "msefast_tensor_stats: null pointer");
    hipLaunchKernelGGL(msefast_stats_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const TensorSearch*>(state), stats_out);
    return check_launch("msefast_tensor_stats");
}

extern "C" int osq_calculate_qparams_f64(const double* min_val, const double* max_val, int64_t n, int quant_min, int quant_max,
                                         int symmetric, float* scale_out, void* zero_point_out, int zp_type, osq_stream stream) {
    OSQ_REQUIRE(n >= 0 && min_val && max_val && scale_out, "calculate_qparams_f64: null pointer or n < 0");
    OSQ_REQUIRE(quant_max > quant_min, "calculate_qparams_f64: quant_max must exceed quant_min");
    if (n == 0) return OSQ_OK;
    hipLaunchKernelGGL(qparams_f64_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       min_val, max_val, n, quant_min, quant_max, symmetric, scale_out, zero_point_out, zp_type);
    return check_launch("calculate_qparams_f64");
}

extern "C" int osq_msefast_tensor_commit(const void* state, int update_rule, int64_t cnt, double* min_val,
                                         double* max_val, int quant_min, int quant_max, int symmetric,
                                         float* scale_out, void* zero_point_out, int zp_type, int32_t* nfev,
                                         int32_t* ref_float64, osq_stream stream) {
    OSQ_REQUIRE(state && min_val && max_val, "msefast_tensor_commit: null pointer");
    OSQ_REQUIRE(update_rule == OSQ_UPDATE_RUNNING || update_rule == OSQ_UPDATE_AVERAGE, "msefast_tensor_commit: bad rule");
    hipLaunchKernelGGL(msefast_commit_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const TensorSearch*>(state), update_rule, cnt, min_val, max_val, quant_min, quant_max,
                       symmetric, scale_out, zero_point_out, zp_type, nfev, ref_float64);
    return check_launch("msefast_tensor_commit");
}

#ifdef OSQ_MSE_DBG
// development build only: read and clear the rounds kernel's phase accumulators (8 x uint64)
extern "C" int osq_mse_dbg_read(unsigned long long* out) {
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(osq::g_mse_phase), sizeof(zero)) != hipSuccess) return -2;
    return hipMemcpyToSymbol(HIP_SYMBOL(osq::g_mse_phase), zero, sizeof(zero)) == hipSuccess ? 0 : -2;
}
#endif
