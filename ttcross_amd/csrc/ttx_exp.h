// ttx_exp.h -- fp64 exp() whose result is bit-identical to the run-time library the reference calls.
//
// The reference's integrands call the Fortran intrinsic exp (test_crs_stdnorm.f90:168, lib/mvn_pdf.f90:82), which
// amdflang lowers to libm's exp.  That function is NOT in /root/reference: it is glibc 2.35 libm
// (sysdeps/ieee754/dbl-64/e_exp.c, the Arm Optimized Routines `exp` by Szabolcs Nagy, in glibc since 2.28), in the
// FMA build that the x86-64 ifunc selects on every AVX2+FMA host.  A device exp that differs from it in the last
// ulp of a single evaluation can re-route the pivot path of a whole run, so the published algorithm is restated
// here operation for operation -- including WHERE that build fuses a multiply-add (taken from the code of
// __exp_fma: every a*b+c below that is written fma() is one vfmadd there, every one that is not is a separate
// vmulsd/vaddsd) -- and the device evaluates exactly this sequence (v_fma_f64 is an IEEE fused multiply-add).
//
//   x = k ln2/N + r, N = 128, |r| <= ln2/2N;  exp(x) = 2^(k/N) exp(r);  2^(k/N) = 2^e * H[i] (1 + tail[i]);
//   exp(r) - 1 ~ r + r^2 (C2 + r C3) + r^4 (C4 + r C5).
//
// The table comes from gen_exp_table.py (first principles); the seven scalar constants are those of the
// published algorithm.  tests/test_host_cpu.py compares the host instantiation with the run-time libm on
// random and special arguments bit for bit; tests/test_gpu_parity.py does the same for the device code.
#pragma once
#include <stdint.h>
#include <string.h>
#include "ttx_exp_tab.h"

#if defined(__HIPCC__)
#define TTX_EXP_HD __host__ __device__ inline
__device__ static const uint64_t ttx_exp_tab_dev[256] = TTX_EXP_TAB_INIT;
#else
#define TTX_EXP_HD static inline
#endif
static const uint64_t ttx_exp_tab_host[256] = TTX_EXP_TAB_INIT;

TTX_EXP_HD uint64_t ttx_exp_u64(double x) { uint64_t b; memcpy(&b, &x, 8); return b; }
TTX_EXP_HD double ttx_exp_f64(uint64_t b) { double x; memcpy(&x, &b, 8); return x; }

TTX_EXP_HD double ttx_exp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t *T = ttx_exp_tab_dev;
#else
    const uint64_t *T = ttx_exp_tab_host;
#endif
    const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52;
    const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    const uint64_t xb = ttx_exp_u64(x);
    uint32_t abstop = (uint32_t)(xb >> 52) & 0x7ff;
    if (abstop - 0x3c9u >= 0x3fu) {
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;          // |x| < 2^-54
        if (abstop >= 0x409u) {                                       // |x| >= 1024
            if (xb == 0xfff0000000000000ull) return 0.0;              // -inf
            if (abstop >= 0x7ffu) return 1.0 + x;                     // +inf, nan
            return (xb >> 63) ? 0.0 : ttx_exp_f64(0x7ff0000000000000ull);   // underflow / overflow
        }
        abstop = 0;                                                   // 512 <= |x| < 1024: scale may leave the normal range
    }
    const double z = __builtin_fma(x, InvLn2N, Shift);
    const uint64_t ki = ttx_exp_u64(z);
    const double kd = z - Shift;
    double r = __builtin_fma(kd, NegLn2hiN, x);
    r = __builtin_fma(kd, NegLn2loN, r);
    const uint32_t idx = 2u * (uint32_t)(ki & 127u);
    uint64_t sbits = T[idx + 1] + (ki << 45);
    const double tail = ttx_exp_f64(T[idx]);
    const double p23 = __builtin_fma(r, C3, C2);
    const double rt = r + tail;
    const double r2 = r * r;
    const double p45 = __builtin_fma(r, C5, C4);
    const double lo = __builtin_fma(p23, r2, rt);
    const double r4 = r2 * r2;
    const double tmp = __builtin_fma(r4, p45, lo);
    if (abstop != 0) { const double scale = ttx_exp_f64(sbits); return __builtin_fma(scale, tmp, scale); }
    if ((ki & 0x80000000ull) == 0) {
        // k > 0: the exponent of scale may have overflowed by <= 460
        sbits -= 1009ull << 52;
        const double scale = ttx_exp_f64(sbits);
        return 0x1p1009 * __builtin_fma(scale, tmp, scale);
    }
    // k < 0: care in the subnormal range (the product is rounded on its own here, as in the library build)
    sbits += 1022ull << 52;
    const double scale = ttx_exp_f64(sbits);
    const double st = scale * tmp;
    double y = scale + st;
    if (y < 1.0) {
        double l2 = scale - y;
        l2 = l2 + st;
        const double hi = 1.0 + y;
        double t = 1.0 - hi;
        t = t + y;
        t = t + l2;
        t = t + hi;
        y = t - 1.0;
        if (y == 0.0) y = 0.0;
    }
    return 0x1p-1022 * y;
}
