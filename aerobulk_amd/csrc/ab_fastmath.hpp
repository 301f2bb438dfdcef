// ab_fastmath.hpp — fp64 elementary functions sized for the flux kernels (gfx950).
//
// Why: the kernels are bound by fp64 VALU issue (rocprofv3: ~24k VALU instructions per cell, VALU
// busy ~86 %, HBM at 2 % of peak — profiles/r1a_*).  The ROCm device library spends 98 VALU
// instructions on log(), 105 on log10(), 83 on atan(), 42 on exp(), 22 on sqrt(), 12-14 on a
// division because it guarantees <1 ulp over the whole IEEE domain (denormals, huge arguments,
// double-double tails).  The physics here only ever feeds these functions normal, moderate
// arguments and needs ~1e-15 relative accuracy (parity bar: 1e-10 on the fluxes), so each function
// below is: hardware seed (v_rcp_f64 / v_rsq_f64 / v_log_f32+v_exp_f32) or frexp range reduction,
// one Newton/Halley step, a near-minimax polynomial (tools/gen_poly.py, 60-digit Chebyshev fits;
// truncation errors quoted per table) evaluated with FMAs.  Measured accuracy on the MI355X:
// tests/test_gpu_math.py (max error in ulp vs 50-digit references).
//
// The same source compiles for the host (AB_FASTMATH_HOST: seeds emulated with float-rounded
// values) so that tests/test_fastmath_host.py can check the algorithms without a GPU.
#pragma once

#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
#include <hip/hip_runtime.h>
#define AB_FM __device__ __forceinline__
#else
#include <cmath>
#include <cstdint>
#include <cstring>
#define AB_FM inline
#endif

namespace ab {
namespace fm {

// ---------------------------------------------------------------- primitives
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
AB_FM double p_rcp(double x) { return __builtin_amdgcn_rcp(x); }      // v_rcp_f64, ~2^-23 relative
AB_FM double p_rsq(double x) { return __builtin_amdgcn_rsq(x); }      // v_rsq_f64
AB_FM double p_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// Horner step a*b + C with a table coefficient C.  The build disables the 2-address v_fmac_f64 (build.py), so this is
// selected as the 3-address v_fma_f64 with C in an SGPR pair.  (An inline-asm v_fma_f64 did the same before, but the
// hazard recognizer pads every asm block of a dependent chain with an s_nop.)
AB_FM double p_fmac(double a, double b, double c) { return __builtin_fma(a, b, c); }
AB_FM double p_mant(double x) { return __builtin_amdgcn_frexp_mant(x); }  // in [0.5,1)
AB_FM int p_exp(double x) { return __builtin_amdgcn_frexp_exp(x); }
AB_FM double p_ldexp(double x, int e) { return __builtin_amdgcn_ldexp(x, e); }
AB_FM double p_rint(double x) { return __builtin_rint(x); }           // v_rndne_f64
AB_FM float p_log2f(float x) { return __builtin_amdgcn_logf(x); }     // v_log_f32
AB_FM float p_exp2f(float x) { return __builtin_amdgcn_exp2f(x); }    // v_exp_f32
AB_FM double p_abs(double x) { return __builtin_fabs(x); }
AB_FM double p_copysign(double a, double b) { return __builtin_copysign(a, b); }
#else
AB_FM double p_rcp(double x) { return (1.0 / x) * (1.0 + 1.2e-7); }   // deliberately float-grade (2^-23)
AB_FM double p_rsq(double x) { return (1.0 / std::sqrt(x)) * (1.0 - 1.2e-7); }
AB_FM double p_fma(double a, double b, double c) { return std::fma(a, b, c); }
AB_FM double p_fmac(double a, double b, double c) { return std::fma(a, b, c); }
AB_FM double p_mant(double x) { int e; return std::frexp(x, &e); }
AB_FM int p_exp(double x) { int e; (void)std::frexp(x, &e); return e; }
AB_FM double p_ldexp(double x, int e) { return std::ldexp(x, e); }
AB_FM double p_rint(double x) { return std::nearbyint(x); }
AB_FM float p_log2f(float x) { return std::log2(x) * (1.0f + 1e-7f); }
AB_FM float p_exp2f(float x) { return std::exp2(x) * (1.0f - 1e-7f); }
AB_FM double p_abs(double x) { return std::fabs(x); }
AB_FM double p_copysign(double a, double b) { return std::copysign(a, b); }
#endif

// Polynomial coefficient tables.  On the device they live in constant memory (64-byte aligned, padded to a multiple of 4
// coefficients) and ab_load<N>() fetches one with wide scalar loads (s_load_dwordx16 / x8: one SMEM instruction per 8 / 4
// coefficients) instead of two s_mov_b32 per coefficient.
template <int N> struct ab_coefs {
    double v[N];
};
constexpr int ab_pad4(int n) { return (n + 3) & ~3; }
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST) && !defined(AB_NO_CONST_TABLES)
#define AB_TAB __constant__ __attribute__((aligned(64)))
typedef const double __attribute__((address_space(4))) *ab_tabp;
typedef double ab_d4 __attribute__((ext_vector_type(4)));
typedef double ab_d8 __attribute__((ext_vector_type(8)));
// The coefficients of ONE polynomial evaluation.  The empty volatile asm makes the table address opaque at the use site,
// so that the scalar loads stay next to the polynomial that consumes them.  Left alone LLVM hoists every table load out
// of the iteration loop (the memory is invariant), 64 coefficients = 128 SGPRs do not fit, they are spilled to VGPR lanes
// and every coefficient use pays two v_readlane_b32 plus hazard s_nops (10 % of the loop's instructions).
template <int N> AB_FM ab_coefs<N> ab_load(const double *g)
{
    ab_tabp p = (ab_tabp)g;
#ifndef AB_TABLES_HOISTABLE
    asm volatile("" : "+s"(p));
#endif
    ab_coefs<N> r;
    constexpr int NP = ab_pad4(N);
#pragma unroll
    for (int i = 0; i + 8 <= NP; i += 8) {
        const ab_d8 t = *(const ab_d8 __attribute__((address_space(4))) *)(p + i);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i + k < N) r.v[i + k] = t[k];
    }
    if (NP % 8) {
        constexpr int i = NP & ~7;
        const ab_d4 t = *(const ab_d4 __attribute__((address_space(4))) *)(p + i);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k < N) r.v[i + k] = t[k];
    }
    return r;
}
// Horner evaluation sum_i g[i] x^i of a table, highest coefficients first, EIGHT coefficients (16 SGPRs) at a time: the
// address of the next chunk is made to depend on the partial sum, so that a 21-coefficient table occupies 16 scalar
// registers instead of 48 while it is evaluated (the kernels are at the SGPR limit: every table register beyond that is
// a spilled loop invariant, reloaded with v_readlane_b32 in the iteration loop).
template <int N> AB_FM double horner_coefs(const double *g, double x)
{
#ifdef AB_WHOLE_TABLES
    const ab_coefs<N> c = ab_load<N>(g);
    double p = c.v[N - 1];
#pragma unroll
    for (int i = N - 2; i >= 0; --i) p = p_fmac(p, x, c.v[i]);
    return p;
#else
    constexpr int NP = ab_pad4(N), TOP = (NP - 1) & ~7;
    double p = 0.0;
#pragma unroll
    for (int i = TOP; i >= 0; i -= 8) {
        ab_tabp q = (ab_tabp)g + i;
        if (i == TOP)
            asm volatile("" : "+s"(q));
        else
            asm volatile("" : "+s"(q), "+v"(p));
        if (NP - i >= 8) {
            const ab_d8 t = *(const ab_d8 __attribute__((address_space(4))) *)q;
#pragma unroll
            for (int k = 7; k >= 0; --k)
                if (i + k < N) p = (i + k == N - 1) ? t[k] : p_fmac(p, x, t[k]);
        } else {
            const ab_d4 t = *(const ab_d4 __attribute__((address_space(4))) *)q;
#pragma unroll
            for (int k = 3; k >= 0; --k)
                if (i + k < N) p = (i + k == N - 1) ? t[k] : p_fmac(p, x, t[k]);
        }
    }
    return p;
#endif
}
#else
#define AB_TAB static const
template <int N> AB_FM ab_coefs<N> ab_load(const double *g)
{
    ab_coefs<N> r;
    for (int i = 0; i < N; ++i) r.v[i] = g[i];
    return r;
}
template <int N> AB_FM double horner_coefs(const double *g, double x)
{
    double p = g[N - 1];
    for (int i = N - 2; i >= 0; --i) p = p_fmac(p, x, g[i]);
    return p;
}
#endif
AB_TAB double kLogP[ab_pad4(7)] = {0.666666666666667, 0.39999999999899505, 0.28571428625975487, 0.2222221113479508,
                          0.18182889125261723, 0.15331721600556042, 0.14616449685043406};
AB_TAB double kExpP[ab_pad4(10)] = {0.5000000000000001, 0.16666666666666669, 0.041666666666624164, 0.008333333333330065,
                           0.0013888888917196719, 0.00019841269863040545, 2.4801521322368692e-05,
                           2.7557268480310024e-06, 2.7620075879983367e-07, 2.5100375832561234e-08};
AB_TAB double kAtanP[ab_pad4(11)] = {-0.3333333333333333, 0.1999999999999552, -0.14285714284666542, 0.11111111015256361,
                            -0.09090904578123903, 0.07692183190826087, -0.06664511447381948, 0.0585814891280221,
                            -0.0508544973794026, 0.03923165829558719, -0.01917688711906226};

// ---------------------------------------------------------------- division, reciprocal, square root
// a/b: rcp seed, one Newton step on the reciprocal (2^-46), one residual correction of the quotient.
AB_FM double qdiv(double a, double b)
{
    double r = p_rcp(b);
    r = p_fma(p_fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return p_fma(p_fma(-b, q, a), r, q);
}
AB_FM double qrcp(double b)
{
    double r = p_rcp(b);
    r = p_fma(p_fma(-b, r, 1.0), r, r);
    return p_fma(p_fma(-b, r, 1.0), r, r);
}
// sqrt(x), x > 0 strictly (normal): Goldschmidt-coupled step + residual correction
AB_FM double qsqrt_pos(double x)
{
    const double y = p_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = p_fma(-h, g, 0.5);
    g = p_fma(g, r, g);
    h = p_fma(h, r, h);
    return p_fma(p_fma(-g, g, x), h, g);
}
AB_FM double qsqrt(double x) { return x > 0.0 ? qsqrt_pos(x) : (x == 0.0 ? 0.0 : __builtin_nan("")); }

// ---------------------------------------------------------------- log
// log(x), x > 0 normal.  x = m 2^e, m in [sqrt(1/2), sqrt 2); s = (m-1)/(m+1);
// log m = 2 atanh s = 2s + s^3 P(s^2), |s| <= 0.1716, P degree 6 (truncation 4.6e-18 relative).
AB_FM double qlog(double x)
{
    double m = p_mant(x);
    int e = p_exp(x);
    const bool lo = m < 0.70710678118654752;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = qdiv(f, 2.0 + f);
    const double u = s * s;
    const double p = horner_coefs<7>(kLogP, u);
    const double ef = (double)e;
    const double t = p_fma(s * u, p, ef * 2.3190468138462996e-17);  // s^3 P + e ln2_lo
    return p_fma(ef, 0.6931471805599453, (s + s) + t);
}
AB_FM double qlog10(double x) { return qlog(x) * 0.4342944819032518; }

// ---------------------------------------------------------------- exp
// exp(r) = 1 + r + r^2 P(r) on |r| <= ln2/2, P degree 9 (truncation 1.6e-17 relative)
AB_FM double exp_kernel(double r)
{
    const double p = horner_coefs<10>(kExpP, r);
    return 1.0 + p_fma(r * r, p, r);
}
// exp(x), any finite x (saturates to 0 / inf through ldexp)
AB_FM double qexp(double x)
{
    const double k = p_rint(x * 1.4426950408889634);
    double r = p_fma(-k, 0.6931471803691238, x);        // ln2_hi: 21 trailing zero bits, k*hi exact
    r = p_fma(-k, 1.9082149292705877e-10, r);           // ln2_lo
    return p_ldexp(exp_kernel(r), (int)k);
}
// 10^x
AB_FM double qexp10(double x)
{
    const double k = p_rint(x * 3.321928094887362);
    double r = p_fma(-k, 0.3010299955494702, x);        // log10(2)_hi (21 trailing zero bits)
    r = p_fma(-k, 1.1451100898021838e-10, r);           // log10(2)_lo
    return p_ldexp(exp_kernel(r * 2.302585092994046), (int)k);
}

// ---------------------------------------------------------------- atan
// atan(x): one division maps |x| to |t| <= tan(pi/8); atan t = t + t^3 P(t^2), P degree 10 (7e-18)
AB_FM double qatan(double x)
{
    const double ax = p_abs(x);
    const bool big = ax > 2.414213562373095;
    const bool mid = ax > 0.41421356237309503;
    const double num = big ? -1.0 : (mid ? ax - 1.0 : ax);
    const double den = big ? ax : (mid ? ax + 1.0 : 1.0);
    const double bhi = big ? 1.5707963267948966 : (mid ? 0.7853981633974483 : 0.0);
    const double blo = big ? 6.123233995736766e-17 : (mid ? 3.061616997868383e-17 : 0.0);
    const double t = qdiv(num, den);
    const double u = t * t;
    const double p = horner_coefs<11>(kAtanP, u);
    const double r = bhi + (p_fma(t * u, p, blo) + t);
    return p_copysign(r, x);
}

// atan(x) for x >= 1 (every atan of the psi functions: x = (1-a zeta)^(1/4) >= 1 on the unstable side): two ranges only
AB_FM double qatan_ge1(double x)
{
    const bool big = x > 2.414213562373095;
    const double t = qdiv(big ? -1.0 : x - 1.0, big ? x : x + 1.0);
    const double u = t * t;
    const double p = horner_coefs<11>(kAtanP, u);
    return (big ? 1.5707963267948966 : 0.7853981633974483) + (p_fma(t * u, p, big ? 6.123233995736766e-17 : 3.061616997868383e-17) + t);
}
// 1/sqrt(x), x > 0 normal: rsq seed, one Newton step, one residual correction (<= 1 ulp)
AB_FM double qrsqrt_pos(double x)
{
    double y = p_rsq(x);
    y = p_fma(y * 0.5, p_fma(-(x * y), y, 1.0), y);
    return p_fma(y * 0.5, p_fma(-(x * y), y, 1.0), y);
}

// ---------------------------------------------------------------- cube roots
// x^(-1/3) for x in [2^-100, 2^100] (float range): fp32 log2/exp2 seed (~3e-7), one Halley step (cubic)
AB_FM double qrcbrt_mid(double x)
{
    const double r = (double)p_exp2f(p_log2f((float)x) * -0.33333334f);
    const double h = p_fma(-(x * (r * r)), r, 1.0);      // 1 - x r^3
    return p_fma(r * h, p_fma(h, 0.2222222222222222, 0.3333333333333333), r);
}
// cbrt(x), x >= 0.  Arguments below 2^-100 return 0 (callers add the square of it to O(1) terms).
AB_FM double qcbrt(double x)
{
    if (!(x > 7.888609052210118e-31)) return x == x ? 0.0 : x;
    const double r = qrcbrt_mid(x);
    return x * r * r;
}

}  // namespace fm
}  // namespace ab
