// ab_math.hpp — per-precision device math for the flux kernels (gfx950 only).
//
// The flux kernels are bound by fp64 VALU issue (hundreds of transcendentals per cell against
// ~100 B of HBM traffic), so
//   * every `x**y` of the reference is strength-reduced: constant exponents become sqrt/cbrt
//     compositions or exp(y*log x);
//   * fp64 log/exp/atan/sqrt/cbrt and division come from ab_fastmath.hpp (2-4x fewer VALU
//     instructions than the ROCm device library, <= 3 ulp on the ranges used here).
// None of this is bit-exact w.r.t. libm; all of it stays far inside the 1e-10 parity bar
// (measured: tests/test_gpu_math.py, tests/test_gpu_golden.py).
#pragma once
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
#include <type_traits>
#include <hip/hip_runtime.h>
#endif

#include "ab_fastmath.hpp"

// First statement of a rarely taken block that must stay a BRANCH: an empty volatile asm cannot be speculated, so the block is not
// if-converted into selects executed by every wave (host builds: nothing)
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
#define AB_KEEP_BRANCH() asm volatile("")
#else
#define AB_KEEP_BRANCH() ((void)0)
#endif

namespace ab {

// Block prologue of every kernel whose fp64 math may take a log: fill the LDS table of fm::qlog, then a barrier.  The fp32
// paths use the hardware transcendentals and need nothing.  Must run before any early return of the kernel.
template <class R> __device__ __forceinline__ void math_tables_init()
{
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
    if constexpr (sizeof(R) == 8) {
        fm::lds_tables_init();
        __syncthreads();
    }
#endif
}

// Pointers to a lane's own LDS slots, typed with the LDS address space: volatile accesses through a generic pointer stay
// flat_load/flat_store with a 64-bit address pair each (InferAddressSpaces leaves volatile operations alone); through these
// they are ds_read/ds_write on one 32-bit base with immediate offsets.
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
template <class R> using lds_vptr = volatile R __attribute__((address_space(3))) *;
template <class R> using lds_cvptr = const volatile R __attribute__((address_space(3))) *;
#else
template <class R> using lds_vptr = volatile R *;
template <class R> using lds_cvptr = const volatile R *;
#endif

template <class R> struct Mth;

template <> struct Mth<double> {
    using R = double;
    static __device__ __forceinline__ R log(R x) { return fm::qlog(x); }      // x > 0, normal
    static __device__ __forceinline__ R log10(R x) { return fm::qlog10(x); }
    static __device__ __forceinline__ R exp(R x) { return fm::qexp(x); }
    static __device__ __forceinline__ R exp10(R x) { return fm::qexp10(x); }
    static __device__ __forceinline__ R atan(R x) { return fm::qatan(x); }
    static __device__ __forceinline__ R atan_ge1(R x) { return fm::qatan_ge1(x); }   // x >= 1
    static __device__ __forceinline__ R rsqrt_pos(R x) { return fm::qrsqrt_pos(x); }  // x > 0
    static __device__ __forceinline__ R sqrt(R x) { return fm::qsqrt(x); }        // x >= 0
    static __device__ __forceinline__ R sqrt_pos(R x) { return fm::qsqrt_pos(x); }  // x > 0 strictly
    static __device__ __forceinline__ R cbrt(R x) { return fm::qcbrt(x); }        // x >= 0
    static __device__ __forceinline__ R rcbrt(R x) { return fm::qrcbrt_mid(x); }  // 2^-100 < x < 2^100
    static __device__ __forceinline__ R rqrt(R x) { return fm::qrqrt_mid(x); }    // x^(-1/4), 2^-100 < x < 2^100
    static __device__ __forceinline__ R div(R a, R b) { return fm::qdiv(a, b); }
    static __device__ __forceinline__ R rcp(R b) { return fm::qrcp(b); }
    static __device__ __forceinline__ R fma(R a, R b, R c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ R abs(R x) { return __builtin_fabs(x); }
    static __device__ __forceinline__ R floor(R x) { return __builtin_floor(x); }
    static __device__ __forceinline__ R copysign(R a, R b) { return __builtin_copysign(a, b); }
};

// fp32 (config 5: tolerance restated at ~1e-4): hardware transcendentals (v_log_f32, v_exp_f32, v_rcp_f32, v_rsq_f32,
// v_sqrt_f32, all ~1 ulp) and nothing else — no IEEE division/sqrt fix-up sequences, no library cbrt/atan.
template <> struct Mth<float> {
    using R = float;
    static __device__ __forceinline__ R log2(R x) { return fm::p_log2f(x); }
    static __device__ __forceinline__ R exp2(R x) { return fm::p_exp2f(x); }
    static __device__ __forceinline__ R log(R x) { return log2(x) * 0.6931471805599453f; }
    static __device__ __forceinline__ R log10(R x) { return log2(x) * 0.3010299956639812f; }
    static __device__ __forceinline__ R exp(R x) { return exp2(x * 1.4426950408889634f); }
    static __device__ __forceinline__ R exp10(R x) { return exp2(x * 3.321928094887362f); }
    static __device__ __forceinline__ R rcp(R b) { return fm::f_rcp(b); }
    static __device__ __forceinline__ R div(R a, R b) { return a * fm::f_rcp(b); }
    static __device__ __forceinline__ R fma(R a, R b, R c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ R sqrt(R x) { return fm::f_sqrt(x); }
    static __device__ __forceinline__ R sqrt_pos(R x) { return fm::f_sqrt(x); }
    static __device__ __forceinline__ R rsqrt_pos(R x) { return fm::f_rsq(x); }
    static __device__ __forceinline__ R cbrt(R x) { return x > 0.f ? exp2(log2(x) * 0.33333334f) : 0.f; }
    static __device__ __forceinline__ R rcbrt(R x) { return exp2(log2(x) * -0.33333334f); }
    static __device__ __forceinline__ R rqrt(R x) { return exp2(log2(x) * -0.25f); }
    static __device__ __forceinline__ R abs(R x) { return __builtin_fabsf(x); }
    static __device__ __forceinline__ R floor(R x) { return __builtin_floorf(x); }
    static __device__ __forceinline__ R copysign(R a, R b) { return __builtin_copysignf(a, b); }
    // atan: same argument reduction as the fp64 version, odd polynomial of degree 9 on |t| <= tan(pi/8) (|err| < 3e-8)
    static __device__ __forceinline__ R atan(R x)
    {
        const R ax = abs(x);
        const bool big = ax > 2.4142135f, mid = ax > 0.41421357f;
        const R t = (big ? -1.f : (mid ? ax - 1.f : ax)) * rcp(big ? ax : (mid ? ax + 1.f : 1.f));
        const R u = t * t;
        R p = 0.0805374449538f;
        p = __builtin_fmaf(p, u, -0.138776856032f);
        p = __builtin_fmaf(p, u, 0.199777106478f);
        p = __builtin_fmaf(p, u, -0.333329491539f);
        const R r = (big ? 1.5707963267948966f : (mid ? 0.7853981633974483f : 0.f)) + __builtin_fmaf(t * u, p, t);
        return copysign(r, x);
    }
    static __device__ __forceinline__ R atan_ge1(R x) { return atan(x); }
};

// x**y for x > 0 (returns 0 for x == 0 and y > 0, like pow)
template <class R> __device__ __forceinline__ R pow_pos(R x, R y)
{
    return x > R(0) ? Mth<R>::exp(y * Mth<R>::log(x)) : R(0);
}
template <class R> __device__ __forceinline__ R vmax(R a, R b) { return a > b ? a : b; }
template <class R> __device__ __forceinline__ R vmin(R a, R b) { return a < b ? a : b; }
// Fortran SIGN(MAX(ABS(x),eps),x): sign-preserving floor
template <class R> __device__ __forceinline__ R sfloor(R x, R eps)
{
    return Mth<R>::copysign(vmax(Mth<R>::abs(x), eps), x);
}
// Fortran SIGN(MIN(ABS(x),cap),x): symmetric clamp
template <class R> __device__ __forceinline__ R sclamp(R x, R cap)
{
    return Mth<R>::copysign(vmin(Mth<R>::abs(x), cap), x);
}
// A value the reference keeps in a variable of its own and subtracts from another in two places that must agree to the last bit
// (q_s = 0.98 q_sat(T_s): Ce = (u*/Ub) q*/(q_zu - q_s) inside TURB_*, then E = rho Ub Ce (q_zu - q_s) in BULK_FORMULA: the ratio of
// the two differences is exactly 1, however ill-conditioned each of them is).  Under -ffp-contract=fast the product would be fused
// into whichever subtraction sees it as a product — q_zu - 0.98 q_sat unrounded in the iteration, the rounded loop-carried value
// after it — and the two differences part by half an ulp of q_s: 3e-8 of Q_L where q_zu - q_s = 3e-11.  The empty asm makes the
// value opaque: it is rounded once, as in the reference, and every difference sees the same number.
template <class R> __device__ __forceinline__ R rounded(R x)
{
    if constexpr (std::is_floating_point<R>::value) {
#if defined(AB_NO_ROUNDED)      // (tools/build_variant.sh: the kernels as they were, to show that the tests see the difference)
#elif defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
        asm("" : "+v"(x));
#elif defined(__x86_64__)
        asm("" : "+x"(x));
#else
        asm("" : "+r"(x));
#endif
    }
    return x;
}
// An operand of a rarely taken block whose arithmetic must stay INSIDE the block: loop-invariant code motion would otherwise hoist it in
// front of the loop, where every wave pays for it (the volatile asm is neither hoisted nor speculated)
template <class R> __device__ __forceinline__ R pinned(R x)
{
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
    if constexpr (std::is_floating_point<R>::value) asm volatile("" : "+v"(x));
#endif
    return x;
}
// `0.5 + SIGN(0.5,x)` == 1  <=>  sign bit of x clear (SIGN(0.5,-0.) = -0.5 on IEEE processors)
// (fp64 on the device: the sign bit sits in the high word — one 32-bit compare against an inline constant; the generic form compiled to a
// 64-bit integer compare with its constant copied into a VGPR pair: 3 issue slots, 25 times per cell in the cool skin alone)
template <class R> __device__ __forceinline__ bool nonneg(R x)
{
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
    if constexpr (std::is_same<R, double>::value) return __double2hiint(x) >= 0;   // (an opaque copy of the high word to keep the compare 32-bit: tried in round 4, 98 -> 139 instructions per cell: the copy costs more than the wider compare)
#endif
    return !__builtin_signbit(x);
}

}  // namespace ab
