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
// truncation errors quoted per table) evaluated with FMAs.  log and exp, the two most frequent, are
// table-driven (tools/gen_logtab.py, tools/gen_exptab.py): 1 640 B of tables per block in LDS, one
// ds_read per call, no division in log (an fp64 v_rcp/v_rsq/v_sqrt issues at a quarter of the FMA
// rate, profiles/r1_instr_rates.txt).  Measured accuracy on the MI355X: tests/test_gpu_math.py (max
// error in ulp vs 80-bit references).
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
#ifndef __HIPCC__   // plain C++ compiler (tests/fastmath_host.cpp, tools/costmodel.cpp): the HIP function attributes mean nothing
#define __device__
#define __host__
#define __forceinline__ inline
#endif
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
AB_FM float f_rcp(float x) { return __builtin_amdgcn_rcpf(x); }       // v_rcp_f32
AB_FM float f_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }     // v_sqrt_f32
AB_FM float f_rsq(float x) { return __builtin_amdgcn_rsqf(x); }       // v_rsq_f32
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
AB_FM float f_rcp(float x) { return 1.0f / x; }
AB_FM float f_sqrt(float x) { return std::sqrt(x); }
AB_FM float f_rsq(float x) { return 1.0f / std::sqrt(x); }
#endif

// Polynomial coefficient tables.  On the device they live in constant memory (64-byte aligned, padded to a multiple of 4
// coefficients) and ab_load<N>() fetches one with wide scalar loads (s_load_dwordx16 / x8: one SMEM instruction per 8 / 4
// coefficients) instead of two s_mov_b32 per coefficient.
template <int N> struct ab_coefs {
    double v[N];
};
constexpr int ab_pad4(int n) { return (n + 3) & ~3; }
// fp64 constants that have to sit in VGPRs.  VOP3 reads ONE scalar operand (SGPR pair or inline constant), so the second constant of
// an FMA — above all the first Horner step c[N-1] x + c[N-2] of every polynomial — costs two v_mov_b32 = one issue slot
// (profiles/r2_instr_rates.txt).  The LDS pipe is all but idle in these kernels: such constants are kept in a small LDS table
// and fetched by a broadcast ds_read_b64, which costs the VALU nothing.  The entries repeat coefficients of the tables they
// belong to (tests/test_math_tables.py checks that they are the same numbers).
enum ab_const { kC_LogQ4 = 0, kC_ExpQ2, kC_AtanP9, kC_PsikM21, kC_PsikH21, kC_PsicL24, kC_PsicG19, kC_Goff13, kC_Third, kC_Quarter,
                kC_TwoNinths, kC_5_32, kC_N = 12, kC_None = -1 };
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
AB_FM double ldsc(int i);   // entry i of the LDS constant table (below)
// CL (optional): entry of the LDS constant table that holds c[N-2] — or c[N-1] where that coefficient is alone in its chunk
// ((N-1) % 8 == 0) — so that the first step needs no VGPR copy of a scalar coefficient.
template <int N, int CL = kC_None> AB_FM double horner_coefs(const double *g, double x)
{
    constexpr int NP = ab_pad4(N), TOP = (NP - 1) & ~7;
#ifdef AB_NO_LDS_CONSTS
    constexpr int C = kC_None;
#else
    constexpr int C = CL;
#endif
    constexpr bool top_alone = (N - 1) % 8 == 0;
    double p = 0.0;
    if (C != kC_None && top_alone) p = ldsc(C);
#pragma unroll
    for (int i = TOP; i >= 0; i -= 8) {
        if (C != kC_None && top_alone && i == TOP) continue;   // the top chunk held c[N-1] only
        ab_tabp q = (ab_tabp)g + i;
        if (i == TOP)
            asm volatile("" : "+s"(q));
        else
            asm volatile("" : "+s"(q), "+v"(p));
        if (NP - i >= 8) {
            const ab_d8 t = *(const ab_d8 __attribute__((address_space(4))) *)q;
#pragma unroll
            for (int k = 7; k >= 0; --k) {
                if (i + k >= N) continue;
                if (i + k == N - 1) { if (C == kC_None) p = t[k]; }
                else if (i + k == N - 2 && C != kC_None && !top_alone) p = p_fma(t[k + 1], x, ldsc(C));
                else p = p_fmac(p, x, t[k]);
            }
        } else {
            const ab_d4 t = *(const ab_d4 __attribute__((address_space(4))) *)q;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                if (i + k >= N) continue;
                if (i + k == N - 1) { if (C == kC_None) p = t[k]; }
                else if (i + k == N - 2 && C != kC_None && !top_alone) p = p_fma(t[k + 1], x, ldsc(C));
                else p = p_fmac(p, x, t[k]);
            }
        }
    }
    return p;
}
#else
#define AB_TAB static const
template <int N> AB_FM ab_coefs<N> ab_load(const double *g)
{
    ab_coefs<N> r;
    for (int i = 0; i < N; ++i) r.v[i] = g[i];
    return r;
}
template <int N, int CL = kC_None> AB_FM double horner_coefs(const double *g, double x)
{
    double p = g[N - 1];
    for (int i = N - 2; i >= 0; --i) p = p_fmac(p, x, g[i]);
    return p;
}
#endif
AB_TAB double kConstTab[kC_N] = {-0.16667855714108057, 0.04166670739519204, 0.03923165829558719, 6.446912234869585e-09,
                                  4.251847861014437e-10, -7.386244013567441e-10, 7.426603654517986e-10, 5.607982060924723e-12,
                                  0.3333333333333333, 0.25, 0.2222222222222222, 0.15625};
// T[j] = 2^(j/64) of qexp (tools/gen_exptab.py 64)
constexpr int kExpN = 64;
AB_TAB double kExpTab[kExpN] = {
    1.0, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284,
    1.0442737824274138, 1.0556451783605572, 1.0671404006768237, 1.0787607977571199,
    1.0905077326652577, 1.102382583307841, 1.1143867425958924, 1.1265216186082418,
    1.1387886347566916, 1.1511892299529827, 1.1637248587775775, 1.1763969916502812,
    1.189207115002721, 1.202156731452703, 1.215247359980469, 1.22848053610687,
    1.241857812073484, 1.255380757024691, 1.2690509571917332, 1.2828700160787783,
    1.2968395546510096, 1.3109612115247644, 1.3252366431597413, 1.339667524053303,
    1.3542555469368927, 1.3690024229745905, 1.383909881963832, 1.3989796725383112,
    1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647,
    1.4768261459394993, 1.4929077282912648, 1.5091644275934228, 1.5255981507445384,
    1.5422108254079407, 1.559004400237837, 1.5759808451078865, 1.593142151342267,
    1.6104903319492543, 1.6280274218573478, 1.645755478153965, 1.6636765803267364,
    1.681792830507429, 1.7001063537185235, 1.718619298122478, 1.7373338352737062,
    1.7562521603732995, 1.7753764925265212, 1.7947090750031072, 1.8142521755003989,
    1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656,
    1.9152065613971474, 1.9360617934922943, 1.9571441241754002, 1.978456026387951};
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
// 1/b: rcp seed r0 = (1 - e)/b with |e| <= 2^-22, then ONE cubic step r0 (1 + e + e^2) = (1 - e^3)/b: three FMA-class operations
// instead of the four of two Newton steps; the only rounding that matters is the last FMA's (<= 0.5 ulp + 2^-66).
AB_FM double qrcp(double b)
{
    const double r = p_rcp(b);
    const double e = p_fma(-b, r, 1.0);
    return p_fma(r, p_fma(e, e, e), r);
}
// sqrt(x), x > 0 strictly (normal): g0 = x y, y = rsq(x) (2^-23); two residual corrections g += (x - g^2) (y/2).  The half
// reciprocal root is never refined: it only multiplies residuals (2^-23, then 2^-45 relative), so its own 2^-23 error enters at
// 2^-46 and 2^-68.  Six FMA-class operations after the seed (was seven with the coupled Goldschmidt update), <= 0.5 ulp + 2^-67.
AB_FM double qsqrt_pos(double x)
{
    const double y = p_rsq(x);
    double g = x * y;
    const double h = 0.5 * y;
    g = p_fma(p_fma(-g, g, x), h, g);
    return p_fma(p_fma(-g, g, x), h, g);
}
AB_FM double qsqrt(double x) { return x > 0.0 ? qsqrt_pos(x) : (x == 0.0 ? 0.0 : __builtin_nan("")); }

// ---------------------------------------------------------------- log
// log(x), x > 0 normal (1e-300 < x < 1e300), WITHOUT a division (tools/gen_logtab.py):
//    x = 2^n m, m in [1/sqrt2, sqrt2), by integer arithmetic on the high word;  i = the top six bits of (high word of m - 0x3fe6a09e):
//    64 bins of equal width in the HIGH WORD (2^-7 wide in m below 1, 2^-6 above);  r = m invc[i] - 1 (one FMA, |r| <= 0.00797);
//    log x = n ln2 + logc[i] + r + r^2 Q(r),  Q degree 5 (7.9e-18 relative to log1p(r)).
// invc[i] = a double within 2048 ulp of 1/centre of bin i and logc[i] = -log(invc[i]) of that very value, so the identity is exact;
// the value is picked so that logc[i] is within 0.002 ulp of a double (Gal's accurate tables: no rounding error of its own where it
// cancels against r); the bin that holds 1.0 (i = 37) has (1, 0): arguments next to 1 lose nothing.  On the device the 64 pairs live in LDS (1 KB per block, filled
// by lds_tables_init(); one ds_read_b128 per log on the LDS pipe).  Issue slots (profiles/r2_instr_rates.txt: 32-bit add / and /
// shift right 0.5, conversions and fp64 operations 1): 3.5 for the range reduction and the table address (the index is a shift and
// a mask of the integer the reduction already holds: no rint(64 m) through the fp64 pipe, no multiplication by the entry size),
// 13 for the rest; the atanh form, whose (m-1)/(m+1) costs a quarter-rate v_rcp_f64 plus five FMAs, took 28.
constexpr int kLogN = 64;
constexpr int kLogHi0 = 0x3fe6a09e;   // high word of the first bin's lower edge
AB_TAB double kLogTab[2 * kLogN] = {
    1.40644436128414, -0.3410647898888517, 1.3911585248550262, -0.3301368711171478,
    1.3762013820078562, -0.31932708200595056, 1.361562443597324, -0.3086328959046604,
    1.3472316620821487, -0.29805186636911646, 1.3331994085272663, -0.28758162380243396,
    1.319456451025955, -0.2772198722680176, 1.3059939344496874, -0.26696438647108617,
    1.292803361414938, -0.2568130088856901, 1.2798765744029772, -0.24676364703454223,
    1.2672057389322124, -0.23681427089685103, 1.254783327728728, -0.22696291045039368,
    1.2426021058121426, -0.21720765332825334, 1.2306551164431476, -0.20754664258973268,
    1.2189356678781564, -0.19797807460130737, 1.2074373208692561, -0.1885001970145037,
    1.1961538768665148, -0.17911130683989204, 1.1850793668733912, -0.1698097486087498,
    1.1742080409227147, -0.16059391262545036, 1.1635343581165691, -0.15146223329117808,
    1.15305297721258, -0.14241318751028523, 1.142758747710458, -0.1334452931647212,
    1.132646701412085, -0.12455710765648759, 1.1227120444296328, -0.1157472265177142,
    1.1129501496031886, -0.10701428207487695, 1.1033565493169324, -0.09835694217646686,
    1.093926928682626, -0.08977390897437691, 1.084657119066004, -0.0812639177537976,
    1.0755430919438906, -0.07282573581642736, 1.0665809530686323, -0.0644581614106576,
    1.057766936914743, -0.05616002269948234, 1.0490974014082133, -0.04793017677979518,
    1.0405688229078331, -0.03976750873701078, 1.0321777914271648, -0.03167093073582645,
    1.0239210060943469, -0.023639381155692068, 1.0157952708155107, -0.015671823748565364,
    1.007797490156108, -0.0077672468426328335, 1.0, 0.0,
    0.9844693264034321, 0.01565253791100063, 0.969555303642437, 0.030917762458129654,
    0.9550864115216001, 0.04595345932607962, 0.9410430147093403, 0.06076642874188761,
    0.9274066159941303, 0.07536317313375712, 0.9141597750029562, 0.0897499142680678,
    0.9012860337864758, 0.10393260917227824, 0.8887698486020315, 0.11791696494722517,
    0.876596527307675, 0.13170845255253164, 0.8647521718262919, 0.1453123196603031,
    0.8532236252176837, 0.15873360263975042, 0.8419984229233082, 0.17197713775426254,
    0.8310647478138687, 0.18504757162179197, 0.8204113886912318, 0.19794937100406718,
    0.8100277019437235, 0.210686831969934, 0.7999035760765578, 0.2232640884828251,
    0.7900293988675742, 0.23568512045706153, 0.7803960269295959, 0.24795376131534103,
    0.7709947574686512, 0.26007370509348016, 0.7618173020615893, 0.272048513116802,
    0.7528557622857108, 0.2838816202798194, 0.7441026070444319, 0.29557634096399765,
    0.7355506514604842, 0.30713587460826597, 0.7271930372063615, 0.31856331096343943,
    0.7190232141607429, 0.32986163504969085, 0.7110349232870385, 0.34103373183800156};
#define AB_LOGQ -0.500000000000001, 0.3333333333333342, -0.2499999997172448, 0.19999999974866162, -0.16667855714108057, 0.14286771218144256   /* Q, degree 5 */
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
// LDS tables carry a larger alignment than the kernels' tile arrays: the layout sorts by alignment, so the tables take the LOWEST addresses and
// every lookup's base fits the 16-bit offset field of ds_read (flux_kernel_cu's tiles fill 128 KB: above them each base cost a v_mov_b32)
#define AB_LDS_TABLE static __shared__ __attribute__((aligned(64)))
AB_LDS_TABLE double s_logtab[2 * kLogN];
AB_LDS_TABLE double s_exptab[kExpN];
AB_LDS_TABLE double s_ctab[kC_N];
// every kernel that may evaluate a log calls this first (all threads of a block of at least 2 kLogN = 128), then __syncthreads().
// The three loads of a thread are issued together and waited for once (three copy loops meant three round trips to the constant
// cache at the start of every block: visible in the kernels with short-lived blocks, NCAR above all).
AB_FM void lds_tables_init()
{
    const int t = (int)threadIdx.x;
    const double a = t < 2 * kLogN ? kLogTab[t] : 0.0;
    const double b = t < kExpN ? kExpTab[t] : 0.0;
    const double c = t < kC_N ? kConstTab[t] : 0.0;
    if (t < 2 * kLogN) s_logtab[t] = a;
    if (t < kExpN) s_exptab[t] = b;
    if (t < kC_N) s_ctab[t] = c;
}
// Plain read: the compiler may share one fetch between the polynomials of a basic block.  (A volatile read, one per use, measured
// the same within 0.3 %.)
#ifdef AB_LDSC_VOLATILE
AB_FM double ldsc(int i) { return *(const volatile double __attribute__((address_space(3))) *)&s_ctab[i]; }
#else
AB_FM double ldsc(int i) { return s_ctab[i]; }
#endif
// the pair (invc[i], logc[i]) at byte offset 16 i with ONE ds_read_b128
AB_FM void log_pair(int off16, double &invc, double &logc)
{
    typedef double ab_d2 __attribute__((ext_vector_type(2)));
    const ab_d2 t = *(const ab_d2 *)((const char *)s_logtab + off16);
    invc = t[0];
    logc = t[1];
}
AB_FM int p_lo32(double z) { return __double2loint(z); }
AB_FM int p_hi32(double z) { return __double2hiint(z); }
AB_FM double p_hilo(int hi, int lo) { return __hiloint2double(hi, lo); }
#else
AB_FM void log_pair(int off16, double &invc, double &logc) { invc = kLogTab[off16 >> 3]; logc = kLogTab[(off16 >> 3) + 1]; }
AB_FM double ldsc(int i) { return kConstTab[i]; }
AB_FM int p_lo32(double z) { int64_t b; std::memcpy(&b, &z, 8); return (int)(uint32_t)b; }
AB_FM int p_hi32(double z) { int64_t b; std::memcpy(&b, &z, 8); return (int)(b >> 32); }
AB_FM double p_hilo(int hi, int lo) { const uint64_t b = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double z; std::memcpy(&z, &b, 8); return z; }
#endif
// a literal that would otherwise be copied into VGPRs (second constant of an FMA): entry i of the LDS constant table holds it
AB_FM double vconst(int i, double literal)
{
#ifdef AB_NO_LDS_CONSTS
    (void)i;
    return literal;
#else
    (void)literal;
    return ldsc(i);
#endif
}
// The two short polynomials every log and every exponential evaluates (about 90 times per cell of the headline kernel) take their
// coefficients as instruction literals (two s_mov_b32 each on the scalar unit, which has the room): no s_load and no wait for it,
// no SGPR tuple to keep alive (SGPR spills of the headline kernel 45 -> 23).  The first step's second constant comes from the LDS
// constant table.  Same-box: coare3p6 skin -0.6 %, no skin -1 %, ANDREAS -2.7 % (profiles/r2_notes.md).
template <int CL> AB_FM double horner_lit6(double x, double c0, double c1, double c2, double c3, double c4, double c5)
{
    double p = p_fma(c5, x, vconst(CL, c4));
    p = p_fma(p, x, c3); p = p_fma(p, x, c2); p = p_fma(p, x, c1);
    return p_fma(p, x, c0);
}
template <int CL> AB_FM double horner_lit4(double x, double c0, double c1, double c2, double c3)
{
    double p = p_fma(c3, x, vconst(CL, c2));
    p = p_fma(p, x, c1);
    return p_fma(p, x, c0);
}
AB_FM double qlog(double x)
{
    // x = 2^n m, m in [1/sqrt2, sqrt2), by integer arithmetic on the high word (x > 0 normal): the offset makes the exponent field
    // roll over at a mantissa of sqrt2; what is left below the exponent is the distance of m's high word from the first bin's edge.
    const int hi = p_hi32(x) + (0x3ff00000 - kLogHi0);
    const int n = (hi >> 20) - 0x3ff;
    const int f = hi & 0x000fffff;
    const double m = p_hilo(f + kLogHi0, p_lo32(x));
    double invc, logc;
    log_pair((int)((unsigned)f >> 10) & 0x3f0, invc, logc);      // 16 (f >> 14)
    const double r = p_fma(m, invc, -1.0);
    const double q = horner_lit6<kC_LogQ4>(r, AB_LOGQ);
    const double nf = (double)n;
    const double lo = p_fma(nf, 1.9082149292705877e-10, p_fma(r * r, q, r));   // n ln2_lo + log1p(r)
    return p_fma(nf, 0.6931471803691238, logc + lo);          // n ln2_hi: 21 trailing zero bits, exact
}
AB_FM double qlog10(double x) { return qlog(x) * 0.4342944819032518; }

// ---------------------------------------------------------------- exp
// exp(x) with a 64-entry table (tools/gen_exptab.py 64):  x = (64 e + j) ln2/64 + r, |r| <= ln2/128:
//    exp x = 2^e T[j] (1 + r + r^2 P(r)),  T[j] = 2^(j/64),  P degree 3 (4.4e-18 relative) instead of degree 9 on |r| <= ln2/2
//    (degree 4 with the 32 entries of round 1: the byte per cell the tile sort no longer needs pays for the larger table).
// T lives in LDS next to the log table (512 B per block, lds_tables_init()); 15 VALU slots instead of 19.
// Any finite x: the exponent goes through a saturating conversion and ldexp (0 / inf beyond +-745).
#define AB_EXPQ 0.49999999999985073, 0.16666666666664534, 0.04166670739519204, 0.008333339151693509   /* P, degree 3 */
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
AB_FM double exp_t(int j) { return s_exptab[j]; }
#else
AB_FM double exp_t(int j) { return kExpTab[j]; }
#endif
// 2^e T[j] (1 + r + r^2 P(r)) for k = 64 e + j
AB_FM double exp_finish(double r, int k)
{
    const double t = exp_t(k & (kExpN - 1));
    const double q = p_fma(r * r, horner_lit4<kC_ExpQ2>(r, AB_EXPQ), r);
    return p_ldexp(p_fma(t, q, t), k >> 6);
}
AB_FM double qexp(double x)
{
    const double k = p_rint(x * 92.33248261689366);     // 64/ln2
    double r = p_fma(-k, 0.010830424696905538, x);      // ln2/64, 30-bit head: k*head exact
    r = p_fma(-k, -6.563929801064195e-13, r);           // tail
    return exp_finish(r, (int)k);
}
// 10^x
AB_FM double qexp10(double x)
{
    const double k = p_rint(x * 212.60339807279118);    // 64/log10(2)
    double r = p_fma(-k, 0.004703593680460472, x);      // log10(2)/64, 30-bit head
    r = p_fma(-k, 1.7892345153159123e-12, r);           // tail
    return exp_finish(r * 2.302585092994046, (int)k);
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
    const double p = horner_coefs<11, kC_AtanP9>(kAtanP, u);
    const double r = bhi + (p_fma(t * u, p, blo) + t);
    return p_copysign(r, x);
}

// atan(x) for x >= 1 (every atan of the psi functions: x = (1-a zeta)^(1/4) >= 1 on the unstable side): two ranges only
AB_FM double qatan_ge1(double x)
{
    const bool big = x > 2.414213562373095;
    const double t = qdiv(big ? -1.0 : x - 1.0, big ? x : x + 1.0);
    const double u = t * t;
    const double p = horner_coefs<11, kC_AtanP9>(kAtanP, u);
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
    return p_fma(r * h, p_fma(h, 0.2222222222222222, vconst(kC_Third, 0.3333333333333333)), r);
}
// x^(-1/4) for x in [2^-100, 2^100] (float range): fp32 log2/exp2 seed u (relative error d <= ~1e-6), one cubic step on
// h = 1 - x u^4 (= 4 d): u (1 + h/4 + 5 h^2/32), truncation 15/128 h^3 < 1e-17.  13.5 issue slots; x^0.75 = x * that, where two
// square roots cost 20.
AB_FM double qrqrt_mid(double x)
{
    const double u = (double)p_exp2f(p_log2f((float)x) * -0.25f);
    const double u2 = u * u;
    const double h = p_fma(-x, u2 * u2, 1.0);
    return p_fma(u * h, p_fma(h, 0.15625, vconst(kC_Quarter, 0.25)), u);
}
// The cool skin's pair (mod_phymbl.f90:2030-2044): y = 1 + x^(3/4) and y^(-1/3), x in [2^-100, 2^100].  Both fp32 seeds come from ONE
// conversion of x: u0 = x^(-1/4) by v_log_f32 / v_exp_f32, y0 = 1 + x u0 in fp32, r0 = y0^(-1/3) by a second log / exp — r0 is on its way
// while the fp64 cubic step refines u0 (the chain of qrqrt_mid followed by qrcbrt_mid converted y back to fp32 first).  Same cubic steps,
// same accuracy as qrqrt_mid / qrcbrt_mid (y0 is within 1e-6 of y: r0 within 4e-7 of y^(-1/3), h^3 terms below 1e-18).
AB_FM void qskin_pair(double x, double &y, double &rc)
{
    const float xf = (float)x;
    const float lx = p_log2f(xf);
    const float uf = p_exp2f(lx * -0.25f);
    const float rf = p_exp2f(p_log2f(__builtin_fmaf(xf, uf, 1.0f)) * -0.33333334f);
    const double u = (double)uf;
    const double u2 = u * u;
    const double h = p_fma(-x, u2 * u2, 1.0);
    const double uq = p_fma(u * h, p_fma(h, 0.15625, vconst(kC_Quarter, 0.25)), u);      // x^(-1/4)
    y = p_fma(x, uq, 1.0);
    const double r = (double)rf;
    const double g = p_fma(-(y * (r * r)), r, 1.0);      // 1 - y r^3
    rc = p_fma(r * g, p_fma(g, 0.2222222222222222, vconst(kC_Third, 0.3333333333333333)), r);
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
