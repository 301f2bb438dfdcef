// tools/costmodel.cpp — where do the VALU issue slots of flux_kernel go?  (host program, g++; development aid, not product)
//
// The kernel is bound by fp64 VALU issue; the hardware counters give totals per launch, not per function.  This program runs the
// PRODUCT's per-cell physics (aerobulk_amd/csrc/ab_physics.hpp, compiled for the host) on the synthetic benchmark cells with an
// instrumented scalar type: every arithmetic operation and every elementary function adds its issue cost (in fp64 slots: an FMA,
// MUL or ADD = 1, v_rcp/rsq_f64 = 4, 32-bit ops 0.5, fp32 transcendentals 2: profiles/r1_instr_rates.txt) to the code region that
// is executing (AB_REGION marks in ab_physics.hpp).  Products feeding a sum are counted as one FMA (-ffp-contract=fast).  What it
// does not see: SIMT divergence (a wave pays every path one of its lanes takes), selects/moves the compiler adds, the tile
// machinery.  Compare its total with SQ_INSTS_VALU of the same configuration to judge how much that is.
//
//   g++ -O2 -std=c++17 -o /tmp/costmodel tools/costmodel.cpp && /tmp/costmodel [algo skin nb_iter ni nj]
#define AB_FASTMATH_HOST 1
#define AB_COSTMODEL 1
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../aerobulk_amd/csrc/ab_math.hpp"

namespace ab {

// ---- accounting ---------------------------------------------------------------------------------------------------------
struct Acct {
    std::vector<int> stack;                       // region ids, innermost last
    std::vector<std::string> names{"(other)"};
    std::vector<double> excl{0.}, incl{0.};
    std::vector<long> calls{0};
    std::map<std::string, double> fn_slots;       // by elementary function
    std::map<std::string, long> fn_calls;
    double total = 0;
    int id(const char *n)
    {
        for (size_t i = 0; i < names.size(); ++i)
            if (names[i] == n) return (int)i;
        names.push_back(n); excl.push_back(0.); incl.push_back(0.); calls.push_back(0);
        return (int)names.size() - 1;
    }
    void add(double s)
    {
        total += s;
        excl[stack.empty() ? 0 : stack.back()] += s;
        // inclusive: every distinct region on the stack
        for (size_t i = 0; i < stack.size(); ++i) {
            bool seen = false;
            for (size_t j = 0; j < i; ++j) seen = seen || stack[j] == stack[i];
            if (!seen) incl[stack[i]] += s;
        }
    }
    void fn(const char *n, double s) { fn_slots[n] += s; fn_calls[n] += 1; add(s); }
};
static Acct g_acct;
struct RegionScope {
    explicit RegionScope(const char *n) { int i = g_acct.id(n); g_acct.stack.push_back(i); g_acct.calls[i]++; }
    ~RegionScope() { g_acct.stack.pop_back(); }
};
#define AB_REGION(name) ::ab::RegionScope ab_region_scope_(name)
#define AB_COUNT(name, slots) ::ab::g_acct.fn(name, slots)
#define AB_PSI_LDS_TABLES 1      // the psi functions as the tiled flux kernels evaluate them

// ---- instrumented scalar ------------------------------------------------------------------------------------------------
struct Prod;
struct CD {
    double v;
    constexpr CD() : v(0.) {}
    constexpr CD(double x) : v(x) {}
    CD(const Prod &p);
    constexpr CD(const CD &) = default;
    CD(const volatile CD &o) : v(o.v) {}      // the LDS parking slots of turb_coare are volatile
    CD &operator=(const CD &) = default;
    void operator=(const CD &o) volatile { v = o.v; }
    CD &operator=(const volatile CD &o) { v = o.v; return *this; }
    constexpr explicit operator double() const { return v; }
    constexpr explicit operator float() const { return (float)v; }
    constexpr explicit operator int() const { return (int)v; }
};
struct Prod { double v; };                               // an unrounded-yet product: becomes an FMA if it feeds a sum
inline CD::CD(const Prod &p) : v(p.v) { g_acct.add(1.); }   // used as a value: one v_mul_f64
inline Prod operator*(CD a, CD b) { return Prod{a.v * b.v}; }
inline Prod operator*(Prod a, CD b) { return Prod{CD(a).v * b.v}; }
inline Prod operator*(CD a, Prod b) { return Prod{a.v * CD(b).v}; }
inline Prod operator*(Prod a, Prod b) { return Prod{CD(a).v * CD(b).v}; }
inline CD operator+(CD a, CD b) { g_acct.add(1.); return CD(a.v + b.v); }
inline CD operator-(CD a, CD b) { g_acct.add(1.); return CD(a.v - b.v); }
inline CD operator+(Prod a, CD b) { g_acct.add(1.); return CD(a.v + b.v); }      // fma
inline CD operator+(CD a, Prod b) { g_acct.add(1.); return CD(a.v + b.v); }
inline CD operator-(Prod a, CD b) { g_acct.add(1.); return CD(a.v - b.v); }
inline CD operator-(CD a, Prod b) { g_acct.add(1.); return CD(a.v - b.v); }
inline CD operator+(Prod a, Prod b) { g_acct.add(2.); return CD(a.v + b.v); }    // mul + fma
inline CD operator-(Prod a, Prod b) { g_acct.add(2.); return CD(a.v - b.v); }
inline CD operator-(CD a) { return CD(-a.v); }                                    // source modifier: free
inline Prod operator-(Prod a) { return Prod{-a.v}; }
inline bool operator<(CD a, CD b) { g_acct.add(1.); return a.v < b.v; }          // v_cmp_f64 (+ the select it feeds, below)
inline bool operator>(CD a, CD b) { g_acct.add(1.); return a.v > b.v; }
inline bool operator<=(CD a, CD b) { g_acct.add(1.); return a.v <= b.v; }
inline bool operator>=(CD a, CD b) { g_acct.add(1.); return a.v >= b.v; }
inline bool operator==(CD a, CD b) { g_acct.add(1.); return a.v == b.v; }
inline bool operator!=(CD a, CD b) { g_acct.add(1.); return a.v != b.v; }
inline bool operator<(Prod a, CD b) { return CD(a) < b; }
inline bool operator>(Prod a, CD b) { return CD(a) > b; }
inline bool operator<=(Prod a, CD b) { return CD(a) <= b; }
inline CD &operator+=(CD &a, CD b) { a = a + b; return a; }
// '/' only appears between literals in the physics header (folded at compile time): free
inline CD operator/(CD a, CD b) { return CD(a.v / b.v); }
inline CD operator/(Prod a, CD b) { return CD(a.v / b.v); }
inline CD operator/(CD a, Prod b) { return CD(a.v / b.v); }
inline CD operator/(Prod a, Prod b) { return CD(a.v / b.v); }

// issue cost, in fp64 slots, of the elementary functions of ab_fastmath.hpp (counted from their source at the rates of
// profiles/r2_instr_rates.txt: conversions, shifts left and v_ldexp_f64 a full slot, 32-bit add / and / shift right half; a quarter-rate
// v_rcp/rsq_f64 = 4, ds_read = 0, 32-bit integer/convert = 0.5 .. 1)
template <> struct Mth<CD> {
    using R = CD;
    static R f1(const char *n, double cost, double (*f)(double), R x) { g_acct.fn(n, cost); return R(f(x.v)); }
    static R log(R x) { return f1("log", 16.5, fm::qlog, x); }
    static R log10(R x) { return f1("log10", 17.5, fm::qlog10, x); }
    static R exp(R x) { return f1("exp", 14.5, fm::qexp, x); }
    static R exp10(R x) { return f1("exp10", 15.5, fm::qexp10, x); }
    static R atan(R x) { return f1("atan", 34., fm::qatan, x); }
    static R atan_ge1(R x) { return f1("atan_ge1", 30., fm::qatan_ge1, x); }
    static R rsqrt_pos(R x) { return f1("rsqrt", 12., fm::qrsqrt_pos, x); }
    static R sqrt(R x) { return f1("sqrt", 12., fm::qsqrt, x); }
    static R sqrt_pos(R x) { return f1("sqrt_pos", 10., fm::qsqrt_pos, x); }
    static R cbrt(R x) { return f1("cbrt", 16., fm::qcbrt, x); }
    static R rcbrt(R x) { return f1("rcbrt", 13., fm::qrcbrt_mid, x); }
    static R rcp(R x) { return f1("rcp", 7., fm::qrcp, x); }
    static R rqrt(R x) { return f1("rqrt", 13.5, fm::qrqrt_mid, x); }
    static R div(R a, R b) { g_acct.fn("div", 9.); return R(fm::qdiv(a.v, b.v)); }
    static R div(Prod a, R b) { return div(R(a), b); }
    static R div(R a, Prod b) { return div(a, R(b)); }
    static R div(Prod a, Prod b) { return div(R(a), R(b)); }
    static R fma(R a, R b, R c) { g_acct.add(1.); return R(__builtin_fma(a.v, b.v, c.v)); }
    static R fma(R a, R b, Prod c) { return fma(a, b, R(c)); }
    static R abs(R x) { return R(__builtin_fabs(x.v)); }                  // source modifier: free
    static R abs(Prod x) { return R(__builtin_fabs(R(x).v)); }
    static R floor(R x) { g_acct.add(1.); return R(__builtin_floor(x.v)); }
    static R copysign(R a, R b) { g_acct.add(0.5); return R(__builtin_copysign(a.v, b.v)); }   // v_bfi_b32
};
template <> inline bool nonneg<CD>(CD x) { g_acct.add(0.5); return !__builtin_signbit(x.v); }   // v_cmp on the high word
template <> inline CD vmax<CD>(CD a, CD b) { g_acct.add(1.); return CD(a.v > b.v ? a.v : b.v); }   // v_max_f64
template <> inline CD vmin<CD>(CD a, CD b) { g_acct.add(1.); return CD(a.v < b.v ? a.v : b.v); }
inline CD vmax(Prod a, CD b) { return vmax<CD>(CD(a), b); }
inline CD vmax(CD a, Prod b) { return vmax<CD>(a, CD(b)); }
inline CD vmin(Prod a, CD b) { return vmin<CD>(CD(a), b); }
inline CD vmin(CD a, Prod b) { return vmin<CD>(a, CD(b)); }
inline CD sfloor(Prod x, CD eps) { return sfloor<CD>(CD(x), eps); }
inline CD sclamp(Prod x, CD cap) { return sclamp<CD>(CD(x), cap); }
inline bool nonneg(Prod x) { return nonneg<CD>(CD(x)); }
inline CD pow_pos(Prod x, CD y) { return pow_pos<CD>(CD(x), y); }
// polynomial tables: N-1 FMAs
CD goff_poly(CD x);
template <int N, int CL = -1> CD horner_tab(const double *tab, CD x);
template <int N, int CL = -1> CD horner_tab(const double *tab, Prod x) { return horner_tab<N, CL>(tab, CD(x)); }
inline CD goff_poly(Prod x) { return goff_poly(CD(x)); }

}  // namespace ab

#include "../aerobulk_amd/csrc/ab_physics.hpp"
#include "../aerobulk_amd/csrc/ab_launch.hpp"

namespace ab {
CD goff_poly(CD x) { g_acct.fn("poly_goff", 14.5); return CD(fm::horner_coefs<15>(kGoffA, x.v)); }
// with an entry of the LDS constant table for the second constant of the first step: N - 1 FMAs and half a slot for the read's address
template <int N, int CL> CD horner_tab(const double *tab, CD x) { g_acct.fn("poly_psi", N - 1. + (CL >= 0 ? 0.5 : 1.)); return CD(fm::horner_coefs<N>(tab, x.v)); }
}  // namespace ab

using namespace ab;

// synthetic inputs of SURVEY §8d (same generator as oracle/ab_oracle.c abo_synth_fields; the exact bits do not matter here)
static void synth(long i, long j, double f[8])
{
    static const double A[7] = {0.6180339887498949, 0.5698402909980532, 0.8191725133961645, 0.4142135623730951, 0.2360679774997897, 0.3166247903553998, 0.1231056256176606};
    static const double B[7] = {0.7548776662466927, 0.3247179572447460, 0.6710436067037893, 0.7320508075688772, 0.6457513110645906, 0.6055512754639891, 0.3588989435406740};
    static const double C[7] = {0., 0.1, 0.2, 0.3, 0.4, 0.5, 0.6};
    double r[7];
    for (int m = 0; m < 7; ++m) { const double x = i * A[m] + j * B[m] + C[m]; r[m] = x - floor(x); }
    f[0] = 274.15 + 29. * r[0];
    f[1] = f[0] - 6. + 9. * r[1];
    f[5] = 98000. + 5000. * r[2];
    f[2] = (0.55 + 0.4 * r[3]) * q_sat<double>(f[1], f[5]);
    f[3] = -14. + 28. * r[4];
    f[4] = -14. + 28. * r[5];
    f[6] = 900. * r[6];
    f[7] = 250. + 200. * r[0];
}

int main(int argc, char **argv)
{
    const char *algo = argc > 1 ? argv[1] : "coare3p6";
    const bool skin = argc > 2 ? atoi(argv[2]) != 0 : true;
    const int nb_iter = argc > 3 ? atoi(argv[3]) : 5;
    const long ni = argc > 4 ? atol(argv[4]) : 720, nj = argc > 5 ? atol(argv[5]) : 360;
    const double zt = 2., zu = 10.;
    const Heights<double> hd = make_heights<double>(zt, zu);
    Heights<CD> h;
    h.zt = hd.zt; h.zu = hd.zu; h.log_zt = hd.log_zt; h.log_zu = hd.log_zu; h.log_10 = hd.log_10; h.log_ztu = hd.log_ztu;
    h.log_zu10 = hd.log_zu10; h.fg_ca = hd.fg_ca; h.inv_zu = hd.inv_zu; h.zt_o_zu = hd.zt_o_zu; h.zt_eq_zu = hd.zt_eq_zu; h.fg_cb = hd.fg_cb;
    double sum_ql = 0.;
    const long n = ni * nj;
    for (long j = 1; j <= nj; ++j)
        for (long i = 1; i <= ni; ++i) {
            double f[8];
            synth(i * (4320 / ni), j * (3600 / nj), f);
            CellIn<CD> in;
            CD q_zt = f[2], theta;
            { AB_REGION("pre: theta(zt)"); theta = theta_from_z_p0_t_q<CD>(h.zt, CD(f[5]), CD(f[1]), q_zt); }
            in.sst = f[0]; in.theta_zt = theta; in.q_zt = q_zt; in.slp = f[5];
            { AB_REGION("pre: wind, ssq"); in.wnd = Mth<CD>::sqrt(CD(f[3]) * CD(f[3]) + CD(f[4]) * CD(f[4])); in.ssq = K<CD>::rdct_qsat_salt * q_sat<CD>(in.sst, in.slp); }
            in.qsw = CD((1. - 0.066) * f[6]); in.rlw = f[7];
            CD wl[4] = {0., skin ? (strcmp(algo, "ecmwf") ? 20. : 3.) : 0., 0., 0.};
            CellOut<CD> o;
            if (!strcmp(algo, "coare3p6")) { if (skin) turb_coare<CD, true, kSkinBoth>(h, in, nb_iter, wl, false, o); else turb_coare<CD, true, 0>(h, in, nb_iter, wl, false, o); }
            else if (!strcmp(algo, "coare3p0")) { if (skin) turb_coare<CD, false, kSkinBoth>(h, in, nb_iter, wl, false, o); else turb_coare<CD, false, 0>(h, in, nb_iter, wl, false, o); }
            else if (!strcmp(algo, "ecmwf")) { if (skin) turb_ecmwf<CD, kSkinBoth>(h, in, nb_iter, wl, o); else turb_ecmwf<CD, 0>(h, in, nb_iter, wl, o); }
            else if (!strcmp(algo, "ncar")) turb_ncar<CD>(h, in, nb_iter, o);
            else turb_andreas<CD>(h, in, nb_iter, o);
            CD tau, qh, ql, ev;
            { AB_REGION("post: bulk_formula"); bulk_formula<CD>(h.zu, o.T_s, o.q_s, o.t_zu, o.q_zu, o.Cd, o.Ch, o.Ce, in.wnd, o.Ubzu, in.slp, tau, qh, ql, ev);
              CD s = Mth<CD>::div(tau, in.wnd); CD tx = s * CD(f[3]), ty = s * CD(f[4]); (void)tx; (void)ty; }
            sum_ql += ql.v;
        }
    Acct &a = g_acct;
    printf("%s skin=%d nb_iter=%d, %ld cells: %.0f slots per cell (sum QL %.12e)\n", algo, (int)skin, nb_iter, n, a.total / n, sum_ql);
    printf("\n%-34s %10s %7s %10s %7s %9s\n", "region", "excl/cell", "%", "incl/cell", "%", "calls/cell");
    std::vector<int> ord(a.names.size());
    for (size_t i = 0; i < ord.size(); ++i) ord[i] = (int)i;
    std::sort(ord.begin(), ord.end(), [&](int x, int y) { return a.incl[x] + (x == 0 ? a.excl[0] : 0) > a.incl[y] + (y == 0 ? a.excl[0] : 0); });
    for (int i : ord)
        printf("%-34s %10.1f %6.1f%% %10.1f %6.1f%% %9.2f\n", a.names[i].c_str(), a.excl[i] / n, 100. * a.excl[i] / a.total,
               (i == 0 ? a.excl[i] : a.incl[i]) / n, 100. * (i == 0 ? a.excl[i] : a.incl[i]) / a.total, (double)a.calls[i] / n);
    printf("\n%-12s %10s %7s %10s\n", "function", "slots/cell", "%", "calls/cell");
    double fsum = 0;
    for (auto &kv : a.fn_slots) {
        printf("%-12s %10.1f %6.1f%% %10.2f\n", kv.first.c_str(), kv.second / n, 100. * kv.second / a.total, (double)a.fn_calls[kv.first] / n);
        fsum += kv.second;
    }
    printf("%-12s %10.1f %6.1f%%\n", "plain arith", (a.total - fsum) / n, 100. * (a.total - fsum) / a.total);
    return 0;
}
