// tests/phymbl_host.cpp — the PRODUCT's helper-function header (aerobulk_amd/csrc/ab_phymbl.hpp) compiled for the host: ph_cell<FN> of
// every function over the cells of a call list.  TEST INFRASTRUCTURE (like tests/physics_host.cpp): lets the CPU suite check every
// formula of the header against the reference's own mod_phymbl (tests/golden/phymbl.npz) before a GPU is spent on it; the hardware
// seeds are emulated at their accuracy (AB_FASTMATH_HOST).  Never part of the library: ab_phymbl() has no host path.
//
//   phymbl_host <calls.bin> <out.bin>
//   calls.bin: int32 n, ncol ; ncol x n doubles ; int32 ncalls ; per call { int32 fn, flag, n_in, n_out ; double par0 ; int32 col[11] (-1: absent) }
//   out.bin  : per call n_out x n doubles
#define AB_FASTMATH_HOST 1
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../aerobulk_amd/csrc/ab_phymbl.hpp"

using namespace ab;

struct Call {
    int32_t fn, flag, n_in, n_out;
    double par0;
    int32_t col[11];
};

template <int FN> static void run_fn(const Call &c, const std::vector<const double *> &in, unsigned present, long n, std::vector<double> &out)
{
    constexpr PhShape sh = ph_shape(FN);
    const double par[2] = {c.par0, c.fn == kPhFirstGuessCoare ? 10. : 0.};      // (FIRST_GUESS_COARE: zt = par0, zu = 10 m)
    for (long k = 0; k < n; ++k) {
        double x[12] = {0.}, y[8] = {0.};
        for (int i = 0; i < 12; ++i)
            if ((present >> i) & 1u) x[i] = in[i][k];
        ph_cell<FN, double>(x, present, par, c.flag, y);
        for (int i = 0; i < sh.n_out; ++i) out[(size_t)i * n + k] = y[i];
    }
}

template <int FN = 1> static void dispatch(const Call &c, const std::vector<const double *> &in, unsigned present, long n, std::vector<double> &out)
{
    if constexpr (FN < kPhCount) {
        if (c.fn == FN) run_fn<FN>(c, in, present, n, out);
        else dispatch<FN + 1>(c, in, present, n, out);
    }
}

// e_air (mod_phymbl.f90:1706-1736) with the sweep of ph_cell<kPhEair>, the way ab_phymbl.hip drives it
static std::vector<double> e_air_host(const double *q, const double *p, long n)
{
    std::vector<double> e((size_t)n), t((size_t)n);
    const double par[2] = {0., 0.};
    for (long k = 0; k < n; ++k) {
        const double x[3] = {q[k], p[k], 0.};
        ph_cell<kPhEair, double>(x, 3u, par, 0, &e[k]);
    }
    for (int sweep = 0; sweep < 200; ++sweep) {
        double zdiff = 0.;
        for (long k = 0; k < n; ++k) {
            const double x[3] = {q[k], p[k], e[k]};
            ph_cell<kPhEair, double>(x, 7u, par, 0, &t[k]);
            zdiff += std::fabs(t[k] - e[k]);
        }
        e.swap(t);
        if (!(zdiff > 1.e-6)) break;
    }
    return e;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) return 3;
    int32_t n32, ncol, ncalls;
    if (fread(&n32, 4, 1, fi) != 1 || fread(&ncol, 4, 1, fi) != 1) return 4;
    const long n = n32;
    std::vector<double> cols((size_t)ncol * n);
    if (fread(cols.data(), 8, cols.size(), fi) != cols.size()) return 5;
    if (fread(&ncalls, 4, 1, fi) != 1) return 6;
    FILE *fo = fopen(argv[2], "wb");
    if (!fo) return 7;
    for (int ic = 0; ic < ncalls; ++ic) {
        Call c;
        if (fread(&c.fn, 4, 4, fi) != 4 || fread(&c.par0, 8, 1, fi) != 1 || fread(c.col, 4, 11, fi) != 11) return 8;
        std::vector<const double *> in(12, nullptr);
        unsigned present = 0;
        for (int i = 0; i < c.n_in; ++i)
            if (c.col[i] >= 0) { in[i] = cols.data() + (size_t)c.col[i] * n; present |= 1u << i; }
        std::vector<double> out((size_t)8 * n, 0.), e;
        if (c.fn == kPhEair) {
            e = e_air_host(in[0], in[1], n);
            for (long k = 0; k < n; ++k) out[k] = e[k];
        } else {
            if (c.fn == kPhRhoAirAdv) e = e_air_host(in[1], in[2], n);
            if (c.fn == kPhRhAir) e = e_air_host(in[0], in[2], n);
            if (!e.empty()) { in[3] = e.data(); present |= 8u; }
            dispatch<>(c, in, present, n, out);
        }
        fwrite(out.data(), 8, (size_t)c.n_out * n, fo);
    }
    fclose(fo);
    fclose(fi);
    return 0;
}
