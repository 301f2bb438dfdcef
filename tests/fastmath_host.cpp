// Host check of the algorithms in aerobulk_amd/csrc/ab_fastmath.hpp (seeds emulated at float accuracy):
// prints "name max_ulp_error" for each function, measured against 80-bit long double libm.
#define AB_FASTMATH_HOST 1
#include "../aerobulk_amd/csrc/ab_fastmath.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>

static double ulp_err(double got, long double want)
{
    if (want == 0.0L) return got == 0.0 ? 0.0 : 1e300;
    int e;
    std::frexp((double)want, &e);
    const long double ulp = std::ldexp(1.0L, e - 53);
    return (double)(std::fabs((long double)got - want) / ulp);
}

template <class F, class G> static void sweep(const char *name, F f, G ref, double lo, double hi, bool logspace, int n = 2000000)
{
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    double worst = 0, wx = 0;
    for (int i = 0; i < n; ++i) {
        const double t = U(rng);
        const double x = logspace ? std::exp(std::log(lo) + t * (std::log(hi) - std::log(lo))) : lo + (hi - lo) * t;
        const double e = ulp_err(f(x), ref((long double)x));
        if (e > worst) { worst = e; wx = x; }
    }
    std::printf("%s %.3f at %.17g\n", name, worst, wx);
}

int main()
{
    using namespace ab::fm;
    sweep("log_wide", [](double x) { return qlog(x); }, [](long double x) { return std::log(x); }, 1e-300, 1e300, true);
    sweep("log_near1", [](double x) { return qlog(x); }, [](long double x) { return std::log(x); }, 0.5, 2.0, false);
    sweep("log10", [](double x) { return qlog10(x); }, [](long double x) { return std::log10(x); }, 1e-12, 1e6, true);
    sweep("exp", [](double x) { return qexp(x); }, [](long double x) { return std::exp(x); }, -700.0, 700.0, false);
    sweep("exp_small", [](double x) { return qexp(x); }, [](long double x) { return std::exp(x); }, -3.0, 3.0, false);
    sweep("exp10", [](double x) { return qexp10(x); }, [](long double x) { return std::pow(10.0L, x); }, -300.0, 300.0, false);
    sweep("exp10_small", [](double x) { return qexp10(x); }, [](long double x) { return std::pow(10.0L, x); }, -4.0, 4.0, false);
    sweep("atan", [](double x) { return qatan(x); }, [](long double x) { return std::atan(x); }, -50.0, 50.0, false);
    sweep("atan_wide", [](double x) { return qatan(x); }, [](long double x) { return std::atan(x); }, 1e-8, 1e8, true);
    sweep("sqrt", [](double x) { return qsqrt(x); }, [](long double x) { return std::sqrt(x); }, 1e-200, 1e200, true);
    sweep("rcp", [](double x) { return qrcp(x); }, [](long double x) { return 1.0L / x; }, 1e-200, 1e200, true);
    sweep("cbrt", [](double x) { return qcbrt(x); }, [](long double x) { return std::cbrt(x); }, 1e-25, 1e25, true);
    sweep("rcbrt", [](double x) { return qrcbrt_mid(x); }, [](long double x) { return 1.0L / std::cbrt(x); }, 1e-25, 1e25, true);
    sweep("rqrt", [](double x) { return qrqrt_mid(x); }, [](long double x) { return 1.0L / std::sqrt(std::sqrt(x)); }, 1e-30, 1e30, true);
    sweep("rqrt_used", [](double x) { return qrqrt_mid(x); }, [](long double x) { return 1.0L / std::sqrt(std::sqrt(x)); }, 1e-15, 1e5, true);
    {   // division on pairs
        std::mt19937_64 rng(7);
        std::uniform_real_distribution<double> U(-300.0, 300.0);
        double worst = 0;
        for (int i = 0; i < 2000000; ++i) {
            const double a = std::pow(10.0, U(rng) / 4), b = std::pow(10.0, U(rng) / 4);
            const double e = ulp_err(qdiv(a, b), (long double)a / (long double)b);
            if (e > worst) worst = e;
        }
        std::printf("div %.3f at 0\n", worst);
    }
    std::printf("edge exp(-1e4)=%g exp(1e4)=%g sqrt(0)=%g cbrt(0)=%g cbrt(1e-40)=%g log(1)=%g atan(0)=%g\n", qexp(-1e4), qexp(1e4),
                qsqrt(0.0), qcbrt(0.0), qcbrt(1e-40), qlog(1.0), qatan(0.0));
    return 0;
}
