// ab_cxx.cpp — aerobulk::model() on top of the C ABI (include/aerobulk_amd.h).
// Interface mirrors the reference's C++ wrapper (src/aerobulk.cpp:22-138); the body is new:
// it binds the same two C symbols the reference binds (aerobulk_cxx_skin / aerobulk_cxx_no_skin,
// src/aerobulk.cpp:5-19), which here are implemented by the HIP runtime instead of Fortran.
#include "../../include/aerobulk.hpp"
#include "../../include/aerobulk_amd.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>

namespace aerobulk
{

std::string algorithm_to_string(algorithm algo)
{
    const int a = static_cast<int>(algo);
    if (a < 0 || a > 5) return "unknown";
    return ab_algo_name(a);
}

int check_sizes(int count, ...)
{
    va_list ap;
    va_start(ap, count);
    int first = 0;
    for (int i = 0; i < count; ++i) {
        const int sz = va_arg(ap, int);
        if (i == 0) first = sz;
        else if (sz != first) {
            va_end(ap);
            std::fprintf(stderr, "aerobulk::check_sizes: input #%d has %d elements, expected %d\n", i, sz, first);
            std::abort();
        }
    }
    va_end(ap);
    return first;
}

static int common_size(std::initializer_list<const std::vector<double> *> fields)
{
    const std::size_t m = (*fields.begin())->size();
    int i = 0;
    for (const auto *f : fields) {
        if (f->size() != m) {
            std::fprintf(stderr, "aerobulk::model: input #%d has %zu elements, expected %zu\n", i, f->size(), m);
            std::abort();
        }
        ++i;
    }
    return static_cast<int>(m);
}

void model(const int jt, const int Nt, algorithm algo, double zt, double zu, const std::vector<double> &sst,
           const std::vector<double> &t_zt, const std::vector<double> &hum_zt, const std::vector<double> &U_zu,
           const std::vector<double> &V_zu, const std::vector<double> &slp, std::vector<double> &QL,
           std::vector<double> &QH, std::vector<double> &Tau_x, std::vector<double> &Tau_y,
           std::vector<double> &Evap, const int Niter, const bool l_use_skin, const std::vector<double> &rad_sw,
           const std::vector<double> &rad_lw, std::vector<double> &T_s)
{
    const std::string calgo = algorithm_to_string(algo);
    const int l = static_cast<int>(calgo.size());
    const int m = common_size({&sst, &t_zt, &hum_zt, &U_zu, &V_zu, &slp, &rad_sw, &rad_lw});
    for (auto *o : {&QL, &QH, &Tau_x, &Tau_y, &Evap, &T_s}) o->resize(m);
    aerobulk_cxx_skin(&jt, &Nt, calgo.c_str(), &zt, &zu, sst.data(), t_zt.data(), hum_zt.data(), U_zu.data(),
                      V_zu.data(), slp.data(), QL.data(), QH.data(), Tau_x.data(), Tau_y.data(), Evap.data(), &Niter,
                      &l_use_skin, rad_sw.data(), rad_lw.data(), T_s.data(), &l, &m);
}

void model(const int jt, const int Nt, algorithm algo, double zt, double zu, const std::vector<double> &sst,
           const std::vector<double> &t_zt, const std::vector<double> &hum_zt, const std::vector<double> &U_zu,
           const std::vector<double> &V_zu, const std::vector<double> &slp, std::vector<double> &QL,
           std::vector<double> &QH, std::vector<double> &Tau_x, std::vector<double> &Tau_y,
           std::vector<double> &Evap, const int Niter)
{
    const std::string calgo = algorithm_to_string(algo);
    const int l = static_cast<int>(calgo.size());
    const int m = common_size({&sst, &t_zt, &hum_zt, &U_zu, &V_zu, &slp});
    for (auto *o : {&QL, &QH, &Tau_x, &Tau_y, &Evap}) o->resize(m);
    aerobulk_cxx_no_skin(&jt, &Nt, calgo.c_str(), &zt, &zu, sst.data(), t_zt.data(), hum_zt.data(), U_zu.data(),
                         V_zu.data(), slp.data(), QL.data(), QH.data(), Tau_x.data(), Tau_y.data(), Evap.data(),
                         &Niter, &l, &m);
}

}  // namespace aerobulk
