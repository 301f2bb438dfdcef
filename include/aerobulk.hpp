// include/aerobulk.hpp — C++ API of the MI355X-native bulk-flux engine.
//
// Source-compatible with the reference's C++ interface (brodeau/aerobulk include/aerobulk.hpp:11-42):
// same namespace, enum values, function names and argument meaning, so a caller of
// aerobulk::model() recompiles against this header and links libaerobulk_amd.so instead of
// libaerobulk_cxx.a + libaerobulk.a.  The implementation (aerobulk_amd/csrc/ab_cxx.cpp) goes
// through the C ABI of include/aerobulk_amd.h to hand-written HIP kernels; nothing is computed
// on the CPU.
#ifndef AEROBULK_AMD_AEROBULK_HPP
#define AEROBULK_AMD_AEROBULK_HPP 1

#include <string>
#include <vector>

namespace aerobulk
{
    // reference: include/aerobulk.hpp:13-21
    enum class algorithm : int
    {
        OTHER    = 0,
        COARE3p0 = 1,
        COARE3p6 = 2,
        NCAR     = 3,
        ECMWF    = 4,
        ANDREAS  = 5
    };

    // reference: src/aerobulk.cpp:22-49 ("other", "coare3p0", ..., "unknown")
    std::string algorithm_to_string(algorithm algo);

    // reference: src/aerobulk.cpp:52-65 — all `count` sizes (passed as int) must agree; returns the
    // common size, aborts otherwise (the reference asserts).
    int check_sizes(int count, ...);

    // aerobulk_model with rad_sw/rad_lw inputs and T_s output (reference: src/aerobulk.cpp:83-109).
    // Outputs are resized to the input length by the callee.
    void model(const int jt, const int Nt, algorithm algo, double zt, double zu,
               const std::vector<double> &sst, const std::vector<double> &t_zt,
               const std::vector<double> &hum_zt, const std::vector<double> &U_zu,
               const std::vector<double> &V_zu, const std::vector<double> &slp,
               std::vector<double> &QL, std::vector<double> &QH, std::vector<double> &Tau_x,
               std::vector<double> &Tau_y, std::vector<double> &Evap,
               const int Niter, const bool l_use_skin, const std::vector<double> &rad_sw,
               const std::vector<double> &rad_lw, std::vector<double> &T_s);

    // aerobulk_model without radiation / skin temperature (reference: src/aerobulk.cpp:115-138).
    void model(const int jt, const int Nt, algorithm algo, double zt, double zu,
               const std::vector<double> &sst, const std::vector<double> &t_zt,
               const std::vector<double> &hum_zt, const std::vector<double> &U_zu,
               const std::vector<double> &V_zu, const std::vector<double> &slp,
               std::vector<double> &QL, std::vector<double> &QH, std::vector<double> &Tau_x,
               std::vector<double> &Tau_y, std::vector<double> &Evap,
               const int Niter);
}

#endif
