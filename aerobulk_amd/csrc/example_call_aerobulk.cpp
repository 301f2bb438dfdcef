// Example / parity driver for the C++ API (include/aerobulk.hpp): the call pattern of the reference's
// src/tests/example_call_aerobulk.cpp — aerobulk::model() for the five algorithms on two cells,
// skin schemes on for COARE*/ECMWF.  Prints "RESULT <algo> QH1 QH2 QL1 QL2 E1 E2 Ts1 Ts2 Tx1 Tx2 Ty1 Ty2".
#include "aerobulk.hpp"

#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv)
{
    const int nbiter = argc > 1 ? std::atoi(argv[1]) : 50;
    const double rt0 = 273.15, zt = 2., zu = 10.;
    const std::vector<double> sst{22. + rt0, 22. + rt0}, t_zt{20. + rt0, 25. + rt0}, q_zt{0.012, 0.012},
        U{5., 5.}, V{0., 0.}, slp{101000., 101000.}, rsw{0., 0.}, rlw{350., 350.};
    const aerobulk::algorithm algos[5] = {aerobulk::algorithm::COARE3p0, aerobulk::algorithm::COARE3p6,
                                          aerobulk::algorithm::ECMWF, aerobulk::algorithm::NCAR,
                                          aerobulk::algorithm::ANDREAS};
    for (int ia = 0; ia < 5; ++ia) {
        std::vector<double> QL, QH, Tx, Ty, E, Ts(sst);
        if (ia < 3)
            aerobulk::model(1, 1, algos[ia], zt, zu, sst, t_zt, q_zt, U, V, slp, QL, QH, Tx, Ty, E, nbiter, true, rsw, rlw, Ts);
        else
            aerobulk::model(1, 1, algos[ia], zt, zu, sst, t_zt, q_zt, U, V, slp, QL, QH, Tx, Ty, E, nbiter);
        std::printf("RESULT %-8s %.16e %.16e %.16e %.16e %.16e %.16e %.16e %.16e %.16e %.16e %.16e %.16e\n",
                    aerobulk::algorithm_to_string(algos[ia]).c_str(), QH[0], QH[1], QL[0], QL[1], E[0], E[1], Ts[0], Ts[1],
                    Tx[0], Tx[1], Ty[0], Ty[1]);
    }
    return 0;
}
