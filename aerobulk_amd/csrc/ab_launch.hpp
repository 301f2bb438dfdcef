// ab_launch.hpp — host-side helpers shared by the kernel translation units (ab_kernels.hip, ab_turb_kernels.hip).
// Internal; include after ab_physics.hpp.
#pragma once
#include <cmath>

namespace ab {

#ifndef AB_BLOCK
#define AB_BLOCK 256
#endif
constexpr int kBlock = AB_BLOCK;  // 256: 4 waves of 64 lanes, one per SIMD

template <class R> static Heights<R> make_heights(double zt, double zu)
{
    Heights<R> h;
    h.zt = (R)zt;
    h.zu = (R)zu;
    h.log_zt = (R)log(zt);
    h.log_zu = (R)log(zu);
    h.log_10 = (R)log(10.);
    h.log_ztu = (R)log(zt / zu);
    h.log_zu10 = (R)log(zu / 10.);
    h.fg_ca = (R)(0.035 * log(10. / 0.0001) / log(zu / 0.0001));  // mod_common_coare.f90:107
    h.inv_zu = (R)(1. / zu);
    h.zt_o_zu = (R)(zt / zu);
    h.fg_cb = -(R)(0.004 * 600. * 1.2 * 1.2 * 1.2) * h.inv_zu;   // zc_b of FIRST_GUESS_COARE (mod_common_coare.f90:141), one product in R
    h.zt_eq_zu = (fabs(zu - zt) < 0.01) ? 1 : 0;
    return h;
}

// host copy of WL_COARE's solar-time test for uniform longitude 0 (mod_skin_coare.f90:146-163)
static int dawn_at_lon0(int isd)
{
    int s = isd % 86400;
    if (s < 0) s += 86400;
    const double hr = (double)s / 3600.;
    return (hr > 4.) && (hr <= 6.5);
}

}  // namespace ab
