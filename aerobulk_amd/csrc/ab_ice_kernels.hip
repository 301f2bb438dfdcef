// ab_ice_kernels.hip — the sea-ice bulk algorithms TURB_ICE_NEMO / AN05 / LU12 / LG15 as HIP kernels (gfx950).
//
// ice_kernel<R,ALGO>: one lane per cell, coalesced streaming of 5-6 input and 6-13 output fields, the nb_iter iteration in
// registers (ab_physics_ice.hpp).  Algorithmic bytes per cell, fp64: 5 in + 6 out = 88 B (+ 8 B ice concentration for
// LU12, + 8 B per OPTIONAL output).  NEMO / LU12 are HBM-bound (no iteration), AN05 / LG15 VALU-bound like the open-ocean
// algorithms.
#include "ab_kernels.hpp"
#include "ab_physics_ice.hpp"
#include "ab_launch.hpp"

namespace ab {

template <class R> struct IceArgs {
    const R *Ts_i, *theta_zt, *qs_i, *q_zt, *U_zu, *frice;
    R *out[13];
    long n;
    Heights<R> h;
    int nb_iter;
};

template <class R, int ALGO> __global__ void __launch_bounds__(kBlock) ice_kernel(const IceArgs<R> a)
{
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= a.n) return;
    IceIn<R> in;
    in.Ts_i = a.Ts_i[k];
    in.theta_zt = a.theta_zt[k];
    in.qs_i = a.qs_i[k];
    in.q_zt = a.q_zt[k];
    in.wnd = a.U_zu[k];
    in.frice = (ALGO == 3) ? a.frice[k] : R(0.);
    IceOut<R> o;
    if (ALGO == 1) {
        turb_ice_const<R>(a.h, in, KIce<R>::rCd_ice, o);
    } else if (ALGO == 2) {
        turb_ice_an05<R>(a.h, in, a.nb_iter, o);
    } else if (ALGO == 3) {   // skin drag of z0 = 0.69 mm + form drag (mod_blk_ice_lu12.f90:156-160)
        const R zi = Mth<R>::rcp(a.h.log_zu - R(-7.278818960372969));
        turb_ice_const<R>(a.h, in, K<R>::vkarmn2 * zi * zi + cdn10_f_lu13<R>(in.frice), o);
    } else {
        turb_ice_lg15<R>(a.h, in, a.frice[a.n - 1], a.nb_iter, o);   // wave-uniform load of the last cell's concentration
    }
    const R d[13] = {o.Cd, o.Ch, o.Ce, o.t_zu, o.q_zu, o.Ub, o.CdN, o.ChN, o.CeN, o.z0, o.us, o.L, o.UN10};
#pragma unroll
    for (int i = 0; i < 13; ++i)
        if (a.out[i]) a.out[i][k] = d[i];
}

template <class R, int ALGO> static hipError_t launch_t(const IceCall &c, hipStream_t stream)
{
    IceArgs<R> a;
    a.Ts_i = (const R *)c.Ts_i; a.theta_zt = (const R *)c.theta_zt; a.qs_i = (const R *)c.qs_i;
    a.q_zt = (const R *)c.q_zt; a.U_zu = (const R *)c.U_zu; a.frice = (const R *)c.frice;
    for (int i = 0; i < 13; ++i) a.out[i] = (R *)c.out[i];
    a.n = c.n;
    a.h = make_heights<R>(c.zt, c.zu);
    a.nb_iter = c.nb_iter;
    const long nblk = (c.n + kBlock - 1) / kBlock;
    if (nblk <= 0) return hipSuccess;
    hipLaunchKernelGGL((ice_kernel<R, ALGO>), dim3((unsigned)nblk), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

template <class R> static hipError_t launch_r(const IceCall &c, hipStream_t s)
{
    switch (c.algo) {
    case 1: return launch_t<R, 1>(c, s);
    case 2: return launch_t<R, 2>(c, s);
    case 3: return launch_t<R, 3>(c, s);
    case 4: return launch_t<R, 4>(c, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_turb_ice(const IceCall &c, hipStream_t stream)
{
    return c.f32 ? launch_r<float>(c, stream) : launch_r<double>(c, stream);
}

}  // namespace ab
