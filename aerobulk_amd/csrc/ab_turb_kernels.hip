// ab_turb_kernels.hip — the TURB_<algo> routines as stand-alone kernels (gfx950).
//
// turb_kernel<R,ALGO,SKIN>: what a caller of TURB_COARE3P6 / turb_coare3p0 / turb_ecmwf / turb_ncar / turb_andreas gets
// (mod_blk_coare3p6.f90:123-131, mod_blk_coare3p0.f90:54, mod_blk_ecmwf.f90:63, mod_blk_ncar.f90:57, mod_blk_andreas.f90:66):
// transfer coefficients, air temperature / humidity adjusted to zu, bulk wind, the OPTIONAL diagnostics, and T_s / q_s
// updated in place by the cool-skin (bit 0 of SKIN) and / or warm-layer (bit 1) scheme.  This is the entry the reference's
// station drivers use (tests/test_aerobulk_buoy_series_oce.f90:452-487), with the real solar time of WL_COARE.
// Same per-cell physics (ab_physics.hpp) and the same LDS-staged, regrouped tiles as flux_kernel (ab_kernels.hip, ab_tile.hpp).
// these kernels keep four waves per SIMD on every instantiation (some would spill at five; next-tier rows, not tuned per kernel)
#include "ab_kernels.hpp"
#include "ab_physics.hpp"
#include "ab_launch.hpp"
#include "ab_tile.hpp"

namespace ab {

template <class R> struct TurbArgs {
    R *T_s, *q_s;
    const R *theta_zt, *q_zt, *U_zu, *qsw, *rad_lw, *slp, *lon;
    R *out[16];
    R *wl0, *wl1, *wl2, *wl3;
    long n;
    Heights<R> h;
    int nb_iter, wl_load, wl_store, isecday, dawn_uniform;
    int regroup, rounds;
};

// One cell of a TURB_* call: returns the eight values that go back to the caller's arrays (Cd Ch Ce t_zu q_zu Ubzu T_s q_s);
// the OPTIONAL outputs and the warm-layer state are written straight to global memory at cell k.
template <class R, int ALGO, int SKIN>
__device__ __forceinline__ void turb_cell(const TurbArgs<R> &a, const Heights<R> &hh, int nb_iter, long k, const CellIn<R> &in, R (&res)[8],
                                          lds_vptr<R> park = nullptr, int pstride = 0)
{
    constexpr bool WL = (SKIN & kSkinWL) != 0;
    R wl[4] = {R(0.), R(0.), R(0.), R(0.)};
    bool dawn = false;
    if (WL) {
        if (a.wl_load) {
            wl[0] = a.wl0[k];
            wl[1] = a.wl1[k];
            if (ALGO != 4) { wl[2] = a.wl2[k]; wl[3] = a.wl3[k]; }
        } else {  // COARE3Px_INIT mod_blk_coare3p6.f90:84-87 ; ECMWF_INIT mod_blk_ecmwf.f90:403-404
            wl[1] = (ALGO == 4) ? R(3.) : R(20.);
        }
        if (ALGO != 4) dawn = a.lon ? wl_coare_dawn<R>(a.lon[k], a.isecday) : (a.dawn_uniform != 0);
    }
    CellOut<R> o;
    if (ALGO == 1) turb_coare<R, false, SKIN, true>(hh, in, nb_iter, wl, dawn, o, park, pstride);
    else if (ALGO == 2) turb_coare<R, true, SKIN, true>(hh, in, nb_iter, wl, dawn, o, park, pstride);
    else if (ALGO == 3) turb_ncar<R, true>(hh, in, nb_iter, o);
    else if (ALGO == 4) turb_ecmwf<R, SKIN, true>(hh, in, nb_iter, wl, o);
    else turb_andreas<R, true>(hh, in, nb_iter, o);
    const R d[10] = {o.CdN, o.ChN, o.CeN, o.z0, o.us, o.L, o.UN10, o.dT_cs, o.dT_wl, o.Hz_wl};
#pragma unroll
    for (int i = 0; i < 10; ++i)
        if (a.out[6 + i]) a.out[6 + i][k] = d[i];
    if (WL && a.wl_store) {
        a.wl0[k] = wl[0];
        a.wl1[k] = wl[1];
        if (ALGO != 4) { a.wl2[k] = wl[2]; a.wl3[k] = wl[3]; }
    }
    res[0] = o.Cd; res[1] = o.Ch; res[2] = o.Ce; res[3] = o.t_zu; res[4] = o.q_zu; res[5] = o.Ubzu; res[6] = o.T_s; res[7] = o.q_s;
}

// Same four phases as flux_kernel (ab_kernels.hip): owners park the inputs of a tile in LDS and forecast each cell's
// branches, the tile is sorted, waves fetch groups of like-behaved cells, owners store.  NCAR keeps one lane per cell.
template <class R, int ALGO, int SKIN>
__global__ void __launch_bounds__(kBlock, (Tile<R, ALGO, (SKIN != 0), false, kTileFour>::kOcc)) turb_kernel(const TurbArgs<R> a)
{
    math_tables_init<R>();
    constexpr bool ANYSKIN = SKIN != 0;
    if (ALGO == 3) {
        const long k = (long)blockIdx.x * kBlock + threadIdx.x;
        if (k >= a.n) return;
        CellIn<R> in;
        in.sst = a.T_s[k]; in.theta_zt = a.theta_zt[k]; in.ssq = a.q_s[k]; in.q_zt = a.q_zt[k]; in.wnd = a.U_zu[k];
        in.slp = R(101000.); in.qsw = R(0.); in.rlw = R(0.);
        R res[8];
        turb_cell<R, ALGO, SKIN>(a, a.h, a.nb_iter, k, in, res);
#pragma unroll
        for (int i = 0; i < 6; ++i) a.out[i][k] = res[i];
        return;
    }
    using T = Tile<R, ALGO, ANYSKIN, false, kTileFour>;
    __shared__ R s_f[T::kFields][T::kCells];       // in: T_s theta q_s q_zt U [slp] [qsw rlw] ; out: the 8 results (6 without skin)
    __shared__ unsigned short s_inv[T::kCells];
    __shared__ unsigned s_cnt[kSortCounters], s_base[kSortCounters];
    __shared__ int s_next;
    const int tid = threadIdx.x;
    const int rounds = a.rounds;
    const long tile0 = (long)blockIdx.x * ((long)rounds * kBlock);
    if (tid == 0) s_next = 0;
    tile_sort_reset(s_cnt, tid);
    __syncthreads();                                             // counters zeroed before the first atomic of phase 1
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const int j = r * kBlock + tid;
        const long k = tile0 + j;
        int bkt = kBuckets - 1;
        if (k < a.n) {
            const R Ts = a.T_s[k], th = a.theta_zt[k], qs = a.q_s[k], q = a.q_zt[k], w = a.U_zu[k];
            s_f[0][j] = Ts; s_f[1][j] = th; s_f[2][j] = qs; s_f[3][j] = q; s_f[4][j] = w;
            R slp = R(101000.), qsw = R(0.), rlw = R(0.);
            if (ANYSKIN) {
                slp = a.slp[k]; qsw = a.qsw[k]; rlw = a.rad_lw[k];
                s_f[5][j] = slp; s_f[ANYSKIN ? 6 : 0][j] = qsw; s_f[ANYSKIN ? 7 : 0][j] = rlw;
            }
            bkt = 0;
            if (a.regroup) {
                const bool wll = (SKIN & kSkinWL) && a.wl_load;
                // the flux kernel's forecast: with the cool skin off no 0.25 K first guess, with the warm layer off no warm-layer bins
                bkt = forecast_bucket<ALGO, (SKIN & kSkinWL) != 0>((float)((SKIN & kSkinCS) && !(SKIN & kSkinWL) ? Ts - R(0.25) : Ts), (float)th,
                                                                (float)q, (float)w, 0.f, (float)slp, (float)qsw, (float)rlw, wll,
                                                                wll ? (float)a.wl0[k] : 0.f, (wll && ALGO != 4) ? (float)a.wl1[k] : 20.f);
            }
        }
        if (a.regroup) tile_sort_note(s_cnt, s_inv, j, bkt, tid);
    }
    __syncthreads();
    if (a.regroup) {
        tile_sort_place<T::kRounds>(s_cnt, s_base, s_inv, tid, rounds);
    } else {
        for (int r = 0; r < rounds; ++r) s_inv[r * kBlock + tid] = (unsigned short)(r * kBlock + tid);
    }
    __syncthreads();
    const int lane = tid & 63;
    const Heights<R> hh = detached(a.h);      // loop invariants out of their scalar-load tuples (ab_tile.hpp)
    int nb_iter = a.nb_iter;
    uniform_scalar(nb_iter);
#pragma unroll 1
    for (;;) {
        int g = 0;
        if (lane == 0) g = atomicAdd(&s_next, 1);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g >= rounds * (kBlock / 64)) break;
        const int j = s_inv[g * 64 + lane];
        const long k = tile0 + j;
        if (k >= a.n) continue;
        CellIn<R> in;
        in.sst = s_f[0][j]; in.theta_zt = s_f[1][j]; in.ssq = s_f[2][j]; in.q_zt = s_f[3][j]; in.wnd = s_f[4][j];
        in.slp = ANYSKIN ? s_f[5][j] : R(101000.);
        in.qsw = ANYSKIN ? s_f[ANYSKIN ? 6 : 0][j] : R(0.);
        in.rlw = ANYSKIN ? s_f[ANYSKIN ? 7 : 0][j] : R(0.);
        R res[8];
        // the cell's inputs are in registers now: its tile slots serve turb_coare as scratch words (ab_physics.hpp)
        turb_cell<R, ALGO, SKIN>(a, hh, nb_iter, k, in, res, ((SKIN & kSkinWL) && sizeof(R) == 8) ? (lds_vptr<R>)&s_f[0][j] : (lds_vptr<R>)nullptr, T::kCells);
#pragma unroll
        for (int i = 0; i < (ANYSKIN ? 8 : 6); ++i) s_f[i][j] = res[i];    // the slot is read by this lane only: reuse it
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const int j = r * kBlock + tid;
        const long k = tile0 + j;
        if (k >= a.n) break;
#pragma unroll
        for (int i = 0; i < 6; ++i) a.out[i][k] = s_f[i][j];
        if (ANYSKIN) {   // T_s, q_s are INTENT(inout): skin temperature and its saturation humidity on return
            a.T_s[k] = s_f[ANYSKIN ? 6 : 0][j];
            a.q_s[k] = s_f[ANYSKIN ? 7 : 0][j];
        }
    }
}

template <class R, int ALGO, int SKIN> static hipError_t launch_t(const TurbCall &c, hipStream_t stream)
{
    TurbArgs<R> a;
    a.T_s = (R *)c.T_s; a.q_s = (R *)c.q_s;
    a.theta_zt = (const R *)c.theta_zt; a.q_zt = (const R *)c.q_zt; a.U_zu = (const R *)c.U_zu;
    a.qsw = (const R *)c.qsw; a.rad_lw = (const R *)c.rad_lw; a.slp = (const R *)c.slp; a.lon = (const R *)c.lon;
    for (int i = 0; i < 16; ++i) a.out[i] = (R *)c.out[i];
    a.wl0 = (R *)c.wl[0]; a.wl1 = (R *)c.wl[1]; a.wl2 = (R *)c.wl[2]; a.wl3 = (R *)c.wl[3];
    a.n = c.n;
    a.h = make_heights<R>(c.zt, c.zu);
    a.nb_iter = c.nb_iter; a.wl_load = c.wl_load; a.wl_store = c.wl_store; a.isecday = c.isecday;
    a.dawn_uniform = dawn_at_lon0(c.isecday);
    a.regroup = c.regroup ? 1 : 0;
    a.rounds = tile_rounds(c.n, Tile<R, ALGO, (SKIN != 0), false, kTileFour>::kRounds, Tile<R, ALGO, (SKIN != 0), false, kTileFour>::kOcc);
    const long tile = (ALGO == 3) ? kBlock : (long)a.rounds * kBlock;
    const long nblk = (c.n + tile - 1) / tile;
    if (nblk <= 0) return hipSuccess;
    hipLaunchKernelGGL((turb_kernel<R, ALGO, SKIN>), dim3((unsigned)nblk), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

template <class R, int ALGO> static hipError_t launch_s(const TurbCall &c, hipStream_t s)
{
    switch (c.skin & 3) {
    case 0: return launch_t<R, ALGO, 0>(c, s);
    case 1: return launch_t<R, ALGO, kSkinCS>(c, s);
    case 2: return launch_t<R, ALGO, kSkinWL>(c, s);
    default: return launch_t<R, ALGO, kSkinBoth>(c, s);
    }
}

template <class R> static hipError_t launch_r(const TurbCall &c, hipStream_t s)
{
    switch (c.algo) {
    case 1: return launch_s<R, 1>(c, s);
    case 2: return launch_s<R, 2>(c, s);
    case 3: return (c.skin & 3) ? hipErrorInvalidValue : launch_t<R, 3, 0>(c, s);
    case 4: return launch_s<R, 4>(c, s);
    case 5: return (c.skin & 3) ? hipErrorInvalidValue : launch_t<R, 5, 0>(c, s);
    default: return hipErrorInvalidValue;
    }
}

// ---- turb_neutral_10m, mod_blk_neutral_10m.f90:33-209: CdN10, ChN10, CeN10, z0 from the neutral wind at 10 m.
// coare3p0 / coare3p6 / ecmwf: nb_iter passes of z0 = charn u*^2/g + 0.11 nu/u*, CdN10 = (kappa/ln(10/z0))^2 (:86-122); ncar:
// closed form (:166-180).  HBM-streaming: 8 B in, 32 B out per cell.
template <class R, int ALGO>
__global__ void __launch_bounds__(kBlock) neutral10_kernel(const R *U_N10, R *CdN10, R *ChN10, R *CeN10, R *pz0, long n, int nb_iter)
{
    math_tables_init<R>();
    using M = Mth<R>;
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const R rnu0_air = R(1.5E-5), log_zu = R(2.302585092994046);
    R cd, ch, ce, z0;
    if (ALGO == 3) {
        const R Ub = vmax(U_N10[k], R(0.5));
        cd = cd_n10_ncar(Ub);
        const R sq = M::sqrt(cd);
        ch = vmax(R(1.e-3) * sq * R(32.7), K<R>::Cx_min);      // pstab = 0 (:174)
        ce = vmax(R(1.e-3) * (R(34.6) * sq), K<R>::Cx_min);
        z0 = vmin(vmax(R(10.) * M::exp(-M::div(K<R>::vkarmn, sq)), R(0.0001)), R(0.1));
    } else {
        const R Ub = vmax(U_N10[k], R(0.1));
        cd = R(8.575E-5) * Ub + R(0.657E-3);
        const R charn = ALGO == 2 ? charn_coare3p6(Ub) : (ALGO == 1 ? charn_coare3p0(Ub) : R(0.018));
        R u_star = R(0.), lz0 = R(0.), dl = R(1.);
        z0 = R(0.);
#pragma unroll 1
        for (int jit = 1; jit <= nb_iter; ++jit) {
            u_star = Ub * M::sqrt(cd);
            z0 = charn * u_star * u_star * R(1. / 9.8) + M::div(R(0.11) * rnu0_air, u_star);
            lz0 = M::log(z0);
            dl = log_zu - lz0;
            cd = M::div(K<R>::vkarmn2, dl * dl);
        }
        R lz0t, lz0q;
        if (ALGO == 4) {
            const R lt = M::log(M::div(rnu0_air, u_star));
            lz0t = R(-0.916290731874155) + lt;      // ln 0.40
            lz0q = R(-0.4780358009429998) + lt;     // ln 0.62
        } else {
            const R lrr = lz0 + M::log(M::div(u_star, rnu0_air));   // ln(z0 u*/nu)
            lz0t = ALGO == 2 ? vmin(R(-8.740336742730447), R(-9.755067547417855) - R(0.72) * lrr)
                             : vmin(R(-9.115030192171858), R(-9.808177372731803) - R(0.6) * lrr);
            lz0q = lz0t;
        }
        ch = M::div(K<R>::vkarmn2, dl * (log_zu - lz0t));
        ce = M::div(K<R>::vkarmn2, dl * (log_zu - lz0q));
    }
    CdN10[k] = cd; ChN10[k] = ch; CeN10[k] = ce; pz0[k] = z0;
}

template <class R> static hipError_t launch_n10(int algo, int nb_iter, const void *U, void *cd, void *ch, void *ce, void *z0, long n,
                                                hipStream_t st)
{
    const long nblk = (n + kBlock - 1) / kBlock;
    if (nblk <= 0) return hipSuccess;
    const dim3 g((unsigned)nblk), b(kBlock);
    switch (algo) {
    case 1: hipLaunchKernelGGL((neutral10_kernel<R, 1>), g, b, 0, st, (const R *)U, (R *)cd, (R *)ch, (R *)ce, (R *)z0, n, nb_iter); break;
    case 2: hipLaunchKernelGGL((neutral10_kernel<R, 2>), g, b, 0, st, (const R *)U, (R *)cd, (R *)ch, (R *)ce, (R *)z0, n, nb_iter); break;
    case 3: hipLaunchKernelGGL((neutral10_kernel<R, 3>), g, b, 0, st, (const R *)U, (R *)cd, (R *)ch, (R *)ce, (R *)z0, n, nb_iter); break;
    case 4: hipLaunchKernelGGL((neutral10_kernel<R, 4>), g, b, 0, st, (const R *)U, (R *)cd, (R *)ch, (R *)ce, (R *)z0, n, nb_iter); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_neutral10(int algo, int nb_iter, const void *U_N10, void *CdN10, void *ChN10, void *CeN10, void *z0, long n,
                            int f32, hipStream_t stream)
{
    return f32 ? launch_n10<float>(algo, nb_iter, U_N10, CdN10, ChN10, CeN10, z0, n, stream)
               : launch_n10<double>(algo, nb_iter, U_N10, CdN10, ChN10, CeN10, z0, n, stream);
}

hipError_t launch_turb(const TurbCall &c, hipStream_t stream)
{
    return c.f32 ? launch_r<float>(c, stream) : launch_r<double>(c, stream);
}

}  // namespace ab
