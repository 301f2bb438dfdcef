// ab_ice_kernels.hip — the sea-ice bulk algorithms TURB_ICE_NEMO / AN05 / LU12 / LG15 (= LG15_IO over ice) / EASY as HIP kernels (gfx950).
//
// ice_kernel<R,ALGO>: coalesced streaming of 5-6 input and 6-13 output fields, the nb_iter iteration in registers
// (ab_physics_ice.hpp).  Algorithmic bytes per cell, fp64: 5 in + 6 out = 88 B (+ 8 B ice concentration for
// LU12, + 8 B per OPTIONAL output).  NEMO / LU12 are HBM-bound (no iteration), AN05 / LG15 VALU-bound like the open-ocean
// algorithms.
// these kernels keep four waves per SIMD on every instantiation (some would spill at five; next-tier rows, not tuned per kernel)
#include "ab_kernels.hpp"
#include "ab_physics_ice.hpp"
#include "ab_launch.hpp"
#include "ab_tile.hpp"

namespace ab {

template <class R> struct IceArgs {
    const R *Ts_i, *theta_zt, *qs_i, *q_zt, *U_zu, *frice;
    R *out[14];
    long n;
    Heights<R> h;
    int nb_iter, regroup, rounds;
    R cxn[3];   // TURB_ICE_EASY: prescribed CdN, ChN, CeN
};

template <class R, int ALGO>
__device__ __forceinline__ void ice_cell(const IceArgs<R> &a, const Heights<R> &hh, int nb_iter, const IceIn<R> &in, IceOut<R> &o)
{
    if (ALGO == 1) {
        turb_ice_const<R>(hh, in, KIce<R>::rCd_ice, o);
    } else if (ALGO == 2) {
        turb_ice_an05<R>(hh, in, nb_iter, o);
    } else if (ALGO == 3) {   // skin drag of z0 = 0.69 mm + form drag (mod_blk_ice_lu12.f90:156-160)
        const R zi = Mth<R>::rcp(hh.log_zu - R(-7.278818960372969));
        turb_ice_const<R>(hh, in, K<R>::vkarmn2 * zi * zi + cdn10_f_lu13<R>(in.frice), o);
    } else if (ALGO == 4) {
        turb_ice_lg15<R>(hh, in, a.frice[a.n - 1], nb_iter, o);   // wave-uniform load of the last cell's concentration
    } else {
        turb_ice_easy<R>(hh, in, a.cxn[0], a.cxn[1], a.cxn[2], nb_iter, o);
    }
}

// NEMO / LU12 (no iteration, HBM-bound): one lane per cell.  AN05 / LG15 / EASY (nb_iter iterations with stable / unstable branches):
// the LDS-staged, regrouped tiles of flux_kernel (ab_tile.hpp); the bucket is the sign of the air-ice virtual temperature
// difference in four bins.
template <class R, int ALGO> __global__ void __launch_bounds__(kBlock, (Tile<R, ALGO, false, false, kTileFour>::kOcc)) ice_kernel(const IceArgs<R> a)
{
    math_tables_init<R>();
    if (ALGO == 1 || ALGO == 3) {
        const long k = (long)blockIdx.x * kBlock + threadIdx.x;
        if (k >= a.n) return;
        IceIn<R> in;
        in.Ts_i = a.Ts_i[k]; in.theta_zt = a.theta_zt[k]; in.qs_i = a.qs_i[k]; in.q_zt = a.q_zt[k]; in.wnd = a.U_zu[k];
        in.frice = (ALGO == 3) ? a.frice[k] : R(0.);
        IceOut<R> o;
        ice_cell<R, ALGO>(a, a.h, a.nb_iter, in, o);
        const R d[13] = {o.Cd, o.Ch, o.Ce, o.t_zu, o.q_zu, o.Ub, o.CdN, o.ChN, o.CeN, o.z0, o.us, o.L, o.UN10};
#pragma unroll
        for (int i = 0; i < 13; ++i)
            if (a.out[i]) a.out[i][k] = d[i];
        return;
    }
    using T = Tile<R, ALGO, false, false, kTileFour>;               // 6 fields: in Ts theta qs q U ; out the six mandatory results
    __shared__ R s_f[T::kFields][T::kCells];
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
            const R Ts = a.Ts_i[k], th = a.theta_zt[k], qs = a.qs_i[k], q = a.q_zt[k], w = a.U_zu[k];
            s_f[0][j] = Ts; s_f[1][j] = th; s_f[2][j] = qs; s_f[3][j] = q; s_f[4][j] = w;
            bkt = 0;
            if (a.regroup) {
                const float dthv = (float)th * (1.f + 0.608f * (float)q) - (float)Ts * (1.f + 0.608f * (float)qs);
                bkt = 4 * (dthv < -0.3f ? 0 : (dthv < 0.f ? 1 : (dthv < 0.3f ? 2 : 3)));
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
        IceIn<R> in;
        in.Ts_i = s_f[0][j]; in.theta_zt = s_f[1][j]; in.qs_i = s_f[2][j]; in.q_zt = s_f[3][j]; in.wnd = s_f[4][j];
        in.frice = R(0.);
        IceOut<R> o;
        ice_cell<R, ALGO>(a, hh, nb_iter, in, o);
        const R d7[7] = {o.CdN, o.ChN, o.CeN, o.z0, o.us, o.L, o.UN10};
#pragma unroll
        for (int i = 0; i < 7; ++i)
            if (a.out[6 + i]) a.out[6 + i][k] = d7[i];           // OPTIONAL outputs: straight to global memory
        if (ALGO == 4 && a.out[13]) a.out[13][k] = o.CdN_frm;
        s_f[0][j] = o.Cd; s_f[1][j] = o.Ch; s_f[2][j] = o.Ce; s_f[3][j] = o.t_zu; s_f[4][j] = o.q_zu; s_f[5][j] = o.Ub;
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const int j = r * kBlock + tid;
        const long k = tile0 + j;
        if (k >= a.n) break;
#pragma unroll
        for (int i = 0; i < 6; ++i) a.out[i][k] = s_f[i][j];
    }
}

template <class R, int ALGO> static hipError_t launch_t(const IceCall &c, hipStream_t stream)
{
    IceArgs<R> a;
    a.Ts_i = (const R *)c.Ts_i; a.theta_zt = (const R *)c.theta_zt; a.qs_i = (const R *)c.qs_i;
    a.q_zt = (const R *)c.q_zt; a.U_zu = (const R *)c.U_zu; a.frice = (const R *)c.frice;
    for (int i = 0; i < 14; ++i) a.out[i] = (R *)c.out[i];
    a.n = c.n;
    a.h = make_heights<R>(c.zt, c.zu);
    a.nb_iter = c.nb_iter;
    for (int i = 0; i < 3; ++i) a.cxn[i] = (R)c.cxn[i];
    a.regroup = 1;
    a.rounds = tile_rounds(c.n, Tile<R, ALGO, false, false, kTileFour>::kRounds, Tile<R, ALGO, false, false, kTileFour>::kOcc);
    const long tile = (ALGO == 1 || ALGO == 3) ? kBlock : (long)a.rounds * kBlock;
    const long nblk = (c.n + tile - 1) / tile;
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
    case 5: return launch_t<R, 5>(c, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_turb_ice(const IceCall &c, hipStream_t stream)
{
    return c.f32 ? launch_r<float>(c, stream) : launch_r<double>(c, stream);
}

}  // namespace ab
