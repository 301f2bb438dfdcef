// ab_kernels.hip — HIP kernels of the bulk air-sea flux engine, written for gfx950 (MI355X).
//
// flux_kernel<R,ALGO,SKIN,DIAG>: the whole of aerobulk_compute() (mod_aerobulk_compute.f90:22-213)
// fused into ONE pointwise kernel: humidity conversion, scalar wind, q_sat(SST), theta(zt),
// TURB_<algo> with its nb_iter Monin-Obukhov iterations and optional cool-skin/warm-layer,
// BULK_FORMULA and the stress vector.  One lane = one cell at a time; a wave reads 64 consecutive cells of
// each field with one coalesced 8-byte-per-lane load (512 B per wave-instruction) and writes the
// outputs the same way; none of the reference's 15 (Ni,Nj) temporaries exists.  The kernel is
// bound by fp64 VALU throughput, not HBM.  LDS stages the input tile of a block so that
// the cells can be dealt to the waves by kind (lane regrouping, below): same arithmetic per cell, less
// SIMT divergence.
#include "ab_kernels.hpp"
#define AB_PSI_LDS_TABLES 1   // the tiled fp64 kernels of this file keep two psi functions as piecewise tables in LDS (ab_physics.hpp)
#include "ab_physics.hpp"
#include "ab_launch.hpp"
#include "ab_tile.hpp"

namespace ab {

// R: arithmetic type of the cell; S: element type of the caller's arrays and of the warm-layer planes (S = float with
// R = double is the AB_F32_STORAGE session: half the HBM traffic and footprint, full fp64 arithmetic)
template <class R, class S = R> struct FluxArgs {
    const S *sst, *t_zt, *hum, *u, *v, *slp, *rad_sw, *rad_lw, *lon;
    S *ql, *qh, *tau_x, *tau_y, *evap, *t_s;
    S *wl0, *wl1, *wl2, *wl3;
    int *flags;
    long n;
    Heights<R> h;
    int nb_iter, hum_type, wl_load, wl_store, isecday, dawn_uniform;
    int regroup;  // sort the tile's cells into like-behaved waves (see flux_kernel)
    int rounds;   // tile = rounds*256 cells (<= Tile<R,ALGO,SKIN>::kRounds)
    long nfull;   // tiles [0, nfull) have `rounds` rounds; the ntail tiles behind them, handed out last, one round each (launch_t)
    long ntail;
    unsigned long long wl_live;   // bit jit-1 set where MOD(nb_iter, jit) == 0: the iterations whose WL_COARE call is live (0: nb_iter > 64, the kernel divides)
    int *queue;   // flux_kernel_cu: three tile counters in device memory (zero between launches)
};

// optional per-cell diagnostics of TURB_* (ab_session_set_diagnostics); read only by the DIAG instantiations
template <class S> struct DiagArgs {
    S *p[16];   // Cd Ch Ce t_zu q_zu Ubzu | CdN ChN CeN z0 u_star L UN10 | dT_cs dT_wl Hz_wl ; nullptr = not wanted
};

// fields are read once and written once
#ifdef AB_NT_FIELDS
template <class T> __device__ __forceinline__ T ldnt(const T *p) { return __builtin_nontemporal_load(p); }
template <class T> __device__ __forceinline__ void stnt(T *p, T v) { __builtin_nontemporal_store(v, p); }
#else
template <class T> __device__ __forceinline__ T ldnt(const T *p) { return *p; }
template <class T> __device__ __forceinline__ void stnt(T *p, T v) { *p = v; }
#endif

// ---- lane regrouping ---------------------------------------------------------------------------------------------
// The iteration takes divergent paths per cell: stable / unstable psi functions, warm layer gaining heat / idle.  On
// spatially incoherent input (the quasi-random benchmark fields are the worst case) every wave holds all kinds of cells and
// executes every path.  A block therefore owns a TILE of consecutive cells (512 with the skin schemes, 768 without; 1024 /
// 1280 in fp32), and works in four phases:
//   1. owners (thread = cell, natural order, coalesced loads): pre-processing of aerobulk_compute, an fp32 forecast of the
//      two predicates -> one of 16 buckets, pre-processed inputs parked in LDS (field-major, <= 39 KB);
//   2. counting sort of the tile's cell indices by bucket (ranks from LDS atomics while phase 1 runs, ab_tile.hpp);
//   3. waves fetch groups of 64 like-behaved cells (dynamic queue, most work first come) and run TURB_* + BULK_FORMULA;
//      results go back to the cell's LDS slot;
//   4. owners store the results (coalesced).
// The arithmetic per cell, hence every output bit, is the same as in natural order (tests/test_gpu_regroup.py); global
// loads and stores stay coalesced and each field is still read once and written once.
// One cell from its pre-processed inputs to the six outputs of aerobulk_compute: TURB_<algo> (mod_aerobulk_compute.f90
// :129-176), BULK_FORMULA and the stress vector (:184-194).  k: global cell index (warm-layer state, diagnostics, longitude).
// A: anchor type of SST, theta, q (ab_physics.hpp, "ANCHORS"): R, or double with R = float in the mixed mode.
template <class R, int ALGO, bool SKIN, bool DIAG, bool TILED = false, class S = R, class A = R, int LTABS = 1>
__device__ __forceinline__ void compute_cell(const FluxArgs<R, S> &a, const DiagArgs<S> &dg, const Heights<R> &hh, int nb_iter, long k, A sst,
                                             A theta_zt, A q_zt, R uu, R vv, R slp, R qsw, R rlw, R &QL, R &QH, R &tx, R &ty,
                                             R &zEvap, A &T_s, lds_cvptr<R> pu = nullptr, lds_cvptr<R> pv = nullptr,
                                             lds_vptr<R> park = nullptr, int pstride = 0)
{
    using M = Mth<R>;
    constexpr bool kMixed = !std::is_same<R, A>::value;
    CellIn<R, A> in;
    in.sst = sst;
    in.theta_zt = theta_zt;
    in.q_zt = q_zt;
    in.slp = slp;
    in.wnd = M::sqrt(uu * uu + vv * vv);                                       // :111
    // :114 (e_sat from its LDS table in the tiled fp64 kernels with the skin schemes, and in every tiled mixed kernel)
    in.ssq = rounded(K<A>::rdct_qsat_salt * q_sat<A, (TILED && (SKIN || kMixed) && kPsiTabDefault)>(sst, A(slp)));
    in.qsw = qsw;
    in.rlw = rlw;

    R wl[4] = {R(0.), R(0.), R(0.), R(0.)};
    bool dawn = false;
    if (SKIN) {
        if (a.wl_load) {
            wl[0] = a.wl0[k];
            wl[1] = a.wl1[k];
            if (ALGO != 4) { wl[2] = a.wl2[k]; wl[3] = a.wl3[k]; }
        } else {  // COARE3Px_INIT mod_blk_coare3p6.f90:84-87 ; ECMWF_INIT mod_blk_ecmwf.f90:403-404
            wl[1] = (ALGO == 4) ? R(3.) : R(20.);
        }
        if (ALGO != 4) dawn = a.lon ? wl_coare_dawn<R>((R)a.lon[k], a.isecday) : (a.dawn_uniform != 0);
    }

    CellOut<R, A> o;
    constexpr int kSkin = SKIN ? kSkinBoth : 0;   // aerobulk_compute: cool skin and warm layer together
    constexpr int kCsgLds = (TILED && sizeof(R) == 8) ? LTABS : 0;   // flux_kernel filled COARE's LDS extras (the cool skin's table with the skin schemes, the psi tables without); 2: flux_kernel_cu's CU-wide psi and WL tables too
    if (ALGO == 1) turb_coare<R, false, kSkin, DIAG, A, kCsgLds>(hh, in, nb_iter, wl, dawn, o, park, pstride, a.wl_live);
    else if (ALGO == 2) turb_coare<R, true, kSkin, DIAG, A, kCsgLds>(hh, in, nb_iter, wl, dawn, o, park, pstride, a.wl_live);
    else if (ALGO == 3) turb_ncar<R, DIAG, A, (sizeof(R) == 8 && kPsiTabDefault)>(hh, in, nb_iter, o);   // flux_kernel's direct path filled the pair of psi tables
    else if (ALGO == 4) turb_ecmwf<R, kSkin, DIAG, A>(hh, in, nb_iter, wl, o);
    else turb_andreas<R, DIAG, A, (TILED && sizeof(R) == 8)>(hh, in, nb_iter, o);   // flux_kernel filled psi_m's stable-side table
    if (DIAG) {
        const R d[16] = {o.Cd, o.Ch, o.Ce, R(o.t_zu), R(o.q_zu), o.Ubzu, o.CdN, o.ChN, o.CeN, o.z0, o.us, o.L, o.UN10,
                         o.dT_cs, o.dT_wl, o.Hz_wl};
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (dg.p[i]) dg.p[i][k] = (S)d[i];
    }
    if (SKIN && a.wl_store) {
        a.wl0[k] = (S)wl[0];
        a.wl1[k] = (S)wl[1];
        if (ALGO != 4) { a.wl2[k] = (S)wl[2]; a.wl3[k] = (S)wl[3]; }
    }

    R zTaum;
    bulk_formula<R, A>(hh.zu, o.T_s, o.q_s, o.t_zu, o.q_zu, o.Cd, o.Ch, o.Ce, in.wnd, o.Ubzu, slp, zTaum, QH, QL, zEvap);
    if (zTaum > R(10.)) atomicOr(a.flags, 1);                                  // mod_phymbl.f90:1250-1253
    tx = R(0.);
    ty = R(0.);
    if (in.wnd > R(1.E-3)) {
        const R s = M::div(zTaum, in.wnd);
        // tiled path: u and v are still in the cell's LDS slots; re-reading them here keeps 4 VGPRs free across the iteration
        if constexpr (TILED) {
            tx = s * *pu;
            ty = s * *pv;
        } else {
            tx = s * uu;
            ty = s * vv;
        }
    }
    T_s = o.T_s;
}

// The kernel's arguments re-read from the kernarg segment through a pointer the compiler cannot see through: the scalar loads stay at the
// phase that uses them instead of being hoisted to the kernel's entry and kept alive (or spilled to VGPR lanes) across the iteration.
// What the offsets below assume, enforced: the explicit arguments start at offset 0 of the segment, FluxArgs first, DiagArgs behind it at
// the next multiple of its alignment (<= 8: both hold pointers, longs and doubles only) — the layout the HIP ABI gives two by-value
// structs (code object v5: no reordering, no preloaded arguments for these kernels).  launch_t checks the arithmetic on the host once.
template <class R, class S> struct KernargLayout {
    static_assert(alignof(FluxArgs<R, S>) <= 8 && alignof(DiagArgs<S>) <= 8, "kernarg_at: offsets are multiples of 8");
    static_assert(std::is_trivially_copyable<FluxArgs<R, S>>::value && std::is_trivially_copyable<DiagArgs<S>>::value, "by-value kernel arguments");
    static constexpr unsigned diag_offset = (unsigned)((sizeof(FluxArgs<R, S>) + alignof(DiagArgs<S>) - 1) / alignof(DiagArgs<S>) * alignof(DiagArgs<S>));
    static_assert(diag_offset == (unsigned)((sizeof(FluxArgs<R, S>) + 7) & ~(size_t)7), "DiagArgs follows FluxArgs at the next multiple of 8");
};
template <class T> __device__ __forceinline__ const T &kernarg_at(unsigned off)
{
    const __attribute__((address_space(4))) char *p = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const T *)(p + off);
}

// ---- the four phases of a tile, shared by flux_kernel (a 256-thread block owns the tile) and flux_kernel_cu (a team of a CU-wide workgroup
// does).  s_f: the tile's field rows [kFields][CELLS] in LDS; tid: the thread's index among the 256 that own the tile.
template <class R> struct RawCell { R sst, t_zt, hum, uu, vv, slp, rsw, rlw; };
// tile t of the launch: `rounds` rounds of 256 cells; the LAST tiles of the grid are one-round tiles: the chip drains over the life of a
// short tile instead of a long one (launch_t)
template <class R, class S> __device__ __forceinline__ void tile_of(const FluxArgs<R, S> &a, long t, long &t0, int &rr)
{
    rr = a.rounds;
    t0 = t * ((long)rr * kBlock);
    if (t >= a.nfull) { t0 = a.nfull * ((long)rr * kBlock) + (t - a.nfull) * kBlock; rr = 1; }
}
// owners load their cells (coalesced); streamed once: non-temporal, so that the fields do not push the piecewise tables out of the L1
template <class R, class S, bool SKIN> __device__ __forceinline__ RawCell<R> tile_fetch(const FluxArgs<R, S> &a, int tid, long tile0, int rounds, int r)
{
    RawCell<R> w{R(290.), R(290.), R(0.01), R(1.), R(1.), R(101000.), R(0.), R(0.)};
    const long k = tile0 + r * kBlock + tid;
    if (r < rounds && k < a.n) {
        w.sst = (R)ldnt(a.sst + k); w.t_zt = (R)ldnt(a.t_zt + k); w.hum = (R)ldnt(a.hum + k); w.uu = (R)ldnt(a.u + k); w.vv = (R)ldnt(a.v + k);
        w.slp = (R)ldnt(a.slp + k);
        if (SKIN) { w.rsw = (R)ldnt(a.rad_sw + k); w.rlw = (R)ldnt(a.rad_lw + k); }
    }
    return w;
}
// phase 1: pre-processing mod_aerobulk_compute.f90:99-126 of the owners' cells, parked in LDS; the forecast of each cell's bucket.  The
// loads of round r + 1 are in flight while round r is pre-processed (`nxt`: those of round 0, issued by the caller)
template <class R, int ALGO, bool SKIN, class S, class A, int CELLS>
__device__ __forceinline__ void tile_phase1(const FluxArgs<R, S> &a, int tid, long tile0, int rounds, RawCell<R> nxt, R (*s_f)[CELLS], unsigned *s_cnt,
                                            unsigned short *s_inv)
{
    constexpr bool kMixed = !std::is_same<R, A>::value;
    constexpr int kThLo = (SKIN ? 8 : 6);      // mixed: one more row at the end, the low part of theta
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const int j = r * kBlock + tid;
        const long k = tile0 + j;
        const RawCell<R> w = nxt;
        nxt = tile_fetch<R, S, SKIN>(a, tid, tile0, rounds, r + 1);
        int bkt = kBuckets - 1;                                  // cells beyond n: last bucket, skipped in phase 3
        if (k < a.n) {
            const R sst = w.sst, t_zt = w.t_zt, hum = w.hum, uu = w.uu, vv = w.vv, slp = w.slp;
            constexpr bool kEsatTab = (SKIN || kMixed) && kPsiTabDefault;
            A q_zt;
            if (a.hum_type == 0) q_zt = A(hum);                                     // 'sh'
            else if (a.hum_type == 1) q_zt = q_air_dp<A, kEsatTab>(A(hum), A(vmax(slp, R(50000.))));   // 'dp' :103
            else q_zt = q_air_rh<A, kEsatTab>(A(hum), A(t_zt), A(vmax(slp, R(50000.))));              // 'rh' :105
            const A theta = theta_from_z_p0_t_q<A, kEsatTab>(A(a.h.zt), A(slp), A(t_zt), q_zt);       // :118
            s_f[0][j] = sst; s_f[1][j] = R(theta); s_f[2][j] = R(q_zt); s_f[3][j] = uu; s_f[4][j] = vv; s_f[5][j] = slp;
            if constexpr (kMixed) s_f[kThLo][j] = R(theta - A(R(theta)));
            R qsw = R(0.), rlw = R(0.);
            if (SKIN) {
                qsw = (R(1.) - K<R>::roce_alb0) * w.rsw;                            // :135,146,161
                rlw = w.rlw;
                s_f[SKIN ? 6 : 0][j] = qsw; s_f[SKIN ? 7 : 0][j] = rlw;
            }
            if (a.regroup) {
                const bool wll = SKIN && a.wl_load;
                bkt = forecast_bucket<ALGO, SKIN>((float)sst, (float)theta, (float)q_zt, (float)uu, (float)vv, (float)slp,
                                                  (float)qsw, (float)rlw, wll, wll ? (float)a.wl0[k] : 0.f,
                                                  (wll && ALGO != 4) ? (float)a.wl1[k] : 20.f);
            } else {
                bkt = 0;
            }
        }
        if (a.regroup) tile_sort_note(s_cnt, s_inv, j, bkt, tid);
    }
}
// phase 2: who computes which cell (the counting sort's placement; the counters are zeroed again for the owners' next tile)
template <int MAXROUNDS, class R, class S, class Sync>
__device__ __forceinline__ void tile_phase2(const FluxArgs<R, S> &a, int tid, int rounds, unsigned *s_cnt, unsigned *s_base, unsigned short *s_inv, Sync sync)
{
    if (a.regroup) {
        tile_sort_place<MAXROUNDS>(s_cnt, s_base, s_inv, tid, rounds, sync);
        tile_sort_reset(s_cnt, tid);   // (behind the barrier inside tile_sort_place: the counts have been read)
    } else {
        for (int r = 0; r < rounds; ++r) s_inv[r * kBlock + tid] = (unsigned short)(r * kBlock + tid);
    }
}
// phase 3: groups of 64 sorted cells, fetched from the tile's queue; the results go back to the cell's LDS slot
template <class R, int ALGO, bool SKIN, bool DIAG, class S, class A, int LTABS, int CELLS>
__device__ __forceinline__ void tile_phase3(const FluxArgs<R, S> &a, const DiagArgs<S> &dg, int tid, long tile0, int rounds, R (*s_f)[CELLS],
                                            const unsigned short *s_inv, int *s_next)
{
    constexpr bool kMixed = !std::is_same<R, A>::value;
    constexpr int kThLo = (SKIN ? 8 : 6);
    const int lane = tid & 63;
    const Heights<R> hh = detached(a.h);      // loop invariants out of their scalar-load tuples (ab_tile.hpp)
    int nb_iter = a.nb_iter;
    uniform_scalar(nb_iter);
#pragma unroll 1
    for (;;) {
        int g = 0;
        if (lane == 0) g = atomicAdd(s_next, 1);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g >= rounds * (kBlock / 64)) break;
        const int j = s_inv[g * 64 + lane];
        const long k = tile0 + j;
        if (k >= a.n) continue;

        R QL, QH, tx, ty, zEvap;
        A T_s;
        A theta = A(s_f[1][j]);
        if constexpr (kMixed) theta = theta + A(s_f[kThLo][j]);
        compute_cell<R, ALGO, SKIN, DIAG, true, S, A, LTABS>(a, dg, hh, nb_iter, k, A(s_f[0][j]), theta, A(s_f[2][j]), s_f[3][j], s_f[4][j], s_f[5][j],
                                          SKIN ? s_f[SKIN ? 6 : 0][j] : R(0.), SKIN ? s_f[SKIN ? 7 : 0][j] : R(0.), QL, QH, tx,
                                          ty, zEvap, T_s, (lds_cvptr<R>)&s_f[3][j], (lds_cvptr<R>)&s_f[4][j],
                                          // rows 0-2, 5, 6 (sst theta q slp qsw) are in registers by now: scratch words for turb_coare
                                          (SKIN && sizeof(R) == 8) ? (lds_vptr<R>)&s_f[0][j] : (lds_vptr<R>)nullptr, CELLS);
        // the cell's LDS slot is read by this lane only: reuse it for the results
        s_f[0][j] = QL; s_f[1][j] = QH; s_f[2][j] = tx; s_f[3][j] = ty; s_f[4][j] = zEvap; s_f[5][j] = R(T_s);
    }
}
// phase 4: owners store (coalesced)
template <class R, class S, int CELLS> __device__ __forceinline__ void tile_phase4(const FluxArgs<R, S> &a, int tid, long tile0, int rounds, R (*s_f)[CELLS])
{
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const int j = r * kBlock + tid;
        const long k = tile0 + j;
        if (k >= a.n) break;
        stnt(a.ql + k, (S)s_f[0][j]);
        stnt(a.qh + k, (S)s_f[1][j]);
        stnt(a.tau_x + k, (S)s_f[2][j]);
        stnt(a.tau_y + k, (S)s_f[3][j]);
        if (a.evap) stnt(a.evap + k, (S)s_f[4][j]);                                // :208
        if (a.t_s) stnt(a.t_s + k, (S)s_f[5][j]);                                  // :206
    }
}

// A: the anchor type (ab_physics.hpp, "ANCHORS").  A = R for the fp64 and fp32 sessions; AB_F32_MIXED is <R = float, S = float,
// A = double>: fp32 arrays and fp32 hardware transcendentals, with SST, theta, q, T_s, q_s, their differences and q_sat in fp64.
template <class R, int ALGO, bool SKIN, bool DIAG, class S = R, class A = R>
// (the DIAG instantiations carry sixteen more live values: four waves per SIMD, on the tiles sized for Tile::kOcc — three for fp64 with the
// skin schemes, whose lean kernels fill their 128 registers)
__global__ void __launch_bounds__(kBlock, (DIAG ? ((sizeof(R) == 8 && SKIN) ? 3 : AB_WAVES_PER_EU) : Tile<R, ALGO, SKIN, !std::is_same<R, A>::value>::kOcc)) flux_kernel(const FluxArgs<R, S> a_in, const DiagArgs<S> dg_in)
{
    constexpr bool kMixed = !std::is_same<R, A>::value;
    if (ALGO == 3) {   // NCAR: the cheapest iteration, little divergence: the tile machinery costs more than it saves
        const FluxArgs<R, S> &a = a_in;
        const DiagArgs<S> &dg = dg_in;
        const long k = (long)blockIdx.x * kBlock + threadIdx.x;
        const bool live = k < a.n;
        // the cell's loads are in flight while the block fills its math tables
        R slp = R(101000.), t_zt = R(290.), hum = R(0.01), sst = R(290.), uu = R(1.), vv = R(1.);
        if (live) { slp = (R)a.slp[k]; t_zt = (R)a.t_zt[k]; hum = (R)a.hum[k]; sst = (R)a.sst[k]; uu = (R)a.u[k]; vv = (R)a.v[k]; }
        if constexpr (sizeof(R) == 8) psi_tables_fill<false>();   // Kansas psi_m / psi_h (before the barrier of math_tables_init)
        math_tables_init<A>();
        if (!live) return;
        A q_zt;
        if (a.hum_type == 0) q_zt = A(hum);
        else if (a.hum_type == 1) q_zt = q_air_dp<A, false>(A(hum), A(vmax(slp, R(50000.))));
        else q_zt = q_air_rh<A, false>(A(hum), A(t_zt), A(vmax(slp, R(50000.))));
        R QL, QH, tx, ty, zEvap;
        A T_s;
        compute_cell<R, ALGO, SKIN, DIAG, false, S, A>(a, dg, a.h, a.nb_iter, k, A(sst), theta_from_z_p0_t_q<A, false>(A(a.h.zt), A(slp), A(t_zt), q_zt), q_zt,
                                                       uu, vv, slp, R(0.), R(0.), QL, QH, tx, ty, zEvap, T_s);
        a.ql[k] = (S)QL;
        a.qh[k] = (S)QH;
        a.tau_x[k] = (S)tx;
        a.tau_y[k] = (S)ty;
        if (a.evap) a.evap[k] = (S)zEvap;
        if (a.t_s) a.t_s[k] = (S)T_s;
        return;
    }
    // Every phase re-reads the kernel arguments it needs from the kernarg segment (kernarg_at: scalar loads from constant memory, next to
    // their use).  Loaded once at the kernel's entry — what the compiler does with by-value arguments — the 19 pointers, the heights and
    // the switches are 80 SGPRs alive across the iteration, where the polynomials' coefficients push them out to VGPR lanes: the headline
    // kernel carried 107 v_readlane / 21 v_writelane (29 SGPRs spilled), none now; -1.0 % (COARE3p6 + skin) ... -2.9 % (NCAR is not
    // touched: noise) on 4320x3600, same-box, six order-balanced passes, bit-identical (profiles/r4_notes.md §2).
    constexpr unsigned kDgOff = KernargLayout<R, S>::diag_offset;
#define AB_ARGS const FluxArgs<R, S> &a = kernarg_at<FluxArgs<R, S>>(0)
#define AB_DIAGS const DiagArgs<S> &dg = kernarg_at<DiagArgs<S>>(kDgOff)
    using T = Tile<R, ALGO, SKIN, kMixed>;
    // field rows: sst theta q_zt u v slp [qsw rlw]; mixed: one more row at the end, the low part of theta (theta is an fp64 anchor
    // parked as a float pair; sst and q_zt are fp32 numbers anyway, exactly in 'sh' mode and to half an fp32 ulp otherwise)
    __shared__ R s_f[T::kFields][T::kCells];
    __shared__ unsigned short s_inv[T::kCells];
    __shared__ unsigned s_cnt[kSortCounters], s_base[kSortCounters];
    __shared__ int s_next;
    const int tid = threadIdx.x;
    long tile0;
    int rounds;
    RawCell<R> nxt;
    {
        AB_ARGS;
        // the re-read arguments against the by-value ones (one thread of the launch): a toolchain that lays the kernarg segment out
        // differently shows as bit 1 of the session's flag word (ab_session_check: AB_ERR_HIP), not as garbage pointers
        if (blockIdx.x == 0 && tid == 0 && (a.n != a_in.n || a.ql != a_in.ql || kernarg_at<DiagArgs<S>>(kDgOff).p[15] != dg_in.p[15])) atomicOr(a_in.flags, 2);
        tile_of(a, (long)blockIdx.x, tile0, rounds);     // <= T::kRounds; fewer on small grids so that every CU gets blocks
        // the loads of round 0 are in flight while the block fills its math tables (a block starts with nothing else to hide that latency behind)
        nxt = tile_fetch<R, S, SKIN>(a, tid, tile0, rounds, 0);
    }
    if (tid == 0) s_next = 0;
    tile_sort_reset(s_cnt, tid);
    // (before the barrier of math_tables_init) fp64: COARE + skin reads its psi tables through L1 (ab_gtables.hpp: indexed by the bits of
    // their argument, no logarithm); COARE without the skin schemes, ECMWF and ANDREAS keep theirs in LDS; the e_sat table with the skin schemes
    if constexpr (sizeof(R) == 8) {
        if constexpr (ALGO == 1 || ALGO == 2) {
            if (SKIN) {
                esat_table_fill(); csg_table_fill();
#ifdef AB_PSI_NOBITS
                psi_coare_lds_fill<false>();
#endif
            } else {
#ifdef AB_NOSKIN_PSI_LOGTAB
                psi_coare_lds_fill();
#else
                psi_bits_lds_fill();
#endif
            }
        }
        else { psi_tables_fill<SKIN>(); if constexpr (ALGO == 5) andreas_stab_fill(); }
    }
    else psi_tables_fill32();                             // (before the barrier below)
    if constexpr (kMixed) esat_table_fill();              // q_sat of the mixed mode is the fp64 one, through its LDS table
    math_tables_init<A>();
    if (sizeof(A) != 8) __syncthreads();                  // (fp64: the barrier of math_tables_init) counters zeroed before phase 1
    { AB_ARGS; tile_phase1<R, ALGO, SKIN, S, A>(a, tid, tile0, rounds, nxt, s_f, s_cnt, s_inv); }
    __syncthreads();
    { AB_ARGS; tile_phase2<T::kRounds>(a, tid, rounds, s_cnt, s_base, s_inv, BlockSync()); }
    __syncthreads();
    { AB_ARGS; AB_DIAGS; tile_phase3<R, ALGO, SKIN, DIAG, S, A, 1>(a, dg, tid, tile0, rounds, s_f, s_inv, &s_next); }
    __syncthreads();
    { AB_ARGS; tile_phase4<R, S>(a, tid, tile0, rounds, s_f); }
#undef AB_ARGS
#undef AB_DIAGS
}

// ---- one workgroup per CU -----------------------------------------------------------------------------------------------------------
// flux_kernel above runs four independent 256-thread blocks per CU, each with its own copy of every LDS table: 4 x 6.7 KB of the CU's
// 160 KB hold the same numbers four times, and the long tables (COARE's psi_m / psi_h, WL_COARE's absorbed fraction: 19 KB) have to be
// read through the L1 — whose texture addresser became the kernel's limiter once they had replaced the arithmetic (round 4: doubling the
// table loads costs +33 %, profiles/r4_notes.md).  flux_kernel_cu is ONE workgroup of sixteen waves per CU: the tables exist once (23 KB,
// the long ones included: nothing goes through the texture addresser but the fields), and the workgroup is four TEAMS of four waves
// (one per SIMD) that each do what a block of flux_kernel does — own tile, own sort, own group queue — on tiles handed out by a queue in
// device memory, for the whole launch (persistent).  A team synchronises with a barrier of its own (an LDS counter: s_barrier is
// workgroup-wide and would put the four teams in step, which costs 20 %, ibid.); the odd teams take a one-round tile first so that the
// teams of a CU are out of phase from the start.
constexpr int kCuTeams = 4, kCuBlock = kCuTeams * kBlock;
struct TeamSync {
    unsigned *bar;        // LDS counter of the team
    unsigned *target;     // this wave's count of the arrivals it has to see (registers)
    __device__ __forceinline__ void operator()() const
    {
        *target += kBlock / 64;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // this wave's LDS writes are done before its arrival shows
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (;;) {
            unsigned seen = __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);
            if ((int)(seen - *target) >= 0) break;
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};

template <class R, int ALGO, bool SKIN, class S = R>
__global__ void __launch_bounds__(kCuBlock, 1) flux_kernel_cu(const FluxArgs<R, S> a_in, const DiagArgs<S> dg_in)
{
    static_assert(sizeof(R) == 8 && (ALGO == 1 || ALGO == 2) && SKIN, "the CU-wide kernel serves the fp64 COARE kernels with the skin schemes");
    using A = R;
    constexpr unsigned kDgOff = KernargLayout<R, S>::diag_offset;
#define AB_ARGS const FluxArgs<R, S> &a = kernarg_at<FluxArgs<R, S>>(0)
#define AB_DIAGS const DiagArgs<S> &dg = kernarg_at<DiagArgs<S>>(kDgOff)
    using T = Tile<R, ALGO, SKIN, false>;
    constexpr int kCells = 2 * kBlock;                      // two-round tiles, as in flux_kernel
    __shared__ R s_fa[kCuTeams][T::kFields][kCells];
    __shared__ unsigned short s_inva[kCuTeams][kCells];
    __shared__ unsigned s_cnta[kCuTeams][kSortCounters], s_basea[kCuTeams][kSortCounters];
    __shared__ int s_nexta[kCuTeams];
    __shared__ long s_tilea[kCuTeams];
    __shared__ unsigned s_bara[kCuTeams];
    const int team = (int)threadIdx.x / kBlock;
    const int tid0 = (int)threadIdx.x % kBlock;
#ifdef AB_CU_TRACE      // diagnostic variant (tools/build_variant.sh trace -DAB_CU_TRACE): where a team's time goes, printed by team 0 / 1 of workgroup 0
    long long tr[24];
    int ntr = 0;
#define AB_TR() do { if (ntr < 24) tr[ntr++] = wall_clock64(); } while (0)
    AB_TR();
#else
#define AB_TR() ((void)0)
#endif
    R (*s_f)[kCells] = s_fa[team];
    unsigned short *s_inv = s_inva[team];
    unsigned *s_cnt = s_cnta[team], *s_base = s_basea[team];
    int *s_next = &s_nexta[team];
    long *s_tile = &s_tilea[team];
    unsigned arrivals = 0;
    const TeamSync sync{&s_bara[team], &arrivals};
    // The team's next tile: the two-round tiles first, then the one-round tiles of the launch's tail; -1 when both pools are empty.
    // A team's FIRST tile is fixed by its position (no atomic): the k-th team of the even ones takes full tile k, the k-th of the odd ones
    // tail tile k (a one-round tile first puts the teams of a CU out of phase), and the two counters count the tiles handed out BEYOND those.
    // (Round 4 took the first tile from the counters too: 1 024 teams, one or two atomics each on one cache line at the same instant, served
    // one after the other by one L2 channel — 10 us of every launch and 20 us of a short one before the first team could start,
    // profiles/r5_notes.md.)
    const long nhalf = (long)gridDim.x * (kCuTeams / 2);                            // teams of each parity
    auto grab = [nhalf](const FluxArgs<R, S> &a, bool tail_first) -> long {         // thread 0 of the team
        int *q = a.queue;
        const long ntiles = a.nfull + a.ntail;
        const long s0 = a.nfull < nhalf ? a.nfull : nhalf, s1 = a.ntail < nhalf ? a.ntail : nhalf;    // handed out by position
        if (tail_first) {
            const long t = a.nfull + s1 + (long)atomicAdd(&q[1], 1);
            if (t < ntiles) return t;
        }
        long t = s0 + (long)atomicAdd(&q[0], 1);
        if (t < a.nfull) return t;
        if (tail_first) return -1;                                                 // (its look at the tail pool came back empty already)
        t = a.nfull + s1 + (long)atomicAdd(&q[1], 1);
        return t < ntiles ? t : -1;
    };
    if (tid0 == 0) {
        s_bara[team] = 0u;
        *s_next = 0;
        const FluxArgs<R, S> &a = kernarg_at<FluxArgs<R, S>>(0);
        if (blockIdx.x == 0 && team == 0 && (a.n != a_in.n || a.ql != a_in.ql || kernarg_at<DiagArgs<S>>(kDgOff).p[15] != dg_in.p[15])) atomicOr(a_in.flags, 2);   // (see flux_kernel)
        const long k = (long)blockIdx.x * (kCuTeams / 2) + team / 2;
        const bool odd = (team & 1) != 0;
        // (a launch with fewer tiles than teams: the teams beyond them leave without touching the counters)
        *s_tile = odd ? (k < a.ntail ? a.nfull + k : (a.nfull > nhalf ? grab(a, true) : -1))
                      : (k < a.nfull ? k : ((a.nfull > nhalf || a.ntail > nhalf) ? grab(a, false) : -1));
    }
    tile_sort_reset(s_cnt, tid0);
#ifdef AB_CU_FILL_LOOPS      // (A/B: the per-table copy loops, eight memory round trips in a row)
    esat_table_fill(); csg_table_fill(); cu_tables_fill();      // every thread of the workgroup: one copy for the CU
    math_tables_init<A>();                                       // (+ the workgroup's one and only s_barrier)
#else
    cu_fill_all();                                               // every table, one copy for the CU, one round trip (+ the workgroup's one and only s_barrier)
#endif
    AB_TR();
#pragma unroll 1
    for (;;) {
        // what derives from the thread index (LDS addresses of the thread's cells, the masks of the sort's prefix scan ...) is formed again
        // for every tile: hoisted out of the tile loop it is fifty registers held, or spilled, around the whole iteration
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        long tile0;
        int rounds;
        // ---- phase 1: owners load their cells (coalesced), pre-processing mod_aerobulk_compute.f90:99-126
        {
            AB_ARGS;
            long t = *s_tile;      // grabbed by thread 0 two team barriers ago (the first one: before the workgroup's barrier)
            t = ((long)__builtin_amdgcn_readfirstlane((int)(t >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)t);
            if (t < 0) break;
            tile_of(a, t, tile0, rounds);
            tile_phase1<R, ALGO, SKIN, S, A>(a, tid, tile0, rounds, tile_fetch<R, S, SKIN>(a, tid, tile0, rounds, 0), s_f, s_cnt, s_inv);
        }
        sync();
        AB_TR();
        // ---- phase 2: who computes which cell; thread 0 asks for the team's next tile
        {
            AB_ARGS;
            if (tid == 0) *s_tile = grab(a, false);      // (every thread of the team has read the current one before the barrier above)
            tile_phase2<2>(a, tid, rounds, s_cnt, s_base, s_inv, sync);
        }
        sync();
        AB_TR();
        // ---- phase 3: groups of 64 sorted cells, fetched from the team's queue
        { AB_ARGS; AB_DIAGS; tile_phase3<R, ALGO, SKIN, false, S, A, 2>(a, dg, tid, tile0, rounds, s_f, s_inv, s_next); }
        AB_TR();
        sync();
        AB_TR();
        // ---- phase 4: owners store (coalesced).  No barrier behind it: phase 4 reads and the next phase 1 writes a thread's OWN tile slots, and
        // the group queue is re-armed by thread 0 two team barriers ahead of its next use
        {
            AB_ARGS;
            tile_phase4<R, S>(a, tid, tile0, rounds, s_f);
            if (tid == 0) *s_next = 0;
        }
        AB_TR();
    }
#ifdef AB_CU_TRACE
    AB_TR();
    if ((blockIdx.x == 0 || blockIdx.x == 137) && tid0 == 0) {      // one printf per team: a line is one hostcall packet
        for (int i = ntr; i < 24; ++i) tr[i] = tr[0];
#define T(i) (int)((tr[i] - tr[0]) * 10)
        printf("CUTRACE wg %d team %d n=%d : %d | %d %d %d %d %d | %d %d %d %d %d | %d %d %d %d %d | %d %d %d %d %d | %d %d (ns since entry)\n", (int)blockIdx.x, team, ntr,
               T(1), T(2), T(3), T(4), T(5), T(6), T(7), T(8), T(9), T(10), T(11), T(12), T(13), T(14), T(15), T(16), T(17), T(18), T(19), T(20), T(21), T(22), T(23));
#undef T
    }
#endif
    if (tid0 == 0) {   // the last team out re-arms the counters for the next launch
        AB_ARGS;
        int *q = a.queue;
        if (atomicAdd(&q[2], 1) == (int)gridDim.x * kCuTeams - 1) { atomicExch(&q[0], 0); atomicExch(&q[1], 0); atomicExch(&q[2], 0); }
    }
#undef AB_ARGS
#undef AB_DIAGS
}

template <class R, int ALGO, bool SKIN, class S = R, class A = R> static hipError_t launch_t(const FluxCall &c, hipStream_t stream)
{
    using T = Tile<R, ALGO, SKIN, !std::is_same<R, A>::value>;
    DiagArgs<S> dg;
    bool diag = false;
    for (int i = 0; i < 16; ++i) { dg.p[i] = (S *)c.diag[i]; diag = diag || (c.diag[i] != nullptr); }
    FluxArgs<R, S> a;
    a.sst = (const S *)c.sst; a.t_zt = (const S *)c.t_zt; a.hum = (const S *)c.hum;
    a.u = (const S *)c.u; a.v = (const S *)c.v; a.slp = (const S *)c.slp;
    a.rad_sw = (const S *)c.rad_sw; a.rad_lw = (const S *)c.rad_lw; a.lon = (const S *)c.lon;
    a.ql = (S *)c.ql; a.qh = (S *)c.qh; a.tau_x = (S *)c.tau_x; a.tau_y = (S *)c.tau_y;
    a.evap = (S *)c.evap; a.t_s = (S *)c.t_s;
    a.wl0 = (S *)c.wl[0]; a.wl1 = (S *)c.wl[1]; a.wl2 = (S *)c.wl[2]; a.wl3 = (S *)c.wl[3];
    a.flags = c.flags;
    a.n = c.n;
    a.h = make_heights<R>(c.zt, c.zu);
    a.nb_iter = c.nb_iter; a.hum_type = c.hum_type; a.wl_load = c.wl_load; a.wl_store = c.wl_store;
    a.wl_live = 0;
    if (c.nb_iter <= 64)
        for (int jit = 1; jit <= c.nb_iter; ++jit)
            if (c.nb_iter % jit == 0) a.wl_live |= 1ull << (jit - 1);
    a.isecday = c.isecday;
    a.dawn_uniform = dawn_at_lon0(c.isecday);
    a.regroup = c.regroup ? 1 : 0;
    const long rounds = tile_rounds(c.n, T::kRounds, T::kOcc);
    a.rounds = (int)rounds;
    const long tile = (ALGO == 3) ? kBlock : rounds * kBlock;
    // The hardware starts blocks in index order; the launch ends when the last ones drain, the chip emptying over one block's life.
    // The grid therefore ends with one-round tiles, as many as the chip holds blocks at once: 1440x1080 is 3 037 two-round tiles for
    // 1 280 resident slots — two full waves and a third that holds the chip for a whole life at 37 % occupancy (0.205 ms where the
    // large-grid rate gives 0.156) — or 2 398 long tiles followed by 1 280 short ones.
    long nfull = (c.n + tile - 1) / tile, nblk = nfull;
    // (AEROBULK_AMD_TAIL_X: tuning knob of the probes — one-round tiles per resident slot; default: 1 for the persistent kernel's pool, 0.5 for the
    // blocks of flux_kernel — measured on 1440x1080, where the tail is a quarter of the launch: x0 / 0.5 / 1 / 1.5 / 2 / 3 = 0.1362 / 0.1340 /
    // 0.1381 / 0.1360 / 0.1377 / 0.1399 ms, profiles/r6_notes.md; flux_kernel_cu on 4320x450: x0.5 = x1 within 0.2 %, profiles/r5_notes.md §2)
    static const double tail_knob = []{ const char *e = getenv("AEROBULK_AMD_TAIL_X"); const double v = e ? atof(e) : -1.; return v; }();
    auto shape = [&](double x) {
        nfull = (c.n + tile - 1) / tile; nblk = nfull;
        if (ALGO != 3 && rounds > 1) {
            const long tail = std::min<long>(c.n, (long)(x * (double)(resident_block_slots(T::kOcc) * (long)kBlock)) / kBlock * kBlock);
            nfull = (c.n - tail) / tile;
            nblk = nfull + (c.n - nfull * tile + kBlock - 1) / kBlock;
        }
        a.nfull = nfull;
        a.ntail = nblk - nfull;
    };
    shape(tail_knob >= 0. ? tail_knob : 1.);
    a.queue = c.flags ? c.flags + 4 : nullptr;
    if (nblk <= 0) return hipSuccess;
    // fp64 COARE with the skin schemes on a grid that fills the chip several times over: one workgroup per CU (flux_kernel_cu)
    if constexpr (std::is_same<R, double>::value && std::is_same<A, double>::value && (ALGO == 1 || ALGO == 2) && SKIN) {
        // AEROBULK_AMD_CU_KERNEL=0: never (A/B); =1: on every grid (tests: the golden vectors are 2 048 cells)
        static const int mode = []{ const char *e = getenv("AEROBULK_AMD_CU_KERNEL"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }();
        const long cus = resident_block_slots(4) / 4;
        if (!diag && a.queue && rounds <= 2 && (mode == 1 || (mode < 0 && rounds == 2 && nblk >= 3 * cus * kCuTeams))) {   // from ~1.6 M cells (tools/cu_threshold_probe.py: 4320x450, what one rank of eight owns: 0.98 of the block kernel's time; 4320x225: 1.07)
            // the tile queue (three counters behind the error flag) is re-armed by the last team of every launch and must be zero here;
            // ab_session_check reads the counters with the error flag and reports AB_ERR_STATE if a launch left them otherwise (a
            // launch cut short, a caller that broke the one-stream rule).  (A 12-byte hipMemsetAsync in front of every launch was
            // measured instead: a dispatch of its own, +4.5 us on every launch of a slab.)
#ifdef AB_QUEUE_MEMSET      // (A/B: 4.5 us per launch, profiles/r5_notes.md)
            if (hipError_t e = hipMemsetAsync(a.queue, 0, 3 * sizeof(int), stream); e != hipSuccess) return e;
#endif
            // (AEROBULK_AMD_CU_GRID: probe knob — fewer workgroups than CUs, profiles/r6_notes.md "an integer number of tiles per team")
            static const long grid_knob = []{ const char *e = getenv("AEROBULK_AMD_CU_GRID"); return e ? atol(e) : 0L; }();
            const long grid = (grid_knob > 0 && grid_knob < cus) ? grid_knob : cus;
            hipLaunchKernelGGL((flux_kernel_cu<R, ALGO, SKIN, S>), dim3((unsigned)grid), dim3(kCuBlock), 0, stream, a, dg);
            return hipGetLastError();
        }
    }
    if (tail_knob < 0.) shape(0.5);      // the block kernel's launch shape
    if (diag) hipLaunchKernelGGL((flux_kernel<R, ALGO, SKIN, true, S, A>), dim3((unsigned)nblk), dim3(kBlock), 0, stream, a, dg);
    else hipLaunchKernelGGL((flux_kernel<R, ALGO, SKIN, false, S, A>), dim3((unsigned)nblk), dim3(kBlock), 0, stream, a, dg);
    return hipGetLastError();
}

template <class R, class S = R, class A = R> static hipError_t launch_r(const FluxCall &c, hipStream_t s)
{
    switch (c.algo) {
    case 1: return c.skin ? launch_t<R, 1, true, S, A>(c, s) : launch_t<R, 1, false, S, A>(c, s);
    case 2: return c.skin ? launch_t<R, 2, true, S, A>(c, s) : launch_t<R, 2, false, S, A>(c, s);
    case 3: return launch_t<R, 3, false, S, A>(c, s);
    case 4: return c.skin ? launch_t<R, 4, true, S, A>(c, s) : launch_t<R, 4, false, S, A>(c, s);
    case 5: return launch_t<R, 5, false, S, A>(c, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_flux(const FluxCall &c, hipStream_t stream)
{
    if (c.f32 && c.compute64 == 2) return launch_r<float, float, double>(c, stream);   // AB_F32_MIXED
    if (c.f32 && c.compute64) return launch_r<double, float>(c, stream);   // AB_F32_STORAGE
    return c.f32 ? launch_r<float>(c, stream) : launch_r<double>(c, stream);
}

// ------------------------------------------------------------------------------------------------
// AEROBULK_INIT statistics: sanity mask (mod_aerobulk.f90:105-115, ranges mod_const.f90:138-146) and
// masked sum/min/max of every field for type_of_humidity (mod_phymbl.f90:1957-2007) and
// check_unit_consistency (:1851-1954).  HBM-bound streaming reduction: grid-stride, wave shuffles,
// one partial row per block; the host folds the <=2048 rows.
template <class R>
__global__ void __launch_bounds__(kBlock) init_stats_kernel(const R *sst, const R *ta, const R *hum, const R *u,
                                                            const R *v, const R *slp, const R *rsw, const R *rlw,
                                                            long n, double *partials)
{
    math_tables_init<double>();
    double cnt = 0.;
    double sum[kStatFields], mn[kStatFields], mx[kStatFields];
#pragma unroll
    for (int f = 0; f < kStatFields; ++f) { sum[f] = 0.; mn[f] = 1.e300; mx[f] = -1.e300; }
    const bool rad = (rsw != nullptr) && (rlw != nullptr);
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < n; k += (long)gridDim.x * kBlock) {
        double x[kStatFields];
        x[0] = (double)sst[k]; x[1] = (double)ta[k]; x[2] = (double)slp[k];
        x[3] = (double)u[k]; x[4] = (double)v[k];
        x[5] = sqrt(x[3] * x[3] + x[4] * x[4]);
        x[6] = (double)hum[k];
        x[7] = rad ? (double)rsw[k] : 0.;
        x[8] = rad ? (double)rlw[k] : 0.;
        bool ok = !((x[0] < 270.) || (x[0] > 320.));
        ok = ok && !((x[1] < 180.) || (x[1] > 330.));
        ok = ok && !((x[2] < 80000.) || (x[2] > 110000.));
        ok = ok && !(x[5] > 50.);
        if (rad) {
            ok = ok && !((x[7] < 0.) || (x[7] > 1500.));
            ok = ok && !((x[8] < 0.) || (x[8] > 750.));
        }
        if (ok) {
            cnt += 1.;
#pragma unroll
            for (int f = 0; f < kStatFields; ++f) {
                sum[f] += x[f];
                mn[f] = x[f] < mn[f] ? x[f] : mn[f];
                mx[f] = x[f] > mx[f] ? x[f] : mx[f];
            }
        }
    }
    __shared__ double red[kBlock / 64][kStatStride];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto wsum = [](double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; };
    auto wmin = [](double v) { for (int o = 32; o > 0; o >>= 1) { double t = __shfl_down(v, o, 64); v = t < v ? t : v; } return v; };
    auto wmax = [](double v) { for (int o = 32; o > 0; o >>= 1) { double t = __shfl_down(v, o, 64); v = t > v ? t : v; } return v; };
    cnt = wsum(cnt);
    if (lane == 0) red[wave][0] = cnt;
#pragma unroll
    for (int f = 0; f < kStatFields; ++f) {
        const double s = wsum(sum[f]), a = wmin(mn[f]), b = wmax(mx[f]);
        if (lane == 0) { red[wave][1 + 3 * f] = s; red[wave][2 + 3 * f] = a; red[wave][3 + 3 * f] = b; }
    }
    __syncthreads();
    if (threadIdx.x < kStatStride) {
        const int j = threadIdx.x;
        double r = red[0][j];
        const int kind = (j == 0) ? 0 : ((j - 1) % 3);  // 0 sum, 1 min, 2 max
        for (int w = 1; w < kBlock / 64; ++w) {
            const double t = red[w][j];
            r = (kind == 0) ? r + t : (kind == 1 ? (t < r ? t : r) : (t > r ? t : r));
        }
        partials[(long)blockIdx.x * kStatStride + j] = r;
    }
}

hipError_t launch_init_stats(const void *sst, const void *t_air, const void *hum, const void *u, const void *v,
                             const void *slp, const void *rad_sw, const void *rad_lw, long n, int f32,
                             double *partials, hipStream_t stream, int nblocks)
{
    if (f32)
        hipLaunchKernelGGL(init_stats_kernel<float>, dim3(nblocks), dim3(kBlock), 0, stream, (const float *)sst,
                           (const float *)t_air, (const float *)hum, (const float *)u, (const float *)v,
                           (const float *)slp, (const float *)rad_sw, (const float *)rad_lw, n, partials);
    else
        hipLaunchKernelGGL(init_stats_kernel<double>, dim3(nblocks), dim3(kBlock), 0, stream, (const double *)sst,
                           (const double *)t_air, (const double *)hum, (const double *)u, (const double *)v,
                           (const double *)slp, (const double *)rad_sw, (const double *)rad_lw, n, partials);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// a product the compiler may not contract into an fma with the sum that follows (ROCm's __dmul_rn is a plain `*`)
__device__ __forceinline__ double dadd(double a, double b) { return a + b; }
__device__ __forceinline__ double mul_rn(double a, double b)
{
    double t = a * b;
    asm volatile("" : "+v"(t));
    return t;
}
// Synthetic inputs of SURVEY.md §8d.  Always evaluated in fp64, stored in R.
template <class R>
__global__ void __launch_bounds__(kBlock) synth_kernel(R *sst, R *t_zt, R *q_zt, R *u, R *v, R *slp, R *rsw, R *rlw,
                                                       long ni, long j0, long n)
{
    math_tables_init<double>();
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const double i = (double)(k % ni + 1), j = (double)(j0 + k / ni + 1);
    const double A[7] = {0.6180339887498949, 0.5698402909980532, 0.8191725133961645, 0.4142135623730951,
                         0.2360679774997897, 0.3166247903553998, 0.1231056256176606};
    const double B[7] = {0.7548776662466927, 0.3247179572447460, 0.6710436067037893, 0.7320508075688772,
                         0.6457513110645906, 0.6055512754639891, 0.3588989435406740};
    const double C[7] = {0., 0.1, 0.2, 0.3, 0.4, 0.5, 0.6};
    // every product rounded on its own (mul_rn), never contracted into an fma: bit-identical to the host generator of the
    // oracle (gcc -ffp-contract=off), which frac() of numbers up to ~6000 would not survive otherwise
    double r[7];
#pragma unroll
    for (int m = 0; m < 7; ++m) {
        const double x = dadd(dadd(mul_rn(i, A[m]), mul_rn(j, B[m])), C[m]);
        r[m] = dadd(x, -floor(x));
    }
    const double s = dadd(274.15, mul_rn(29., r[0]));
    const double t = dadd(dadd(s, -6.), mul_rn(9., r[1]));
    const double p = dadd(98000., mul_rn(5000., r[2]));
    sst[k] = (R)s;
    t_zt[k] = (R)t;
    slp[k] = (R)p;
    q_zt[k] = (R)(dadd(0.55, mul_rn(0.4, r[3])) * q_sat<double, false>(t, p));   // q_sat: device math, within an ulp or two of libm
    u[k] = (R)dadd(-14., mul_rn(28., r[4]));
    v[k] = (R)dadd(-14., mul_rn(28., r[5]));
    if (rsw) rsw[k] = (R)mul_rn(900., r[6]);
    if (rlw) rlw[k] = (R)dadd(250., mul_rn(200., r[0]));
}

hipError_t launch_synth(void *sst, void *t_zt, void *q_zt, void *u, void *v, void *slp, void *rad_sw, void *rad_lw,
                        long ni, long j0, long nj_local, int f32, hipStream_t stream)
{
    const long n = ni * nj_local;
    const long nblk = (n + kBlock - 1) / kBlock;
    if (nblk <= 0) return hipSuccess;
    if (f32)
        hipLaunchKernelGGL(synth_kernel<float>, dim3((unsigned)nblk), dim3(kBlock), 0, stream, (float *)sst,
                           (float *)t_zt, (float *)q_zt, (float *)u, (float *)v, (float *)slp, (float *)rad_sw,
                           (float *)rad_lw, ni, j0, n);
    else
        hipLaunchKernelGGL(synth_kernel<double>, dim3((unsigned)nblk), dim3(kBlock), 0, stream, (double *)sst,
                           (double *)t_zt, (double *)q_zt, (double *)u, (double *)v, (double *)slp, (double *)rad_sw,
                           (double *)rad_lw, ni, j0, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Unit-test hook for the fp64 device math of ab_fastmath.hpp (tests/test_gpu_math.py).
__global__ void __launch_bounds__(kBlock) math_test_kernel(int op, const double *x, const double *y, double *o, long n)
{
    psi_tables_fill();
    math_tables_init<double>();
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const double a = x[k], b = y ? y[k] : 1.0;
    double r;
    switch (op) {
    case 0: r = fm::qdiv(a, b); break;
    case 1: r = fm::qrcp(a); break;
    case 2: r = fm::qsqrt(a); break;
    case 3: r = fm::qlog(a); break;
    case 4: r = fm::qlog10(a); break;
    case 5: r = fm::qexp(a); break;
    case 6: r = fm::qexp10(a); break;
    case 7: r = fm::qatan(a); break;
    case 8: r = fm::qcbrt(a); break;
    case 9: r = fm::qrcbrt_mid(a); break;
    case 10: r = e_sat<double, false>(a); break;
    case 13: r = e_sat<double, true>(a); break;     // through the piecewise LDS table of the tiled flux kernels
    case 11: r = pow_pos<double>(a, b); break;
    case 12: r = fm::qrqrt_mid(a); break;
    default: r = 0.; break;
    }
    o[k] = r;
}

hipError_t launch_math_test(int op, const double *x, const double *y, double *o, long n, hipStream_t stream)
{
    const long nblk = (n + kBlock - 1) / kBlock;
    if (nblk <= 0) return hipSuccess;
    hipLaunchKernelGGL(math_test_kernel, dim3((unsigned)nblk), dim3(kBlock), 0, stream, op, x, y, o, n);
    return hipGetLastError();
}

}  // namespace ab

// ------------------------------------------------------------------------------------------------
// tools/isa_profile.py: basic-block execution counters of an INSTRUMENTED build of one kernel (the tool patches the device
// assembly; every basic block adds 1 per wave and 1 per active lane to its pair of counters).  Never compiled into the library.
#ifdef AB_ISA_PROFILE
namespace ab {
__device__ __attribute__((used)) unsigned ab_prof_counters[16384];
}
extern "C" int ab_prof_reset()
{
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(ab::ab_prof_counters)) != hipSuccess) return 1;
    return hipMemset(p, 0, sizeof(unsigned) * 16384) == hipSuccess && hipDeviceSynchronize() == hipSuccess ? 0 : 2;
}
extern "C" int ab_prof_read(unsigned *out, int n)
{
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ab::ab_prof_counters), sizeof(unsigned) * (size_t)n) == hipSuccess ? 0 : 1;
}
#endif
