// ab_phymbl.hpp — the public helper functions of the reference's `mod_phymbl` (src/mod_phymbl.f90:33-139), one cell at a time.
//
// What the GCM-side callers of AeroBulk import next to `aerobulk_model`: potential / virtual temperature, pressure at height, air
// density, saturation humidity (over water and over ice), Obukhov length, bulk Richardson number, the bulk formula, roughness
// conversions, Louis' stability functions ...  ab_phymbl.hip runs ph_cell<FN> over arrays as one elementwise HIP kernel per call;
// aerobulk_amd/fortran/mod_phymbl.f90 gives it the reference's names (the `_sclr` / `_vctr` pairs of every generic).
//
// Where the flux kernels already own the function (ab_physics.hpp: e_sat, q_sat, theta_from_z_p0_t_q, rho_air, visc_air, one_on_l,
// ri_bulk, update_qnsol_tau, alpha_sw, z0tq_lkb, ab_physics_ice.hpp: f_louis) that very function is called, so a helper value handed
// to a caller is the number the engine itself works with.  The others are written from the formulas with the engine's own elementary
// functions (ab_math.hpp).  Each block cites the reference lines whose results it reproduces (<= 1e-12 relative, tests/test_phymbl.py).
//
// The header also compiles for the host (AB_FASTMATH_HOST, tests/phymbl_host.cpp: test infrastructure like tests/physics_host.cpp);
// the library has no host path.
#pragma once
#include "ab_physics.hpp"
#include "ab_physics_ice.hpp"

namespace ab {

// function ids = enum ab_phymbl_fn of include/aerobulk_amd.h
enum {
    kPhPotTemp = 1, kPhAbsTemp, kPhVirtTemp, kPhPzFromP0, kPhThetaFromZ, kPhTFromZ, kPhRhoAir, kPhViscAir, kPhLvap, kPhCpAir,
    kPhGammaMoist, kPhOneOnL, kPhRiBulk, kPhEsat, kPhEsatIce, kPhDEsatDtIce, kPhQsat, kPhDQsatDtIce, kPhQairRh, kPhQairDp,
    kPhRhoAirAdv, kPhQsatCrude, kPhDryStaticEnergy, kPhUpdateQnsolTau, kPhBulkFormula, kPhAlphaSw, kPhQlwNet, kPhZ0FromCd,
    kPhZ0FromUstar, kPhCdFromZ0, kPhFmLouis, kPhFhLouis, kPhUN10FromUstar, kPhUN10FromCdn, kPhUN10FromCd, kPhZ0tqLkb, kPhEair,
    kPhRhAir, kPhDeltaSkinLayer, kPhRoughLengM, kPhRoughLengTq,
    // the PUBLIC functions of the algorithm modules (mod_common_coare, mod_blk_coare3p0 / coare3p6 / ncar / ecmwf / andreas)
    kPhPsiMCoare, kPhPsiHCoare, kPhPsiMNcar, kPhPsiHNcar, kPhPsiMEcmwf, kPhPsiHEcmwf, kPhPsiMAndreas, kPhPsiHAndreas,
    kPhCharnCoare3p0, kPhCharnCoare3p6, kPhCdN10Ncar, kPhChN10Ncar, kPhCeN10Ncar, kPhUStarAndreas, kPhFirstGuessCoare,
    // the skin schemes as their modules export them (mod_skin_coare.f90:28, mod_skin_ecmwf.f90:49)
    kPhCsCoare, kPhCsEcmwf, kPhWlCoare, kPhWlEcmwf, kPhCount
};
static_assert(kPhPsiMCoare == 42 && kPhUStarAndreas == 55 && kPhFirstGuessCoare == 56 && kPhWlEcmwf == 60, "enum ab_phymbl_fn of include/aerobulk_amd.h");

template <class R> struct KPh {   // the constants of mod_const.f90 / mod_phymbl.f90 that the flux kernels do not need
    static constexpr R rtt0 = R(273.16);                    // mod_const.f90:61
    static constexpr R Patm = R(101000.);                   // :96
    static constexpr R rLsub = R(2.834e+6);                 // :92
    static constexpr R emiss_i = R(0.996);                  // :56
    static constexpr R rAg_i = R(-9.09718), rBg_i = R(-3.56654), rCg_i = R(0.876793);   // Goff over ice, mod_phymbl.f90:143-148
    static constexpr R rDg_i = R(0.7858350313586662);       // LOG10(6.1071)
    static constexpr R ln10 = R(2.302585092994046);
};

// LOG for arguments a caller may hand over (zero or negative over masked cells): the engine's log wants x > 0
template <class R> __device__ __forceinline__ R ph_log(R x)
{
    if (x > R(0.)) return Mth<R>::log(x);
    return x == R(0.) ? -__builtin_huge_val() : __builtin_nan("");
}
// x ** y the way the reference's compilers evaluate it for a real exponent (x >= 0)
template <class R> __device__ __forceinline__ R ph_pow(R x, R y)
{
    if (x > R(0.)) return Mth<R>::exp(y * Mth<R>::log(x));
    return x == R(0.) ? (y > R(0.) ? R(0.) : (y == R(0.) ? R(1.) : (R)__builtin_huge_val())) : (R)__builtin_nan("");
}

// e_sat_ice_sclr :815-830 — Goff over ice, T floored at 180 K, triple point 273.16
template <class R> __device__ __forceinline__ R e_sat_ice(R pTa)
{
    using M = Mth<R>;
    const R zta = vmax(pTa, R(180.));
    const R ztmp = M::div(KPh<R>::rtt0, zta);
    const R zle = KPh<R>::rAg_i * (ztmp - R(1.)) + KPh<R>::rBg_i * M::log10(ztmp) + KPh<R>::rCg_i * (R(1.) - zta * R(1. / 273.16))
                  + KPh<R>::rDg_i;
    return R(100.) * M::exp10(zle);
}
// de_sat_dt_ice_sclr :845-861
template <class R> __device__ __forceinline__ R de_sat_dt_ice(R pTa)
{
    using M = Mth<R>;
    const R zta = vmax(pTa, R(180.));
    const R zi = M::rcp(zta);
    const R zde = -(KPh<R>::rAg_i * KPh<R>::rtt0) * (zi * zi) - KPh<R>::rBg_i * zi * R(1. / 2.302585092994046) - KPh<R>::rCg_i * R(1. / 273.16);
    return KPh<R>::ln10 * zde * e_sat_ice<R>(zta);
}
// q_sat_sclr :881-904 with its l_ice switch
template <class R> __device__ __forceinline__ R ph_q_sat(R pTa, R pslp, bool ice)
{
    if (!ice) return q_sat<R, false>(pTa, pslp);
    const R ze_s = e_sat_ice<R>(pTa);
    return Mth<R>::div(K<R>::reps0 * ze_s, pslp - K<R>::one_m_reps0 * ze_s);
}
// Pz_from_P0_tz_qz_sclr :283-318 — three barometric iterations; e_sat(T) does not depend on the iterate (same fused form as
// theta_from_z_p0_t_q of ab_physics.hpp).  zarg returns the last exponent: P0/Pz = exp(-zarg).
template <class R> __device__ __forceinline__ R pz_from_p0_tz_qz(R pz, R pslp, R pTa, R pqa, bool ice, R *zarg_out = nullptr)
{
    using M = Mth<R>;
    const R ze_s = ice ? e_sat_ice<R>(pTa) : e_sat<R, false>(pTa);
    const R c = M::div(-K<R>::grav * pz, K<R>::R_gas * pTa);
    const R zi = M::rcp(K<R>::reps0 * ze_s);
    R zpa = pslp, zarg = R(0.);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const R zf = pqa * ((zpa - K<R>::one_m_reps0 * ze_s) * zi);   // q / q_sat(T, p)
        const R zxm = (R(1.) - zf) * K<R>::rmm_dryair + zf * K<R>::rmm_water;
        zarg = c * zxm;
        zpa = pslp * M::exp(zarg);
    }
    if (zarg_out) *zarg_out = zarg;
    return zpa;
}
// delta_skin_layer_sclr :2010-2046 (the general-purpose form: the cool-skin kernels carry a strength-reduced one, cool_skin())
template <class R> __device__ __forceinline__ R delta_skin_layer(R palpha, R pQd, R pustar_a, bool has_qlat, R Qlat)
{
    using M = Mth<R>;
    R zQd = pQd;
    if (has_qlat) zQd = pQd + M::div(R(0.026) * vmin(Qlat, R(0.)) * K<R>::rCp0_w * R(1. / 2.46e+6), palpha);
    const R zusw = vmax(pustar_a, R(1.E-4)) * K<R>::sq_radrw;
    const R zusw2 = zusw * zusw;
    const R zx = vmax(M::div(palpha * K<R>::rcst_cs, zusw2 * zusw2) * zQd, R(0.));
    const R zlamb = R(6.) * ph_pow(R(1.) + ph_pow(zx, R(0.75)), R(-1. / 3.));
    const R ztmp = M::div(K<R>::rnu0_w, zusw);
    return nonneg(zQd) ? vmin(R(6.) * ztmp, R(0.007)) : zlamb * ztmp;
}

// One cell of function FN.  x[i]: the i-th array argument at this cell (garbage where bit i of `present` is clear: an OPTIONAL
// array that was not passed); par[0]: the function's scalar REAL argument (pz / pzu / pPref), flag: its LOGICAL / INTEGER one
// (l_ice, iflag); y[]: results.  Argument order = the reference's dummy-argument order with the scalars taken out.
template <int FN, class R> __device__ __forceinline__ void ph_cell(const R *x, unsigned present, const R *par, int flag, R *y)
{
    using M = Mth<R>;
    const R vk = K<R>::vkarmn;
    if constexpr (FN == kPhPotTemp) {            // pot_temp :163-200  ( pTa, pPz, [pPref] )
        const R pref = (present & 4u) ? x[2] : par[0];
        y[0] = x[0] * ph_pow(M::div(pref, x[1]), K<R>::rpoiss_dry);
    } else if constexpr (FN == kPhAbsTemp) {     // abs_temp :205-242
        const R pref = (present & 4u) ? x[2] : par[0];
        y[0] = M::div(x[0], vmax(ph_pow(M::div(pref, x[1]), K<R>::rpoiss_dry), R(1.E-9)));
    } else if constexpr (FN == kPhVirtTemp) {    // virt_temp :247-276
        y[0] = virt_temp<R>(x[0], x[1]);
    } else if constexpr (FN == kPhPzFromP0) {    // Pz_from_P0_tz_qz :283-337  ( pz ; pslp, pTa, pqa ; l_ice )
        y[0] = pz_from_p0_tz_qz<R>(par[0], x[0], x[1], x[2], flag != 0);
    } else if constexpr (FN == kPhThetaFromZ) {  // Theta_from_z_P0_T_q :343-375  ( pz ; pslp, pTa, pqa )
        if (flag == 0) {
            y[0] = theta_from_z_p0_t_q<R, false>(par[0], x[0], x[1], x[2]);
        } else {                                  // (the sticky l_ice of Pz_from_P0_tz_qz, see mod_phymbl.f90 of the Fortran host)
            R zarg;
            (void)pz_from_p0_tz_qz<R>(par[0], x[0], x[1], x[2], true, &zarg);
            y[0] = x[1] * M::exp(-K<R>::rpoiss_dry * zarg);
        }
    } else if constexpr (FN == kPhTFromZ) {      // T_from_z_P0_Theta_q :380-420  ( pz ; pslp, pThta, pqa ): four fixed-point sweeps
        R zTa = x[1] - K<R>::rgamma_dry * par[0];
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
            R zarg;
            (void)pz_from_p0_tz_qz<R>(par[0], x[0], zTa, x[2], flag != 0, &zarg);
            zTa = M::div(x[1], vmax(M::exp(-K<R>::rpoiss_dry * zarg), R(1.E-9)));   // abs_temp with pPref = pslp: (P0/Pz)^kappa = exp(-kappa zarg)
        }
        y[0] = zTa;
    } else if constexpr (FN == kPhRhoAir) {      // rho_air :522-546
        y[0] = rho_air<R>(x[0], x[1], x[2]);
    } else if constexpr (FN == kPhViscAir) {     // visc_air :549-574
        y[0] = visc_air<R>(x[0]);
    } else if constexpr (FN == kPhLvap) {        // L_vap :579-598
        y[0] = (R(2.501) - R(0.00237) * (x[0] - K<R>::rt0)) * R(1.e6);
    } else if constexpr (FN == kPhCpAir) {       // cp_air :603-622
        y[0] = K<R>::rCp_dry + K<R>::rCp_vap * x[0];
    } else if constexpr (FN == kPhGammaMoist) {  // gamma_moist :627-661
        const R zta = vmax(x[0], R(180.)), zqa = vmax(x[1], R(1.E-6));
        const R zwa = M::div(zqa, R(1.) - zqa);
        const R ziRT = M::rcp(K<R>::R_dry * zta);
        const R zLvap = (R(2.501) - R(0.00237) * (x[0] - K<R>::rt0)) * R(1.e6);
        y[0] = M::div(K<R>::grav * (R(1.) + zLvap * zwa * ziRT), K<R>::rCp_dry + M::div(zLvap * zLvap * zwa * K<R>::reps0 * ziRT, zta));
    } else if constexpr (FN == kPhOneOnL) {      // One_on_L :666-708
        y[0] = one_on_l<R>(x[0], x[1], x[2], x[3], x[4]);
    } else if constexpr (FN == kPhRiBulk) {      // Ri_bulk :712-772  ( pz ; psst, pThta, pssq, pqa, pub, [pTa_layer, pqa_layer] )
        if ((present & 0x60u) == 0x60u) {
            const R zsstv = virt_temp<R>(x[0], x[2]);
            const R zdthv = virt_temp<R>(x[1], x[3]) - zsstv;
            y[0] = M::div(K<R>::grav * zdthv * par[0], virt_temp<R>(x[5], x[6]) * x[4] * x[4]);
        } else {
            y[0] = ri_bulk<R, R>(par[0], x[0], x[1], x[2], x[3], x[4]);
        }
    } else if constexpr (FN == kPhEsat) {        // e_sat :777-811
        y[0] = e_sat<R, false>(x[0]);
    } else if constexpr (FN == kPhEsatIce) {     // e_sat_ice :815-843
        y[0] = e_sat_ice<R>(x[0]);
    } else if constexpr (FN == kPhDEsatDtIce) {  // de_sat_dt_ice :845-875
        y[0] = de_sat_dt_ice<R>(x[0]);
    } else if constexpr (FN == kPhQsat) {        // q_sat :881-921  ( pTa, pslp ; l_ice )
        y[0] = ph_q_sat<R>(x[0], x[1], flag != 0);
    } else if constexpr (FN == kPhDQsatDtIce) {  // dq_sat_dt_ice :926-958
        const R ze_s = e_sat_ice<R>(x[0]);
        const R ztmp = (K<R>::reps0 - R(1.)) * ze_s + x[1];
        y[0] = M::div(K<R>::reps0 * x[1] * de_sat_dt_ice<R>(x[0]), ztmp * ztmp);
    } else if constexpr (FN == kPhQairRh) {      // q_air_rh :963-985
        y[0] = q_air_rh<R, false>(x[0], x[1], x[2]);
    } else if constexpr (FN == kPhQairDp) {      // q_air_dp :990-1000
        y[0] = q_air_dp<R, false>(x[0], x[1]);
    } else if constexpr (FN == kPhRhoAirAdv) {   // rho_air_adv :1008-1024  ( pTa, pqa, pslp ) + x[3] = e_air(pqa, pslp), made by the caller
        y[0] = M::div(x[2], K<R>::R_dry * M::div(x[0], R(1.) - M::div(x[3], x[2]) * K<R>::one_m_reps0));
    } else if constexpr (FN == kPhQsatCrude) {   // q_sat_crude :1029-1038
        y[0] = M::div(R(640380.), x[1]) * M::exp(M::div(R(-5107.4), x[0]));
    } else if constexpr (FN == kPhDryStaticEnergy) {   // dry_static_energy :1043-1054  ( pz ; pTa, pqa )
        y[0] = K<R>::grav * par[0] + (K<R>::rCp_dry + K<R>::rCp_vap * x[1]) * x[0];
    } else if constexpr (FN == kPhUpdateQnsolTau) {    // UPDATE_QNSOL_TAU :1059-1144  ( pzu ; pts, pqs, pThta, pqa, pust, ptst, pqst, pwnd, pUb, pslp, prlw )
        R qns, tau, qlat;
        update_qnsol_tau<R, R, false>(par[0], x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], x[8], x[9], x[10], qns, tau, qlat);
        y[0] = qns; y[1] = tau; y[2] = qlat;
    } else if constexpr (FN == kPhBulkFormula) {       // BULK_FORMULA :1149-1261  ( pzu ; pts, pqs, pThta, pqa, pCd, pCh, pCe, pwnd, pUb, pslp ; l_ice )
        const R zta = x[2] - K<R>::rgamma_dry * par[0];
        const R zir = M::rcp(K<R>::R_dry * zta * (R(1.) + K<R>::rctv0 * x[3]));   // rho_air twice (:1183-1184), one reciprocal (bulk_formula, ab_physics.hpp)
        R zrho = vmax(x[9] * zir, R(0.8));
        zrho = vmax((x[9] - zrho * K<R>::grav * par[0]) * zir, R(0.8));
        const R zUrho = x[8] * vmax(zrho, R(1.));
        const R zevap = zUrho * x[6] * (x[3] - x[1]);
        y[0] = zUrho * x[4] * x[7];
        y[1] = zUrho * x[5] * (x[2] - x[0]) * (K<R>::rCp_dry + K<R>::rCp_vap * x[3]);
        if (flag != 0) { y[2] = KPh<R>::rLsub * zevap; y[3] = vmin(zevap, R(0.)); }
        else { y[2] = (R(2.501) - R(0.00237) * (x[0] - K<R>::rt0)) * R(1.e6) * zevap; y[3] = zevap; }
        y[4] = zrho;
    } else if constexpr (FN == kPhAlphaSw) {     // alpha_sw :1267-1286
        y[0] = alpha_sw<R>(x[0]);
    } else if constexpr (FN == kPhQlwNet) {      // qlw_net :1291-1330  ( pdwlw, pts ; l_ice )
        const R zt2 = x[1] * x[1];
        y[0] = (flag != 0 ? KPh<R>::emiss_i : K<R>::emiss_w) * (x[0] - K<R>::stefan * zt2 * zt2);
    } else if constexpr (FN == kPhZ0FromCd) {    // z0_from_Cd :1335-1366  ( pzu ; pCd, [ppsi] )
        const R a = vk * M::rsqrt_pos(x[0]) + ((present & 2u) ? x[1] : R(0.));
        y[0] = par[0] * M::exp(-a);
    } else if constexpr (FN == kPhZ0FromUstar) { // z0_from_ustar :1371-1391  ( pzu ; pus, puzu )
        y[0] = par[0] * M::exp(-M::div(vk * x[1], x[0]));
    } else if constexpr (FN == kPhCdFromZ0) {    // Cd_from_z0 :1396-1414  ( pzu ; pz0, [ppsi] )
        const R t = M::rcp(ph_log(M::div(par[0], x[0])) - ((present & 2u) ? x[1] : R(0.)));
        y[0] = K<R>::vkarmn2 * t * t;
    } else if constexpr (FN == kPhFmLouis || FN == kPhFhLouis) {   // f_m_louis :1419-1453, f_h_louis :1458-1492  ( pzu ; pRib, pCxn, pz0 )
        y[0] = f_louis<R>(M::div(par[0], x[2]) + R(1.), x[0], x[1], FN == kPhFmLouis ? R(10.) : R(15.));
    } else if constexpr (FN == kPhUN10FromUstar) {   // UN10_from_ustar :1498-1510  ( pzu ; pUzu, pus, ppsi )
        y[0] = x[0] - x[1] * K<R>::inv_vk * (ph_log(par[0] * R(0.1)) - x[2]);
    } else if constexpr (FN == kPhUN10FromCdn) {     // UN10_from_CDN :1515-1527  ( pzu ; pUb, pCdn, ppsi )
        y[0] = M::div(x[0], R(1.) + M::sqrt(x[1]) * K<R>::inv_vk * (ph_log(par[0] * R(0.1)) - x[2]));
    } else if constexpr (FN == kPhUN10FromCd) {      // UN10_from_CD :1532-1558  ( pzu ; pUb, pCd, ppsi ): LOG(10/z0_from_Cd) in the log domain
        const R sq = M::sqrt(x[1]);
        y[0] = sq * x[0] * K<R>::inv_vk * (ph_log(M::div(R(10.), par[0])) + M::div(vk, sq) + x[2]);
    } else if constexpr (FN == kPhZ0tqLkb) {     // z0tq_LKB :1635-1701  ( iflag ; pRer, pz0 )
        R z0t, z0q;
        z0tq_lkb<R>(x[0], x[1], z0t, z0q);
        y[0] = flag == 2 ? z0q : z0t;
    } else if constexpr (FN == kPhEair) {        // one sweep of e_air's fixed point :1706-1736  ( pqa, pslp ) + x[2] = previous iterate
        y[0] = x[0] * R(1. / (287.05 / 461.495)) * (x[1] - K<R>::one_m_reps0 * x[2]);
    } else if constexpr (FN == kPhRhAir) {       // rh_air :1741-1753  ( pqa, pTa, pslp ) + x[3] = e_air(pqa, pslp)
        y[0] = R(100.) * M::div(x[3], e_sat<R, false>(x[1]));
    } else if constexpr (FN == kPhDeltaSkinLayer) {  // delta_skin_layer_sclr :2010-2046  ( palpha, pQd, pustar_a, [Qlat] )
        y[0] = delta_skin_layer<R>(x[0], x[1], x[2], (present & 8u) != 0, (present & 8u) ? x[3] : R(0.));
    } else if constexpr (FN == kPhRoughLengM) {      // rough_leng_m, src/ice/mod_blk_ice_an05.f90:232-255  ( pus, pnua ): Andreas et al. 2005 eq. 19
        y[0] = an05_rough_leng_m<R>(x[0], x[1]);
    } else if constexpr (FN == kPhRoughLengTq) {     // rough_leng_tq, src/ice/mod_blk_ice_an05.f90:257-312  ( pz0, pus, pnua ) -> z0t, z0q (eq. 22)
        R lt, lq;
        an05_log_z0tq<R>(x[0], R(0.), x[1], x[2], lt, lq);       // LOG(z0s / z0)
        y[0] = x[0] * M::exp(lt); y[1] = x[0] * M::exp(lq);
    // ---- the stability functions and neutral coefficients the algorithm modules export: the very device functions of the flux kernels
    // (ab_physics.hpp), in their forms without LDS tables, valid on the whole real axis of zeta (the closed forms beyond the tables' range)
    } else if constexpr (FN == kPhPsiMCoare) {       // psi_m_coare, mod_common_coare.f90:217-302  ( pzeta )
        psi_coare<R>(x[0], &y[0], nullptr);
    } else if constexpr (FN == kPhPsiHCoare) {       // psi_h_coare :305-392
        psi_coare<R>(x[0], nullptr, &y[0]);
    } else if constexpr (FN == kPhPsiMNcar) {        // psi_m_ncar, mod_blk_ncar.f90:333-376
        psi_ncar<R, false>(x[0], &y[0], nullptr);
    } else if constexpr (FN == kPhPsiHNcar) {        // psi_h_ncar :379-420
        psi_ncar<R, false>(x[0], nullptr, &y[0]);
    } else if constexpr (FN == kPhPsiMEcmwf) {       // psi_m_ecmwf, mod_blk_ecmwf.f90:441-495
        psi_ecmwf<R>(x[0], &y[0], nullptr);
    } else if constexpr (FN == kPhPsiHEcmwf) {       // psi_h_ecmwf :498-548
        psi_ecmwf<R>(x[0], nullptr, &y[0]);
    } else if constexpr (FN == kPhPsiMAndreas) {     // psi_m_andreas, mod_blk_andreas.f90:307-360
        y[0] = psi_m_andreas<R, false>(x[0]);
    } else if constexpr (FN == kPhPsiHAndreas) {     // psi_h_andreas :363-410
        y[0] = psi_h_andreas<R>(x[0]);
    } else if constexpr (FN == kPhCharnCoare3p0) {   // charn_coare3p0, mod_blk_coare3p0.f90:420-447  ( pwnd )
        y[0] = charn_coare3p0<R>(x[0]);
    } else if constexpr (FN == kPhCharnCoare3p6) {   // charn_coare3p6, mod_blk_coare3p6.f90:417-445  ( pwnd )
        y[0] = charn_coare3p6<R>(x[0]);
    } else if constexpr (FN == kPhCdN10Ncar) {       // cd_n10_ncar, mod_blk_ncar.f90:244-284  ( pw10 )
        y[0] = cd_n10_ncar<R>(x[0]);
    } else if constexpr (FN == kPhChN10Ncar) {       // ch_n10_ncar :287-310  ( psqrtcdn10, pstab in [0, 1] )
        y[0] = vmax(R(1.e-3) * x[0] * (R(18.) * x[1] + R(32.7) * (R(1.) - x[1])), K<R>::Cx_min);
    } else if constexpr (FN == kPhCeN10Ncar) {       // ce_n10_ncar :313-330  ( psqrtcdn10 )
        y[0] = vmax(R(1.e-3) * (R(34.6) * x[0]), K<R>::Cx_min);
    } else if constexpr (FN == kPhUStarAndreas) {    // u_star_andreas, mod_blk_andreas.f90:275-305  ( pun10 )
        const R za = x[0] - R(8.271);
        y[0] = R(0.239) + R(0.0433) * (za + M::sqrt(R(0.12) * za * za + R(0.181)));
    } else if constexpr (FN == kPhFirstGuessCoare) {   // FIRST_GUESS_COARE, mod_common_coare.f90:33-179  ( zt, zu ; psst, t_zt, pssq, q_zt, U_zu, pcharn )
        // -> pus, pts, pqs, t_zu, q_zu, Ubzu, pz0: the first block of the engine's TURB_COARE* / TURB_ECMWF kernels, with the heights' logarithms
        // formed here (the flux kernels get them from the host once per launch: ab_launch.hpp make_heights)
        Heights<R> h;
        h.zt = par[0]; h.zu = par[1];
        h.log_zt = ph_log(par[0]); h.log_zu = ph_log(par[1]); h.log_10 = ph_log(R(10.));
        h.log_ztu = ph_log(M::div(par[0], par[1])); h.log_zu10 = ph_log(par[1] * R(0.1));
        h.fg_ca = M::div(R(0.035) * ph_log(R(10. / 0.0001)), ph_log(par[1] * R(1. / 0.0001)));
        h.inv_zu = M::rcp(par[1]); h.zt_o_zu = M::div(par[0], par[1]);
        h.fg_cb = -R(0.004 * 600. * 1.2 * 1.2 * 1.2) * h.inv_zu;
        h.zt_eq_zu = M::abs(par[1] - par[0]) < R(0.01) ? 1 : 0;
        first_guess_coare<R, R>(h, x[0], x[1], x[2], x[3], x[4], x[5], y[0], y[1], y[2], y[3], y[4], y[5], y[6]);
    // ---- the skin schemes standing alone: the device functions TURB_COARE3P0 / 3P6 / ECMWF iterate with, one call per cell
    } else if constexpr (FN == kPhCsCoare) {         // CS_COARE, mod_skin_coare.f90:48-93  ( pQsw, pQnsol, pustar, pSST, pQlat ) -> pdT_cs
        y[0] = cool_skin<R, true, false>(x[0], x[1], x[2], alpha_sw<R>(x[3]), x[4]);
    } else if constexpr (FN == kPhCsEcmwf) {         // CS_ECMWF, mod_skin_ecmwf.f90:68-110  ( pQsw, pQnsol, pustar, pSST ) -> pdT_cs
        y[0] = cool_skin<R, false, false>(x[0], x[1], x[2], alpha_sw<R>(x[3]), R(0.));
    } else if constexpr (FN == kPhWlCoare) {         // WL_COARE, mod_skin_coare.f90:97-250  ( isd ; pQsw, pQnsol, pTau, pSST, plon, dT_wl, Hz_wl, Qnt_ac, Tau_ac ; iwait )
        // -> the four state variables after the call (unchanged when iwait /= 0, :239-248); isd travels as par[0] (an integer below 86400: exact)
        const R zalpha = alpha_sw<R>(x[3]);
        WlCoareCell<R> wc;
        wc.zcd1 = M::sqrt(M::div(R(2.) * R(0.65) * K<R>::rCp0_w, zalpha * K<R>::grav * K<R>::rho0_w));     // :155
        wc.zcd2 = M::sqrt(R(2.) * zalpha * K<R>::grav * R(1. / (0.65 * 1025.))) * R(1. / 271219.5770957547);   // :156
        wc.dawn = wl_coare_dawn<R>(x[4], (int)par[0]);
        R st[4] = {x[5], x[6], x[7], x[8]};
        wl_coare<R, false>(st, wc, x[0], x[1], x[2], flag == 0);
        y[0] = st[0]; y[1] = st[1]; y[2] = st[2]; y[3] = st[3];
    } else if constexpr (FN == kPhWlEcmwf) {         // WL_ECMWF, mod_skin_ecmwf.f90:113-230  ( pQsw, pQnsol, pustar, pSST, dT_wl, Hz_wl, [pustk] ) -> dT_wl
        R dT = x[4];
        wl_ecmwf<R>(dT, x[5], wl_ecmwf_cell<R>(x[5]), x[0], x[1], x[2], alpha_sw<R>(x[3]), (present & 0x40u) ? vmax(x[6], R(0.)) : R(-1.));
        y[0] = dT;
    }
}

// number of array inputs / outputs of each function (0: unknown id); the tables of ab_phymbl.hip and of the host harness
struct PhShape { int n_in, n_in_required, n_out; };
constexpr PhShape ph_shape(int fn)
{
    switch (fn) {
    case kPhPotTemp: case kPhAbsTemp: return {3, 2, 1};
    case kPhVirtTemp: return {2, 2, 1};
    case kPhPzFromP0: case kPhThetaFromZ: case kPhTFromZ: return {3, 3, 1};
    case kPhRhoAir: return {3, 3, 1};
    case kPhViscAir: case kPhLvap: case kPhCpAir: return {1, 1, 1};
    case kPhGammaMoist: return {2, 2, 1};
    case kPhOneOnL: return {5, 5, 1};
    case kPhRiBulk: return {7, 5, 1};
    case kPhEsat: case kPhEsatIce: case kPhDEsatDtIce: return {1, 1, 1};
    case kPhQsat: case kPhDQsatDtIce: return {2, 2, 1};
    case kPhQairRh: return {3, 3, 1};
    case kPhQairDp: return {2, 2, 1};
    case kPhRhoAirAdv: return {3, 3, 1};
    case kPhQsatCrude: return {2, 2, 1};
    case kPhDryStaticEnergy: return {2, 2, 1};
    case kPhUpdateQnsolTau: return {11, 11, 3};
    case kPhBulkFormula: return {10, 10, 5};
    case kPhAlphaSw: return {1, 1, 1};
    case kPhQlwNet: return {2, 2, 1};
    case kPhZ0FromCd: case kPhCdFromZ0: return {2, 1, 1};
    case kPhZ0FromUstar: return {2, 2, 1};
    case kPhFmLouis: case kPhFhLouis: return {3, 3, 1};
    case kPhUN10FromUstar: case kPhUN10FromCdn: case kPhUN10FromCd: return {3, 3, 1};
    case kPhZ0tqLkb: return {2, 2, 1};
    case kPhEair: return {2, 2, 1};
    case kPhRhAir: return {3, 3, 1};
    case kPhDeltaSkinLayer: return {4, 3, 1};
    case kPhRoughLengM: return {2, 2, 1};
    case kPhRoughLengTq: return {3, 3, 2};
    case kPhPsiMCoare: case kPhPsiHCoare: case kPhPsiMNcar: case kPhPsiHNcar: case kPhPsiMEcmwf: case kPhPsiHEcmwf:
    case kPhPsiMAndreas: case kPhPsiHAndreas: case kPhCharnCoare3p0: case kPhCharnCoare3p6: case kPhCdN10Ncar: case kPhCeN10Ncar:
    case kPhUStarAndreas: return {1, 1, 1};
    case kPhChN10Ncar: return {2, 2, 1};
    case kPhFirstGuessCoare: return {6, 6, 7};
    case kPhCsCoare: return {5, 5, 1};
    case kPhCsEcmwf: return {4, 4, 1};
    case kPhWlCoare: return {9, 9, 4};
    case kPhWlEcmwf: return {7, 6, 1};
    default: return {0, 0, 0};
    }
}

}  // namespace ab
