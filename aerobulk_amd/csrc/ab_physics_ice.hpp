// ab_physics_ice.hpp — per-cell device physics of the sea-ice bulk algorithms.
//
// TURB_ICE_NEMO (src/ice/mod_blk_ice_nemo.f90:36-153), TURB_ICE_AN05 (mod_blk_ice_an05.f90:41-405, Andreas et al. 2005),
// TURB_ICE_LU12 (mod_blk_ice_lu12.f90:69-214 with CdN10_f_LU13 of mod_cdn_form_ice.f90:147-191, Lupkes et al. 2012/13),
// TURB_ICE_LG15 (mod_blk_ice_lg15.f90:68-307 with CdN_f_LG15_light of mod_cdn_form_ice.f90:272-307 and the Louis functions
// of mod_phymbl.f90:1419-1479, Lupkes & Gryanik 2015).  Same conventions as ab_physics.hpp; include after it.
#pragma once
#include "ab_physics.hpp"

namespace ab {

template <class R> struct KIce {
    static constexpr R wspd_thrshld = R(0.2);      // mod_const.f90:120
    static constexpr R rCd_ice = R(1.4e-3);        // mod_const.f90:118
    static constexpr R rz0_i_s_0 = R(0.69e-3);     // mod_blk_ice_lg15.f90:56
    static constexpr R rz0_i_f_0 = R(4.54e-4);     // mod_blk_ice_lg15.f90:57
    static constexpr R ralpha_0 = R(0.2);          // mod_blk_ice_lg15.f90:54
    static constexpr R rce10_i_0 = R(3.46e-3);     // mod_cdn_form_ice.f90:34
    static constexpr R rbeta_0 = R(1.4);           // mod_cdn_form_ice.f90:25
    static constexpr R rCe_0 = R(2.23E-3);         // mod_cdn_form_ice.f90:22
};

template <class R> struct IceIn {
    R Ts_i, theta_zt, qs_i, q_zt, wnd, frice;
};
template <class R> struct IceOut {
    R Cd, Ch, Ce, t_zu, q_zu, Ub, CdN, ChN, CeN, z0, us, L, UN10;
    R CdN_frm;   // LG15 only: form-drag part of CdN (TURB_ICE_LG15_IO's optional output)
};

// first lines of every TURB_ICE_*: wind threshold, floors on t/q, non-zero air-ice differences (e.g. an05 :123-132)
template <class R>
__device__ __forceinline__ void ice_first_guess(const IceIn<R> &in, R &Ub, R &t_zu, R &q_zu, R &dt_zu, R &dq_zu)
{
    Ub = vmax(in.wnd, KIce<R>::wspd_thrshld);
    t_zu = vmax(in.theta_zt, R(100.));
    q_zu = vmax(in.q_zt, R(0.1e-6));
    dt_zu = sfloor(t_zu - in.Ts_i, R(1.E-6));
    dq_zu = sfloor(q_zu - in.qs_i, R(1.E-9));
}

// coefficients that do not depend on stability (NEMO: 1.4e-3 ; LU12: skin + form drag): mod_blk_ice_nemo.f90:123-145
template <class R> __device__ __forceinline__ void turb_ice_const(const Heights<R> &h, const IceIn<R> &in, R cd, IceOut<R> &o)
{
    using M = Mth<R>;
    R Ub, t_zu, q_zu, dt_zu, dq_zu;
    ice_first_guess(in, Ub, t_zu, q_zu, dt_zu, dq_zu);
    o.Cd = cd; o.Ch = cd; o.Ce = cd; o.t_zu = t_zu; o.q_zu = q_zu; o.Ub = Ub;
    o.CdN = cd; o.ChN = cd; o.CeN = cd;
    const R sq = M::sqrt(cd);
    const R lz0 = h.log_zu - M::div(K<R>::vkarmn, sq);          // ln z0, z0 = zu exp(-kappa/sqrt(Cd))  mod_phymbl.f90:1349
    o.z0 = M::exp(lz0);
    o.us = sq * Ub;
    const R cs = M::div(cd, sq);
    o.L = M::rcp(one_on_l(t_zu, q_zu, sq * Ub, cs * dt_zu, cs * dq_zu));
    o.UN10 = sq * Ub * K<R>::inv_vk * (h.log_10 - lz0);
}
// CdN10_f_LU13 mod_cdn_form_ice.f90:147-191 (rMu_0 = rNu_0 = 1; rBeta_0 is rbeta_0 = 1.4: Fortran is case-blind)
template <class R> __device__ __forceinline__ R cdn10_f_lu13(R frice)
{
    return KIce<R>::rCe_0 * pow_pos(R(1.) - frice, R(1. + 1. / 14.));
}

// ---------------------------------------------------------------- Andreas et al. 2005
// rough_leng_m :232-255
template <class R> __device__ __forceinline__ R an05_rough_leng_m(R pus, R pnua)
{
    using M = Mth<R>;
    const R zus = vmax(pus, R(1.E-9));
    const R zz = (zus - R(0.18)) * R(10.);
    return M::div(R(0.135) * pnua, zus) + R(0.035 / 9.8) * zus * zus * (R(5.) * M::exp(-zz * zz) + R(1.));
}
// rough_leng_tq :257-312, returns LOG(z0t), LOG(z0q) (only the logs are used, :181-182,208-209)
template <class R> __device__ __forceinline__ void an05_log_z0tq(R pz0, R log_z0, R pus, R pnua, R &lz0t, R &lz0q)
{
    using M = Mth<R>;
    const R zus = vmax(pus, R(1.E-9));
    const R zre = vmax(M::div(zus * pz0, pnua), R(0.));
    const bool smooth = nonneg(R(0.135) - zre), rough = nonneg(zre - R(2.5));
    const bool trans = !smooth && nonneg(R(2.49999) - zre);
    const R zlog = M::log(zre);
    const R zlog2 = zlog * zlog;
    // Table 1 of Andreas et al. 2005; a Reynolds number in (2.49999, 2.5) belongs to no regime (all weights 0), like the reference
    R b0 = smooth ? R(1.25) : (trans ? R(0.149) : (rough ? R(0.317) : R(0.)));
    R b1 = trans ? R(-0.550) : (rough ? R(-0.565) : R(-0.));
    R b2 = rough ? R(-0.183) : R(-0.);
    lz0t = log_z0 + (b0 + b1 * zlog + b2 * zlog2);
    b0 = smooth ? R(1.61) : (trans ? R(0.351) : (rough ? R(0.396) : R(0.)));
    b1 = trans ? R(-0.628) : (rough ? R(-0.512) : R(-0.));
    b2 = rough ? R(-0.180) : R(-0.);
    lz0q = log_z0 + (b0 + b1 * zlog + b2 * zlog2);
}
// psi_m_ice :316-360, psi_h_ice :363-405 (Paulson 1970 / Holtslag & De Bruin 1988)
template <class R> __device__ __forceinline__ void psi_ice(R zta, R *pm, R *ph)
{
    using M = Mth<R>;
    if (nonneg(zta)) {
        const R s = -(R(0.7) * zta + R(0.75) * (zta - R(14.3)) * M::exp(R(-0.35) * zta) + R(10.7));
        if (pm) *pm = s;
        if (ph) *ph = s;
    } else {
        const R x2 = M::sqrt_pos(M::abs(R(1.) - R(16.) * zta));
        const R l2 = M::log((R(1.) + x2) * R(0.5));
        if (ph) *ph = R(2.) * l2;
        if (pm) {
            const R x = M::sqrt_pos(x2);
            *pm = l2 + R(2.) * M::log((R(1.) + x) * R(0.5)) - R(2.) * M::atan_ge1(x) + R(0.5) * K<R>::rpi;
        }
    }
}
template <class R>
__device__ __forceinline__ void turb_ice_an05(const Heights<R> &h, const IceIn<R> &in, int nb_iter, IceOut<R> &o)
{
    using M = Mth<R>;
    const R vk = K<R>::vkarmn;
    R Ubzu, t_zu, q_zu, dt_zu, dq_zu;
    ice_first_guess(in, Ubzu, t_zu, q_zu, dt_zu, dq_zu);
    const R znu_a = visc_air(t_zu);
    R z0 = R(8.0E-4);
    R u_star = R(0.035) * Ubzu * M::div(h.log_10 - R(-7.1308988302963465), h.log_zu - R(-7.1308988302963465));  // ln(8e-4)
    z0 = an05_rough_leng_m(u_star, znu_a);
    R lz0 = M::log(z0);
#pragma unroll 1
    for (int jit = 1; jit <= 2; ++jit) {                               // :147-150
        u_star = vmax(M::div(Ubzu * vk, h.log_zu - lz0), R(1.E-9));
        z0 = an05_rough_leng_m(u_star, znu_a);
        lz0 = M::log(z0);
    }
    R lz0t, lz0q;
    an05_log_z0tq(z0, lz0, u_star, znu_a, lz0t, lz0q);
    R t_star = M::div(dt_zu * vk, h.log_zu - lz0t);
    R q_star = M::div(dq_zu * vk, h.log_zu - lz0q);
#pragma unroll 1
    for (int jit = 1; jit <= nb_iter; ++jit) {
        const R z1oL = one_on_l(t_zu, q_zu, u_star, t_star, q_star);   // clamp +-200 inside (:162)
        const R zeta_u = sclamp(h.zu * z1oL, R(50.));
        z0 = an05_rough_leng_m(u_star, znu_a);
        lz0 = M::log(z0);
        an05_log_z0tq(z0, lz0, u_star, znu_a, lz0t, lz0q);
        R psm, psh;
        psi_ice<R>(zeta_u, &psm, &psh);
        t_star = M::div(dt_zu * vk, h.log_zu - lz0t - psh);
        q_star = M::div(dq_zu * vk, h.log_zu - lz0q - psh);
        u_star = vmax(M::div(Ubzu * vk, h.log_zu - lz0 - psm), R(1.E-9));
        if (!h.zt_eq_zu) {
            const R zeta_t = sclamp(h.zt * z1oL, R(50.));
            R psht;
            psi_ice<R>(zeta_t, nullptr, &psht);
            const R ztmp1 = h.log_ztu + psh - psht;
            t_zu = in.theta_zt - t_star * K<R>::inv_vk * ztmp1;
            q_zu = in.q_zt - q_star * K<R>::inv_vk * ztmp1;
            dt_zu = sfloor(t_zu - in.Ts_i, R(1.E-6));
            dq_zu = sfloor(q_zu - in.qs_i, R(1.E-9));
        }
    }
    const R ztmp0 = M::div(u_star, Ubzu);
    o.Cd = ztmp0 * ztmp0;
    o.Ch = M::div(ztmp0 * t_star, dt_zu);
    o.Ce = M::div(ztmp0 * q_star, dq_zu);
    o.t_zu = t_zu; o.q_zu = q_zu; o.Ub = Ubzu;
    const R zi = M::rcp(h.log_zu - lz0);
    o.CdN = K<R>::vkarmn2 * zi * zi;
    o.ChN = M::div(K<R>::vkarmn2 * zi, h.log_zu - lz0t);
    o.CeN = M::div(K<R>::vkarmn2 * zi, h.log_zu - lz0q);
    o.z0 = z0; o.us = u_star;
    o.L = M::rcp(one_on_l(t_zu, q_zu, u_star, t_star, q_star));
    o.UN10 = u_star * K<R>::inv_vk * (h.log_10 - lz0);
}

// turb_ice_easy mod_blk_ice_easy.f90:44-209: stability correction (psi of Andreas 2005) of PRESCRIBED neutral coefficients
template <class R>
__device__ __forceinline__ void turb_ice_easy(const Heights<R> &h, const IceIn<R> &in, R CdN, R ChN, R CeN, int nb_iter, IceOut<R> &o)
{
    using M = Mth<R>;
    const R zsqrtCDN = M::sqrt(CdN), zi_sq = M::rcp(zsqrtCDN);
    const R Ubzu = vmax(in.wnd, KIce<R>::wspd_thrshld);
    R t_zu = vmax(in.theta_zt, R(100.)), q_zu = vmax(in.q_zt, R(0.1e-6));
    R Cd_i = CdN, Ch_i = ChN, Ce_i = CeN;
    R u_star = R(0.), t_star = R(0.), q_star = R(0.), psm = R(0.);
#pragma unroll 1
    for (int jit = 1; jit <= nb_iter; ++jit) {
        const R dt_zu = t_zu - in.Ts_i, dq_zu = q_zu - in.qs_i;          // no floor here (:147-148)
        const R sq = M::sqrt(Cd_i);
        u_star = sq * Ubzu;
        const R isq = M::rcp(vmax(sq, R(1.E-15)));
        t_star = Ch_i * dt_zu * isq;
        q_star = Ce_i * dq_zu * isq;
        const R z1oL = one_on_l(t_zu, q_zu, u_star, t_star, q_star);
        const R zeta_u = sclamp(h.zu * z1oL, R(50.));
        R psh;
        psi_ice<R>(zeta_u, &psm, &psh);
        const R a = R(1.) + zsqrtCDN * K<R>::inv_vk * (h.log_zu10 - psm);
        Cd_i = vmin(vmax(M::div(CdN, a * a), K<R>::Cx_min), R(1.9E-3));
        const R b = (h.log_zu10 - psh) * K<R>::inv_vk * zi_sq;
        const R c = M::sqrt(Cd_i) * zi_sq;
        Ch_i = vmin(vmax(M::div(ChN * c, R(1.) + ChN * b), K<R>::Cx_min), R(1.9E-3));
        Ce_i = vmin(vmax(M::div(CeN * c, R(1.) + CeN * b), K<R>::Cx_min), R(1.9E-3));
        if (!h.zt_eq_zu) {
            R psht;
            psi_ice<R>(sclamp(h.zt * z1oL, R(50.)), nullptr, &psht);
            const R d = psh - psht + h.log_ztu;
            t_zu = in.theta_zt - t_star * K<R>::inv_vk * d;
            q_zu = vmax(R(0.), in.q_zt - q_star * K<R>::inv_vk * d);
        }
    }
    o.Cd = Cd_i; o.Ch = Ch_i; o.Ce = Ce_i; o.t_zu = t_zu; o.q_zu = q_zu; o.Ub = Ubzu;
    o.CdN = CdN; o.ChN = ChN; o.CeN = CeN;
    const R sq = M::sqrt(Cd_i);
    const R lz0 = h.log_zu - (M::div(K<R>::vkarmn, sq) + psm);          // z0_from_Cd with psi, mod_phymbl.f90:1346
    o.z0 = M::exp(lz0);
    o.us = u_star;
    o.L = M::rcp(one_on_l(t_zu, q_zu, u_star, t_star, q_star));
    o.UN10 = sq * Ubzu * K<R>::inv_vk * (h.log_10 - lz0);               // UN10_from_CD :1545
}

// ---------------------------------------------------------------- Lupkes & Gryanik 2015
// f_m_louis_sclr / f_h_louis_sclr mod_phymbl.f90:1419-1479 (rc_louis = 5: 3 c^2 = 75, a_m = 10, a_h = 15)
template <class R> __device__ __forceinline__ R f_louis(R z_o_z0_p1, R pRib, R pCxn, R ra)
{
    using M = Mth<R>;
    if (nonneg(pRib)) return M::rcp(R(1.) + ra * pRib * M::rsqrt_pos(M::abs(R(1.) + pRib)));
    const R ztu = M::div(pRib, R(1.) + R(75.) * pCxn * M::sqrt_pos(M::abs(-pRib * z_o_z0_p1)));
    return R(1.) - ra * ztu;
}
// frice_last: CdN_f_LG15_light (mod_cdn_form_ice.f90:272-307) assigns its WHOLE result array inside its cell loop, so every
// cell of a call ends up with the form drag of the LAST cell of the array.  Reproduced.
template <class R>
__device__ __forceinline__ void turb_ice_lg15(const Heights<R> &h, const IceIn<R> &in, R frice_last, int nb_iter, IceOut<R> &o)
{
    using M = Mth<R>;
    const R vk = K<R>::vkarmn;
    R Ubzu, t_zu, q_zu, dt_zu, dq_zu;
    ice_first_guess(in, Ubzu, t_zu, q_zu, dt_zu, dq_zu);
    const R lz0s = R(-7.278818960372969);                              // ln(0.69e-3)
    const R lz0f = R(-7.697413359922926);                              // ln(4.54e-4)
    const R zis = M::rcp(h.log_zu - lz0s);
    const R zCdN_s = K<R>::vkarmn2 * zis * zis;                        // Cd_from_z0 mod_phymbl.f90:1396-1414
    const R zChN_s = M::div(K<R>::vkarmn2, (h.log_zu - lz0s) * (h.log_zu - (R(-1.6094379124341003) + lz0s)));  // ln(0.2)
    const R zrlog = M::div(h.log_10 - lz0f, h.log_zu - lz0f);
    const R zCdN_f = KIce<R>::rce10_i_0 * zrlog * zrlog * frice_last * pow_pos(R(1.) - frice_last, KIce<R>::rbeta_0);
    const R zChN_f = M::div(zCdN_f, R(1.) + R(1.6094379124341003) * K<R>::inv_vk * M::sqrt(zCdN_f));   // ln(1/0.2)/kappa
    const R zCdN = zCdN_s + zCdN_f, zz0 = KIce<R>::rz0_i_s_0 + KIce<R>::rz0_i_f_0;
    const R zu_s = h.zu * R(1. / 0.69e-3) + R(1.), zu_f = h.zu * R(1. / 4.54e-4) + R(1.);
    const R zu_t = M::div(h.zu, zz0) + R(1.), zt_t = M::div(h.zt, zz0) + R(1.);
    R Cd_i = zCdN, Ch_i = zChN_s + zChN_f;
    R RiB = ri_bulk(h.zt, in.Ts_i, in.theta_zt, in.qs_i, in.q_zt, Ubzu);
#pragma unroll 1
    for (int jit = 1; jit <= nb_iter; ++jit) {
        R wnd_zt = Ubzu;
        if (!h.zt_eq_zu) {                                             // wind brought to zt :192-201
            const R x1 = h.log_ztu + f_louis(zu_t, RiB, zCdN, R(15.)) - f_louis(zt_t, RiB, zCdN, R(15.));
            wnd_zt = vmin(vmax(Ubzu + (M::sqrt(Cd_i) * Ubzu) * x1, KIce<R>::wspd_thrshld), Ubzu);
        }
        RiB = ri_bulk(h.zt, in.Ts_i, in.theta_zt, in.qs_i, in.q_zt, wnd_zt);
        Cd_i = zCdN_s * f_louis(zu_s, RiB, zCdN_s, R(10.)) + zCdN_f * f_louis(zu_f, RiB, zCdN_f, R(10.));   // :209,219
        Ch_i = zChN_s * f_louis(zu_s, RiB, zCdN_s, R(15.)) + zChN_f * f_louis(zu_f, RiB, zCdN_f, R(15.));   // :210,220
        if (!h.zt_eq_zu) {                                             // :233-250
            const R x1 = h.log_ztu + f_louis(zu_t, RiB, zCdN, R(15.)) - f_louis(zt_t, RiB, zCdN, R(15.));
            const R x2 = M::rsqrt_pos(Cd_i);
            t_zu = in.theta_zt - (Ch_i * dt_zu * x2) * K<R>::inv_vk * x1;
            q_zu = vmax(R(0.), in.q_zt - (Ch_i * dq_zu * x2) * K<R>::inv_vk * x1);
            dt_zu = sfloor(t_zu - in.Ts_i, R(1.E-6));
            dq_zu = sfloor(q_zu - in.qs_i, R(1.E-9));
        }
    }
    o.Cd = Cd_i; o.Ch = Ch_i; o.Ce = Ch_i; o.t_zu = t_zu; o.q_zu = q_zu; o.Ub = Ubzu;
    o.CdN = zCdN; o.ChN = zChN_s + zChN_f; o.CeN = o.ChN; o.CdN_frm = zCdN_f;
    const R lz0 = h.log_zu - M::div(vk, M::sqrt(zCdN));
    o.z0 = M::exp(lz0);
    const R sq = M::sqrt(Cd_i);
    o.us = sq * Ubzu;
    o.L = M::rcp(one_on_l(t_zu, q_zu, sq * Ubzu, M::div(Ch_i * dt_zu, sq), M::div(Ch_i * dq_zu, sq)));
    o.UN10 = sq * Ubzu * K<R>::inv_vk * (h.log_10 - lz0);
}

}  // namespace ab
