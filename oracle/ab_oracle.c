/* oracle/ab_oracle.c — TEST INFRASTRUCTURE ONLY (see ab_oracle.h).
 *
 * Literal, scalar, fp64 restatement of the reference's per-cell arithmetic.
 * Rules followed throughout:
 *   - every decimal literal is the double nearest to its decimal string (the reference
 *     is always built with -fdefault-real-8 / -r8, arch/make.macro_GnuLinux:17);
 *   - `x**y` with a REAL exponent is pow(x,y) (what flang emits), `x**2` is x*x;
 *   - expressions are evaluated left to right exactly as written in the reference;
 *   - Fortran SIGN(a,b) == copysign(a,b) (IEEE signed-zero aware, as flang/gfortran).
 * Compile with -ffp-contract=off (no FMA contraction), see oracle/Makefile.
 */
#include "ab_oracle.h"
#include <math.h>
#include <stddef.h>

/* ---- constants: mod_const.f90:38-114 ------------------------------------------------ */
static const double grav = 9.8;               /* :38 */
static const double rpi = 3.141592653589793;  /* :39 */
static const double roce_alb0 = 0.066;        /* :49 */
static const double emiss_w = 0.98;           /* :55 */
static const double stefan = 5.67E-8;         /* :57 */
static const double rt0 = 273.15;             /* :60 */
static const double rCp0_w = 4190.;           /* :63 */
static const double rho0_w = 1025.;           /* :64 */
static const double rnu0_w = 1.e-6;           /* :65 */
static const double rk0_w = 0.6;              /* :66 */
static const double rCp_dry = 1005.0;         /* :71 */
static const double rCp_vap = 1860.0;         /* :72 */
static const double R_dry = 287.05;           /* :74 */
static const double R_vap = 461.495;          /* :75 */
static const double R_gas = 8.314510;         /* :76 */
static const double rmm_dryair = 28.9647e-3;  /* :78 */
static const double rmm_water = 18.0153e-3;   /* :79 */
static const double rLevap = 2.46e+6;         /* :91 */
static const double rho0_a = 1.2;             /* :99 */
static const double vkarmn = 0.4;             /* :103 */
static const double rdct_qsat_salt = 0.98;    /* :105 */
static const double z0_sea_max = 0.0025;      /* :106 */
static const double Cx_min = 0.1E-3;          /* :114 */
static const double rdt = 3600.;              /* :32 */
static const double gdept1 = 1.;              /* :31 gdept_1d(1) */
#define vkarmn2 (0.4 * 0.4)                                   /* :104 */
#define rpoiss_dry (R_dry / rCp_dry)                          /* :82 */
#define rgamma_dry (grav / rCp_dry)                           /* :83 */
#define reps0 (R_dry / R_vap)                                 /* :86 */
#define rctv0 (R_vap / R_dry - 1.)                            /* :87 */
/* :109  -16*9.80665*rho0_w*rCp0_w*rnu0_w^3/(rk0_w^2), left to right */
#define rcst_cs (-16. * 9.80665 * rho0_w * rCp0_w * rnu0_w * rnu0_w * rnu0_w / (rk0_w * rk0_w))

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double fsign(double a, double b) { return copysign(a, b); }

/* ---- mod_phymbl.f90 helpers -------------------------------------------------------- */

/* e_sat_sclr, mod_phymbl.f90:777-800 (Goff 1957) */
double abo_e_sat(double pTa)
{
    double zta = dmax(pTa, 180.);
    double ztmp = rt0 / zta;
    return 100. * (pow(10., 10.79574 * (1. - ztmp) - 5.028 * log10(zta / rt0)
                       + 1.50475 * 1e-4 * (1. - pow(10., -8.2969 * (zta / rt0 - 1.)))
                       + 0.42873 * 1e-3 * (pow(10., 4.76955 * (1. - ztmp)) - 1.) + 0.78614));
}

/* q_sat_sclr, mod_phymbl.f90:881-904 (l_ice never set on this path) */
double abo_q_sat(double pTa, double pslp)
{
    double ze_s = abo_e_sat(pTa);
    return reps0 * ze_s / (pslp - (1. - reps0) * ze_s);
}

/* q_air_rh, mod_phymbl.f90:963-985 */
double abo_q_air_rh(double prha, double pTa, double pslp)
{
    double ze = 0.01 * prha * abo_e_sat(pTa);
    return ze * reps0 / dmax(pslp - (1. - reps0) * ze, 1.);
}

/* q_air_dp, mod_phymbl.f90:990-1000 */
double abo_q_air_dp(double da, double slp)
{
    double q = dmax(abo_e_sat(da), 0.);
    return q * reps0 / dmax(slp - (1. - reps0) * q, 1.);
}

/* Pz_from_P0_tz_qz_sclr, mod_phymbl.f90:283-318 */
static double pz_from_p0_tz_qz(double pz, double pslp, double pTa, double pqa)
{
    double zpa = pslp;
    for (int it = 0; it < 3; ++it) {
        double zqsat = abo_q_sat(pTa, zpa);
        double zf = pqa / zqsat;
        double zxm = (1. - zf) * rmm_dryair + zf * rmm_water;
        zpa = pslp * exp(-grav * zxm * pz / (R_gas * pTa));
    }
    return zpa;
}

/* Theta_from_z_P0_T_q_vctr :367-375 -> pot_temp_vctr :189-200 (pPref present) */
double abo_theta_from_z_p0_t_q(double pz, double pslp, double pTa, double pqa)
{
    double zPz = pz_from_p0_tz_qz(pz, pslp, pTa, pqa);
    return pTa * pow(pslp / zPz, rpoiss_dry);
}

/* virt_temp_sclr, mod_phymbl.f90:247-269 */
static inline double virt_temp(double pTa, double pqa) { return pTa * (1. + rctv0 * pqa); }

/* rho_air_sclr, mod_phymbl.f90:522-537 */
double abo_rho_air(double pTa, double pqa, double pslp)
{
    return dmax(pslp / (R_dry * pTa * (1. + rctv0 * pqa)), 0.8);
}

/* visc_air_sclr, mod_phymbl.f90:549-563 */
double abo_visc_air(double pTa)
{
    double ztc = pTa - rt0;
    double ztc2 = ztc * ztc;
    return 1.326e-5 * (1. + 6.542E-3 * ztc + 8.301e-6 * ztc2 - 4.84e-9 * ztc2 * ztc);
}

/* L_vap_sclr :579-592 ; cp_air_sclr :603-616 */
static inline double L_vap(double psst) { return (2.501 - 0.00237 * (psst - rt0)) * 1.e6; }
static inline double cp_air(double pqa) { return rCp_dry + rCp_vap * pqa; }

/* One_on_L_sclr, mod_phymbl.f90:666-693 */
double abo_one_on_l(double pThta, double pqa, double pus, double pts, double pqs)
{
    double zqa = (1. + rctv0 * pqa);
    double r = grav * vkarmn * (pts * zqa + rctv0 * pThta * pqs) / dmax(pus * pus * pThta * zqa, 1.E-9);
    return fsign(dmin(fabs(r), 200.), r);
}

/* Ri_bulk_sclr, mod_phymbl.f90:712-747 (layer arguments never passed: SURVEY §5) */
double abo_ri_bulk(double pz, double psst, double pThta, double pssq, double pqa, double pub)
{
    double zsstv = virt_temp(psst, pssq);
    double zdthv = virt_temp(pThta, pqa) - zsstv;
    double ztv = 0.5 * (zsstv + virt_temp(pThta - rgamma_dry * pz, pqa));
    return grav * zdthv * pz / (ztv * pub * pub);
}

/* qlw_net_sclr, mod_phymbl.f90:1291-1314 */
static inline double qlw_net(double pdwlw, double pts)
{
    double zt2 = pts * pts;
    return emiss_w * (pdwlw - stefan * zt2 * zt2);
}

/* BULK_FORMULA_SCLR, mod_phymbl.f90:1149-1203 (l_ice false) */
static void bulk_formula(double pzu, double pts, double pqs, double pThta, double pqa,
                         double pCd, double pCh, double pCe, double pwnd, double pUb, double pslp,
                         double *pTau, double *pQsen, double *pQlat, double *pEvap)
{
    double zta = pThta - rgamma_dry * pzu;
    double zrho = abo_rho_air(zta, pqa, pslp);
    zrho = abo_rho_air(zta, pqa, pslp - zrho * grav * pzu);
    double zUrho = pUb * dmax(zrho, 1.);
    *pTau = zUrho * pCd * pwnd;
    double zevap = zUrho * pCe * (pqa - pqs);
    *pQsen = zUrho * pCh * (pThta - pts) * cp_air(pqa);
    *pQlat = L_vap(pts) * zevap;
    if (pEvap) *pEvap = zevap;
}

/* UPDATE_QNSOL_TAU_SCLR, mod_phymbl.f90:1059-1103 */
static void update_qnsol_tau(double pzu, double pts, double pqs, double pThta, double pqa,
                             double pust, double ptst, double pqst, double pwnd, double pUb,
                             double pslp, double prlw, double *pQns, double *pTau, double *Qlat)
{
    double zdt = pThta - pts;  zdt = fsign(dmax(fabs(zdt), 1.E-09), zdt);
    double zdq = pqa - pqs;    zdq = fsign(dmax(fabs(zdq), 1.E-12), zdq);
    double zz0 = pust / pUb;
    double zCd = zz0 * zz0;
    double zCh = zz0 * ptst / zdt;
    double zCe = zz0 * pqst / zdq;
    double zQsen, zQlat;
    bulk_formula(pzu, pts, pqs, pThta, pqa, zCd, zCh, zCe, pwnd, pUb, pslp, pTau, &zQsen, &zQlat, NULL);
    double zQlw = qlw_net(prlw, pts);
    *pQns = zQlat + zQsen + zQlw;
    if (Qlat) *Qlat = zQlat;
}

/* alpha_sw_sclr, mod_phymbl.f90:1267-1280 */
double abo_alpha_sw(double psst) { return 2.1e-5 * pow(dmax(psst - rt0 + 3.2, 0.), 0.79); }

/* delta_skin_layer_sclr, mod_phymbl.f90:2010-2046 */
double abo_delta_skin_layer(double palpha, double pQd, double pustar_a, int with_qlat, double Qlat)
{
    const double sq_radrw = sqrt(rho0_a / rho0_w); /* mod_const.f90:112 */
    double zQd = pQd;
    if (with_qlat) zQd = pQd + 0.026 * dmin(Qlat, 0.) * rCp0_w / rLevap / palpha;
    double ztf = 0.5 + fsign(0.5, zQd);
    double zusw = dmax(pustar_a, 1.E-4) * sq_radrw;
    double zusw2 = zusw * zusw;
    double zlamb = 6. * pow(1. + pow(dmax(palpha * rcst_cs / (zusw2 * zusw2) * zQd, 0.), 0.75), (-1. / 3.));
    double ztmp = rnu0_w / zusw;
    return (1. - ztf) * zlamb * ztmp + ztf * dmin(6. * ztmp, 0.007);
}

/* z0_from_Cd_sclr with ppsi, mod_phymbl.f90:1335-1352 */
static inline double z0_from_cd_psi(double pzu, double pCd, double ppsi)
{
    return pzu * exp(-(vkarmn / sqrt(pCd) + ppsi));
}

/* UN10_from_CD_sclr, mod_phymbl.f90:1532-1547 */
static inline double un10_from_cd(double pzu, double pUb, double pCd, double ppsi)
{
    return sqrt(pCd) * pUb / vkarmn * log(10. / z0_from_cd_psi(pzu, pCd, ppsi));
}

/* z0tq_LKB, mod_phymbl.f90:1635-1701 (per element) */
double abo_z0tq_lkb(int iflag, double zrr, double pz0)
{
    static const double XA[2][8] = {
        {0.177, 1.376, 1.026, 1.625, 4.661, 34.904, 1667.19, 5.88e5},
        {0.292, 1.808, 1.393, 1.956, 4.994, 30.709, 1448.68, 2.98e5}};
    static const double XB[2][8] = {
        {0., 0.929, -0.599, -1.018, -1.475, -2.067, -2.907, -3.935},
        {0., 0.826, -0.528, -0.870, -1.297, -1.845, -2.682, -3.616}};
    static const double XRAN[9] = {0., 0.11, 0.825, 3.0, 10.0, 30.0, 100., 300., 1000.};
    double r = -999.;
    if ((zrr > 0.) && (zrr < 1000.)) {
        int jm = 0, found = 0;
        while (!found) {
            jm = jm + 1;
            found = ((zrr > XRAN[jm - 1]) && (zrr <= XRAN[jm]));
        }
        r = XA[iflag - 1][jm - 1] * pow(zrr, XB[iflag - 1][jm - 1]) * pz0 / zrr;
    }
    return dmin(dmax(fabs(r), 1.E-9), 0.05);
}

/* ---- mod_common_coare.f90 ---------------------------------------------------------- */

/* psi_m_coare_sclr, mod_common_coare.f90:217-254 */
double abo_psi_m_coare(double pzeta)
{
    double zphi_m = pow(fabs(1. - 15. * pzeta), .25);
    double zpsi_k = 2. * log((1. + zphi_m) / 2.) + log((1. + zphi_m * zphi_m) / 2.) - 2. * atan(zphi_m) + 0.5 * rpi;
    double zphi_c = pow(fabs(1. - 10.15 * pzeta), .3333);
    double zpsi_c = 1.5 * log((1. + zphi_c + zphi_c * zphi_c) / 3.) - 1.7320508 * atan((1. + 2. * zphi_c) / 1.7320508) + 1.813799447;
    double zf = pzeta * pzeta;
    zf = zf / (1. + zf);
    double zc = dmin(50., 0.35 * pzeta);
    double zstb = 0.5 + fsign(0.5, pzeta);
    return (1. - zstb) * ((1. - zf) * zpsi_k + zf * zpsi_c)
           - zstb * (1. + 1. * pzeta + 0.6667 * (pzeta - 14.28) / exp(zc) + 8.525);
}

/* psi_h_coare_sclr, mod_common_coare.f90:305-344 */
double abo_psi_h_coare(double pzeta)
{
    double zphi_h = pow(fabs(1. - 15. * pzeta), .5);
    double zpsi_k = 2. * log((1. + zphi_h) / 2.);
    double zphi_c = pow(fabs(1. - 34.15 * pzeta), .3333);
    double zpsi_c = 1.5 * log((1. + zphi_c + zphi_c * zphi_c) / 3.) - 1.7320508 * atan((1. + 2. * zphi_c) / 1.7320508) + 1.813799447;
    double zf = pzeta * pzeta;
    zf = zf / (1. + zf);
    double zc = dmin(50., 0.35 * pzeta);
    double zstb = 0.5 + fsign(0.5, pzeta);
    return (1. - zstb) * ((1. - zf) * zpsi_k + zf * zpsi_c)
           - zstb * (pow(fabs(1. + 2. * pzeta / 3.), 1.5) + .6667 * (pzeta - 14.28) / exp(zc) + 8.525);
}

/* FIRST_GUESS_COARE_SCLR, mod_common_coare.f90:33-179 */
static void first_guess_coare(double zt, double zu, double psst, double t_zt, double pssq, double q_zt,
                              double U_zu, double pcharn,
                              double *pus, double *pts, double *pqs, double *t_zu_o, double *q_zu_o,
                              double *Ubzu, double *pz0)
{
    const double zzi0 = 600., zBeta0 = 1.2;
    int l_zt_equal_zu = (fabs(zu - zt) < 0.01);
    double t_zu = dmax(t_zt, 180.);
    double q_zu = dmax(q_zt, 1.e-6);
    double zz0 = 0.0001;
    double zlog_10 = log(10.);
    double zlog_zt = log(zt);
    double zlog_zu = log(zu);
    double zc_a = 0.035 * log(10. / zz0) / log(zu / zz0);
    double zc_b = 0.004 * zzi0 * zBeta0 * zBeta0 * zBeta0;
    double zdt = t_zu - psst;  zdt = fsign(dmax(fabs(zdt), 1.E-09), zdt);
    double zdq = q_zu - pssq;  zdq = fsign(dmax(fabs(zdq), 1.E-12), zdq);
    double zNu_a = abo_visc_air(t_zu);
    double zUb = sqrt(U_zu * U_zu + 0.5 * 0.5);
    double zus = zc_a * zUb;
    zz0 = pcharn * zus * zus / grav + 0.11 * zNu_a / zus;
    zz0 = dmin(dmax(fabs(zz0), 1.E-8), 1.);
    double zlog_z0 = log(zz0);
    double zCd = (vkarmn / (zlog_zu - zlog_z0));
    zCd = zCd * zCd;
    double z1_o_sqrt_Cd10 = (zlog_10 - zlog_z0) / vkarmn;
    double zz0t = 10. / exp(vkarmn / (0.00115 * z1_o_sqrt_Cd10));
    zz0t = dmin(dmax(fabs(zz0t), 1.E-8), 1.);
    double zlog_z0t = log(zz0t);
    double zRib = abo_ri_bulk(zu, psst, t_zu, pssq, q_zu, zUb);
    double zcc = vkarmn2 / (zCd * (zlog_zt - zlog_z0t));
    double zcc_ri = zcc * zRib;
    double z1_o_Ribcu = -zc_b / zu;
    double zstab = 0.5 + fsign(0.5, zRib);
    double zzeta_u = (1. - zstab) * zcc_ri / (1. + zRib * z1_o_Ribcu)
                     + zstab * (zcc_ri + 27. / 9. * zRib * zRib);
    zus = dmax(zUb * vkarmn / (zlog_zu - zlog_z0 - abo_psi_m_coare(zzeta_u)), 1.E-9);
    double ztmp = vkarmn / (zlog_zu - zlog_z0t - abo_psi_h_coare(zzeta_u));
    double zts = zdt * ztmp;
    double zqs = zdq * ztmp;
    if (!l_zt_equal_zu) {
        double zzeta_t = zt * zzeta_u / zu;
        double zprf = log(zt / zu) + abo_psi_h_coare(zzeta_u) - abo_psi_h_coare(zzeta_t);
        t_zu = t_zt - zts / vkarmn * zprf;
        q_zu = q_zt - zqs / vkarmn * zprf;
        q_zu = (0.5 + fsign(0.5, q_zu)) * q_zu;
        zdt = t_zu - psst;  zdt = fsign(dmax(fabs(zdt), 1.E-09), zdt);
        zdq = q_zu - pssq;  zdq = fsign(dmax(fabs(zdq), 1.E-12), zdq);
        zts = zdt * ztmp;
        zqs = zdq * ztmp;
    }
    *pus = zus; *pts = zts; *pqs = zqs; *Ubzu = zUb; *t_zu_o = t_zu; *q_zu_o = q_zu;
    zz0 = pcharn * zus * zus / grav + 0.11 * zNu_a / zus;
    *pz0 = dmin(dmax(fabs(zz0), 1.E-8), 1.);
}

/* ---- Charnock parameters ----------------------------------------------------------- */
/* charn_coare3p6_sclr, mod_blk_coare3p6.f90:417-432 */
double abo_charn_coare3p6(double pwnd) { return dmax(dmin(0.0017 * pwnd - 0.005, 0.028), 0.); }

/* charn_coare3p0, mod_blk_coare3p0.f90:420-447 */
double abo_charn_coare3p0(double zw)
{
    double zgt10 = 0.5 + fsign(0.5, (zw - 10.));
    double zgt18 = 0.5 + fsign(0.5, (zw - 18.));
    return (1. - zgt10) * 0.011
           + zgt10 * ((1. - zgt18) * (0.011 + (0.018 - 0.011) * (zw - 10.) / (18. - 10.)) + zgt18 * (0.018));
}

/* ---- skin schemes: mod_skin_coare.f90 ---------------------------------------------- */

/* CS_COARE (c0=0.137, with Qlat) mod_skin_coare.f90:48-93 ; CS_ECMWF (c0=0.065, no Qlat) mod_skin_ecmwf.f90:68-110 */
static double cool_skin(double pQsw, double pQnsol, double pustar, double pSST, int coare, double pQlat)
{
    double c0 = coare ? 0.137 : 0.065;
    double zQabs = pQnsol;
    double zdelta = abo_delta_skin_layer(abo_alpha_sw(pSST), zQabs, pustar, coare, pQlat);
    for (int jc = 0; jc < 4; ++jc) {
        double zfr = dmax(c0 + 11. * zdelta - 6.6E-5 / zdelta * (1. - exp(-zdelta / 8.E-4)), 0.01);
        zQabs = pQnsol + zfr * pQsw;
        zdelta = abo_delta_skin_layer(abo_alpha_sw(pSST), zQabs, pustar, coare, pQlat);
    }
    return zQabs * zdelta / rk0_w;
}

static inline double fmodulo(double a, double p) /* Fortran MODULO for reals */
{
    double r = fmod(a, p);
    if (r != 0. && ((r < 0.) != (p < 0.))) r += p;
    return r;
}
static inline int imodulo(int a, int p)
{
    int r = a % p;
    if (r != 0 && ((r < 0) != (p < 0))) r += p;
    return r;
}

/* WL_COARE, mod_skin_coare.f90:97-250.  st[0..3] = dT_wl, Hz_wl, Qnt_ac, Tau_ac of this cell */
static void wl_coare(double *st, double pQsw, double pQnsol, double pTau, double pSST, double plon,
                     int isd, int iwait)
{
    const double Hwl_max = 20., Rich0 = 0.65, zfr0 = 0.5;
    double zQabs = 0.;
    double zfr = zfr0;
    int l_exit = 0, l_destroy_wl = 0;
    double zdTwl = st[0];
    double zHwl = dmax(dmin(st[1], Hwl_max), 0.1);
    double zqac = st[2];
    double ztac = st[3];

    double rlag_gw_h = -1. * fmodulo((360. - fmodulo(plon, 360.)) / 15., 24.);
    rlag_gw_h = -1. * fsign(dmin(fabs(rlag_gw_h), fabs(fmodulo(rlag_gw_h, 24.))), rlag_gw_h + 12.);
    int ilag_gw_s = (int)(rlag_gw_h * 3600.);
    int isd_sol = imodulo(isd + ilag_gw_s, 24 * 3600);
    double rhr_sol = (double)isd_sol / 3600.;

    double zalpha = abo_alpha_sw(pSST);
    double zcd1 = sqrt(2. * Rich0 * rCp0_w / (zalpha * grav * rho0_w));
    double zcd2 = sqrt(2. * zalpha * grav / (Rich0 * rho0_w)) / (pow(rCp0_w, 1.5));

    if ((rhr_sol > 4.) && (rhr_sol <= 6.5)) { l_exit = 1; l_destroy_wl = 1; }

    if (!l_exit) {
        zfr = 1. - (0.28 * 0.014 * (1. - exp(-zHwl / 0.014)) + 0.27 * 0.357 * (1. - exp(-zHwl / 0.357))
                    + 0.45 * 12.82 * (1 - exp(-zHwl / 12.82))) / zHwl;
        zQabs = zfr * pQsw + pQnsol;
        if ((fabs(zdTwl) < 1.E-6) && (zQabs <= 0.)) l_exit = 1;
    }
    if ((!l_exit) && (st[2] + zQabs * rdt <= 0.)) { l_exit = 1; l_destroy_wl = 1; }

    if (!l_exit) {
        ztac = st[3] + dmax(.002, pTau) * rdt;
        for (int jl = 0; jl < 5; ++jl) {
            zfr = 1. - (0.28 * 0.014 * (1. - exp(-zHwl / 0.014)) + 0.27 * 0.357 * (1. - exp(-zHwl / 0.357))
                        + 0.45 * 12.82 * (1 - exp(-zHwl / 12.82))) / zHwl;
            zQabs = zfr * pQsw + pQnsol;
            zqac = st[2] + zQabs * rdt;
            if (zqac <= 0.) break;
            zHwl = dmax(dmin(Hwl_max, zcd1 * ztac / sqrt(zqac)), 0.1);
        }
        if (zqac <= 0.) {
            l_destroy_wl = 1; l_exit = 1;
        } else {
            zdTwl = zcd2 * pow(zqac, 1.5) / ztac * dmax(zqac / fabs(zqac), 0.);
            double flg = 0.5 + fsign(0.5, gdept1 - zHwl);
            zdTwl = zdTwl * (flg + (1. - flg) * gdept1 / zHwl);
        }
    }
    if (l_destroy_wl) { zdTwl = 0.; zfr = 0.75; zHwl = Hwl_max; zqac = 0.; ztac = 0.; }
    (void)zfr;
    if (iwait == 0) { st[0] = zdTwl; st[1] = zHwl; st[2] = zqac; st[3] = ztac; }
}

/* PHI, mod_skin_ecmwf.f90:233-253 (Takaya 2010) */
double abo_phi_takaya(double pzeta)
{
    double zzt2 = pzeta * pzeta;
    double ztf = 0.5 + fsign(0.5, pzeta);
    return ztf * (1. + (5. * pzeta + 4. * zzt2) / (1. + 3. * pzeta + 0.25 * zzt2))
           + (1. - ztf) * 1. / sqrt(1. - 16. * (-fabs(pzeta)));
}

/* WL_ECMWF, mod_skin_ecmwf.f90:113-230 (pustk absent).  st[0]=dT_wl, st[1]=Hz_wl */
static void wl_ecmwf(double *st, double pQsw, double pQnsol, double pustar, double pSST)
{
    const double sq_radrw = sqrt(rho0_a / rho0_w);
    const double zRhoCp_w = rho0_w * rCp0_w;
    const double rNuwl0 = 0.5;
    double zHwl = st[1];
    double flg = 0.5 + fsign(0.5, gdept1 - zHwl);
    double ztcorr = flg + (1. - flg) * gdept1 / zHwl;
    double zdTwl_b = dmax(st[0] / ztcorr, 0.);
    double zalpha = abo_alpha_sw(pSST);
    double zfr = 1. - 0.28 * exp(-71.5 * zHwl) - 0.27 * exp(-2.8 * zHwl) - 0.45 * exp(-0.07 * zHwl);
    double zQabs = zfr * pQsw + pQnsol;
    double zusw = dmax(pustar, 1.E-4) * sq_radrw;
    double zusw2 = zusw * zusw;
    double zla = 0.3;
    double zfLa = dmax(pow(zla, (-2. / 3.)), 1.);
    double zwf = 0.5 + fsign(0.5, zQabs);
    double zcst1 = vkarmn * grav * zalpha;
    double zL2 = zcst1 * zQabs / (zRhoCp_w * zusw2 * zusw);
    double zcst2 = zcst1 / (5. * zHwl * zusw2);
    double zcst0 = rdt * (rNuwl0 + 1.) / zHwl;
    double zA = zcst0 * zQabs / (rNuwl0 * zRhoCp_w);
    double zcst3 = -zcst0 * vkarmn * zusw * zfLa;
    double zdTwl_n = zdTwl_b;
    for (int jc = 0; jc < 10; ++jc) {
        zdTwl_n = 0.5 * (zdTwl_n + zdTwl_b);
        double zL1 = sqrt(zdTwl_n * zcst2);
        double zeta = (1. - zwf) * zHwl * zL1 + zwf * zHwl * zL2;
        double zB = zcst3 / abo_phi_takaya(zeta);
        zdTwl_n = dmax(zdTwl_b + zA + zB * zdTwl_n, 0.);
    }
    st[0] = zdTwl_n * ztcorr;
}

/* ---- COARE 3.0 / 3.6: mod_blk_coare3p6.f90:123-413, mod_blk_coare3p0.f90:54-358 ------ */
typedef struct {
    double Cd, Ch, Ce, t_zu, q_zu, Ubzu, T_s, q_s;
    /* optional diagnostics of the TURB_* routines (CdN ChN CeN xz0 xu_star xL xUN10 pdT_cs pdT_wl pHz_wl) */
    double CdN, ChN, CeN, z0, us, L, UN10, dT_cs, dT_wl, Hz_wl;
} turb_out;

static void turb_coare(int v36, double zt, double zu, double sst, double t_zt, double q_s_in, double q_zt,
                       double zUzu, int l_skin, int nb_iter, double Qsw, double rad_lw, double slp,
                       double *wl, int isd, double plon, turb_out *o)
{
    const double zi0 = 600., zeta_abs_max = 50.;
    const double Beta0 = v36 ? 1.2 : 1.25;
    int l_zt_equal_zu = (fabs(zu - zt) < 0.01);
    double zm_ztzu = l_zt_equal_zu ? 0. : 1.;    /* 3p0 only :195 */
    double xSST = sst, T_s = sst, q_s = q_s_in;
    const int l_cs = l_skin & 1, l_wl = l_skin & 2;   /* l_skin: bit0 = l_use_cs, bit1 = l_use_wl */
    if (l_skin) {                                  /* 3p6 :271-276 ; 3p0 :211-214 */
        if (l_cs) T_s = T_s - 0.25;
        q_s = rdct_qsat_salt * abo_q_sat(dmax(T_s, 200.), slp);
    }
    double zlog_10 = log(10.), zlog_zt = log(zt), zlog_zu = log(zu);
    double zus, zts, zqs, t_zu, q_zu, Ubzu, zz0;
    first_guess_coare(zt, zu, T_s, t_zt, q_s, q_zt, zUzu,
                      v36 ? abo_charn_coare3p6(zUzu) : abo_charn_coare3p0(zUzu),
                      &zus, &zts, &zqs, &t_zu, &q_zu, &Ubzu, &zz0);
    double zlog_z0 = log(zz0);
    double znu_a = abo_visc_air(v36 ? t_zu : t_zt);  /* 3p6 :294 vs 3p0 :237 */
    double zdt = t_zu - T_s;  zdt = fsign(dmax(fabs(zdt), 1.E-09), zdt);
    double zdq = q_zu - q_s;  zdq = fsign(dmax(fabs(zdq), 1.E-12), zdq);
    double zdT_cs = 0., zzta_t = 0., zlog_z0t_last = 0., z1oL_last = 0.;

    for (int jit = 1; jit <= nb_iter; ++jit) {
        double zus2 = zus * zus;
        double z1oL = abo_one_on_l(t_zu, q_zu, zus, zts, zqs);
        z1oL = fsign(dmin(fabs(z1oL), 200.), z1oL);
        z1oL_last = z1oL;
        double zgust2 = Beta0 * Beta0 * zus2 * pow(dmax(-zi0 * z1oL / vkarmn, 0.), (2. / 3.));
        Ubzu = dmax(sqrt(zUzu * zUzu + zgust2), 0.2);
        double zzta_u = zu * z1oL;
        zzta_u = fsign(dmin(fabs(zzta_u), zeta_abs_max), zzta_u);
        if (!v36 || !l_zt_equal_zu) {              /* 3p6 :319-322 conditional; 3p0 :262-263 always */
            zzta_t = zt * z1oL;
            zzta_t = fsign(dmin(fabs(zzta_t), zeta_abs_max), zzta_t);
        }
        double zUn10 = zus / vkarmn * (zlog_10 - zlog_z0);
        zz0 = (v36 ? abo_charn_coare3p6(zUn10) : abo_charn_coare3p0(zUn10)) * zus2 / grav + 0.11 * znu_a / zus;
        zz0 = dmin(dmax(fabs(zz0), 1.E-9), 1.);
        zlog_z0 = log(zz0);
        double ztmp1, zz0t;
        if (v36) {
            ztmp1 = pow(znu_a / (zz0 * zus), 0.72);            /* :333 */
            zz0t = dmin(1.6E-4, 5.8E-5 * ztmp1);               /* :334 */
        } else {
            ztmp1 = pow(znu_a / (zz0 * zus), 0.6);             /* 3p0 :275 */
            zz0t = dmin(1.1E-4, 5.5E-5 * ztmp1);               /* 3p0 :276 */
        }
        zz0t = dmin(dmax(fabs(zz0t), 1.E-9), 1.);
        double zlog_z0t = log(zz0t);
        zlog_z0t_last = zlog_z0t;
        double ztmp0 = abo_psi_h_coare(zzta_u);
        ztmp1 = vkarmn / (zlog_zu - zlog_z0t - ztmp0);
        zts = zdt * ztmp1;
        zqs = zdq * ztmp1;
        zus = dmax(Ubzu * vkarmn / (zlog_zu - zlog_z0 - abo_psi_m_coare(zzta_u)), 1.E-9);
        if (v36) {
            if (!l_zt_equal_zu) {                               /* :346-351 */
                ztmp1 = zlog_zt - zlog_zu + ztmp0 - abo_psi_h_coare(zzta_t);
                t_zu = t_zt - zts / vkarmn * ztmp1;
                q_zu = q_zt - zqs / vkarmn * ztmp1;
            }
        } else {                                                /* 3p0 :289-291 */
            ztmp1 = zlog_zt - zlog_zu + ztmp0 - abo_psi_h_coare(zzta_t);
            t_zu = t_zt - zm_ztzu * zts / vkarmn * ztmp1;
            q_zu = q_zt - zm_ztzu * zqs / vkarmn * ztmp1;
        }
        if (l_cs) {                                 /* cool skin :353-363 */
            double zQns, zTau, zQlat;
            update_qnsol_tau(zu, T_s, q_s, t_zu, q_zu, zus, zts, zqs, zUzu, Ubzu, slp, rad_lw, &zQns, &zTau, &zQlat);
            zdT_cs = cool_skin(Qsw, zQns, zus, xSST, 1, zQlat);
            T_s = xSST + zdT_cs;
            if (l_wl) T_s = T_s + wl[0];
            q_s = rdct_qsat_salt * abo_q_sat(dmax(T_s, 200.), slp);
        }
        if (l_wl) {                                 /* warm layer :365-376 */
            double zQns, zTau;
            update_qnsol_tau(zu, T_s, q_s, t_zu, q_zu, zus, zts, zqs, zUzu, Ubzu, slp, rad_lw, &zQns, &zTau, NULL);
            wl_coare(wl, Qsw, zQns, zTau, xSST, plon, isd, nb_iter % jit);
            T_s = xSST + wl[0];
            if (l_cs) T_s = T_s + zdT_cs;
            q_s = rdct_qsat_salt * abo_q_sat(dmax(T_s, 200.), slp);
        }
        if (!v36 || l_skin || !l_zt_equal_zu) {    /* 3p6 :378-381 conditional; 3p0 :317-318 always */
            zdt = t_zu - T_s;  zdt = fsign(dmax(fabs(zdt), 1.E-09), zdt);
            zdq = q_zu - q_s;  zdq = fsign(dmax(fabs(zdq), 1.E-12), zdq);
        }
    }
    double ztmp0 = zus / Ubzu;
    o->Cd = dmax(ztmp0 * ztmp0, Cx_min);
    o->Ch = dmax(ztmp0 * zts / zdt, Cx_min);
    o->Ce = dmax(ztmp0 * zqs / zdq, Cx_min);
    o->t_zu = t_zu; o->q_zu = q_zu; o->Ubzu = Ubzu; o->T_s = T_s; o->q_s = q_s;
    /* optional outputs, mod_blk_coare3p6.f90:392-407 (3p0 :337-352) */
    ztmp0 = 1. / (zlog_zu - zlog_z0);
    o->CdN = dmax(vkarmn2 * ztmp0 * ztmp0, Cx_min);
    {
        double zt1 = vkarmn2 * ztmp0 / (zlog_zu - zlog_z0t_last);
        o->ChN = dmax(zt1, Cx_min);
        o->CeN = dmax(zt1, Cx_min);
    }
    o->z0 = zz0; o->us = zus; o->L = 1. / z1oL_last; o->UN10 = zus / vkarmn * (zlog_10 - zlog_z0);
    o->dT_cs = zdT_cs; o->dT_wl = l_wl ? wl[0] : 0.; o->Hz_wl = l_wl ? wl[1] : 0.;
}

/* ---- ECMWF: mod_blk_ecmwf.f90 ------------------------------------------------------ */
static inline double cap_zeta(double z) { return dmin(dmax(z, -50.), 5.); } /* :551-564 */

/* psi_m_ecmwf_scl :441-477 */
double abo_psi_m_ecmwf(double pzeta)
{
    double zc = 5. / 0.35;
    double zta = cap_zeta(pzeta);
    double zx2 = sqrt(fabs(1. - 16. * zta));
    double zx = sqrt(zx2);
    double ztmp = 1. + zx;
    double zpsi_unst = log(0.125 * ztmp * ztmp * (1. + zx2)) - 2. * atan(zx) + 0.5 * rpi;
    double zpsi_stab = -2. / 3. * (zta - zc) * exp(-0.35 * zta) - zta - 2. / 3. * zc;
    double zstab = 0.5 + fsign(0.5, zta);
    return zstab * zpsi_stab + (1. - zstab) * zpsi_unst;
}

/* psi_h_ecmwf_scl :498-533 */
double abo_psi_h_ecmwf(double pzeta)
{
    double zc = 5. / 0.35;
    double zta = cap_zeta(pzeta);
    double zx2 = sqrt(fabs(1. - 16. * zta));
    double zpsi_unst = 2. * log(0.5 * (1. + zx2));
    double zpsi_stab = -2. / 3. * (zta - zc) * exp(-0.35 * zta) - pow(fabs(1. + 2. / 3. * zta), 1.5) - 2. / 3. * zc + 1.;
    double zstab = 0.5 + fsign(0.5, zta);
    return zstab * zpsi_stab + (1. - zstab) * zpsi_unst;
}

/* turb_ecmwf :63-383 */
static void turb_ecmwf(double zt, double zu, double sst, double zt_zt, double q_s_in, double zq_zt,
                       double zUzu, int l_skin, int nb_iter, double Qsw, double rad_lw, double slp,
                       double *wl, turb_out *o)
{
    const double charn0_ecmwf = 0.018, zi0 = 1000., Beta0 = 1., alpha_M = 0.11, alpha_H = 0.40, alpha_Q = 0.62;
    double zm_ztzu = (fabs(zu - zt) < 0.01) ? 0. : 1.;
    double zSST = sst, zT_s = sst, zq_s = q_s_in;
    const int l_cs = l_skin & 1, l_wl = l_skin & 2;   /* bit0 = l_use_cs, bit1 = l_use_wl (:209-216) */
    if (l_skin) {
        if (l_cs) zT_s = zT_s - 0.25;
        zq_s = rdct_qsat_salt * abo_q_sat(dmax(zT_s, 200.), slp);
    }
    double zlog_10 = log(10.), zlog_zu = log(zu), zlog_ztu = log(zt / zu);
    double zus, zts, zqs, zt_zu, zq_zu, zUbzu, zz0;
    first_guess_coare(zt, zu, zT_s, zt_zt, zq_s, zq_zt, zUzu, charn0_ecmwf,
                      &zus, &zts, &zqs, &zt_zu, &zq_zu, &zUbzu, &zz0);
    double zlog_z0 = log(zz0);
    double znu_a = abo_visc_air(zt_zt);
    double zdt = zt_zu - zT_s;  zdt = fsign(dmax(fabs(zdt), 1.E-09), zdt);
    double zdq = zq_zu - zq_s;  zdq = fsign(dmax(fabs(zdq), 1.E-12), zdq);
    double z1oL = abo_one_on_l(zt_zu, zq_zu, zus, zts, zqs);
    double zzeta_u = zu * z1oL;
    double zzeta_t;
    double zz0t = dmin(dmax(fabs(1. / (0.1 * exp(vkarmn / (0.00115 / (vkarmn / (zlog_10 - zlog_z0)))))), 1.E-9), 1.);
    double zlog_z0t = log(zz0t);
    double zFm = zlog_zu - zlog_z0 - abo_psi_m_ecmwf(zzeta_u) + abo_psi_m_ecmwf(zz0 * z1oL);
    double zpsi_h_u = abo_psi_h_ecmwf(zzeta_u);
    double zFh = zlog_zu - zlog_z0t - zpsi_h_u + abo_psi_h_ecmwf(zz0t * z1oL);
    double zlog_z0q = 0., zpsi_h_z0q = 0., zdT_cs = 0.;

    for (int jit = 1; jit <= nb_iter; ++jit) {
        double zRib = abo_ri_bulk(zu, zT_s, zt_zu, zq_s, zq_zu, zUbzu);
        z1oL = zRib * zFm * zFm / zFh / zu;
        z1oL = fsign(dmin(fabs(z1oL), 200.), z1oL);
        zzeta_u = zu * z1oL;
        double zpsi_m_u = abo_psi_m_ecmwf(zzeta_u);
        zpsi_h_u = abo_psi_h_ecmwf(zzeta_u);
        zzeta_t = zt * z1oL;
        double zpsi_h_t = abo_psi_h_ecmwf(zzeta_t);
        zFm = zlog_zu - zlog_z0 - zpsi_m_u + abo_psi_m_ecmwf(zz0 * z1oL);
        zus = zUbzu * vkarmn / zFm;
        double zus2 = zus * zus;
        double ztmp0 = znu_a / zus;
        zz0 = dmin(fabs(alpha_M * ztmp0 + charn0_ecmwf * zus2 / grav), 0.001);
        zz0t = dmin(fabs(alpha_H * ztmp0), 0.001);
        double zz0q = dmin(fabs(alpha_Q * ztmp0), 0.001);
        zlog_z0 = log(zz0);
        zlog_z0t = log(zz0t);
        zlog_z0q = log(zz0q);
        double zpsi_m_z0 = abo_psi_m_ecmwf(zz0 * z1oL);
        double zpsi_h_z0t = abo_psi_h_ecmwf(zz0t * z1oL);
        zpsi_h_z0q = abo_psi_h_ecmwf(zz0q * z1oL);
        ztmp0 = Beta0 * Beta0 * zus2 * pow(dmax(-zi0 * z1oL / vkarmn, 0.), (2. / 3.));
        zUbzu = dmax(sqrt(zUzu * zUzu + ztmp0), 0.2);
        ztmp0 = zpsi_h_u - zpsi_h_z0t;
        double ztmp1 = vkarmn / (zlog_zu - zlog_z0t - ztmp0);
        zts = zdt * ztmp1;
        ztmp1 = zlog_ztu + ztmp0 - zpsi_h_t + zpsi_h_z0t;
        zt_zu = zt_zt - zm_ztzu * zts / vkarmn * ztmp1;
        ztmp0 = zpsi_h_u - zpsi_h_z0q;
        ztmp1 = vkarmn / (zlog_zu - zlog_z0q - ztmp0);
        zqs = zdq * ztmp1;
        ztmp1 = zlog_ztu + ztmp0 - zpsi_h_t + zpsi_h_z0q;
        zq_zu = dmax(zq_zt - zm_ztzu * zqs / vkarmn * ztmp1, 0.);
        zFm = zlog_zu - zlog_z0 - zpsi_m_u + zpsi_m_z0;
        zFh = zlog_zu - zlog_z0t - zpsi_h_u + zpsi_h_z0t;
        if (l_cs) {                                 /* :319-329 */
            double zQns, zTau;
            update_qnsol_tau(zu, zT_s, zq_s, zt_zu, zq_zu, zus, zts, zqs, zUzu, zUbzu, slp, rad_lw, &zQns, &zTau, NULL);
            zdT_cs = cool_skin(Qsw, zQns, zus, zSST, 0, 0.);
            zT_s = zSST + zdT_cs;
            if (l_wl) zT_s = zT_s + wl[0];
            zq_s = rdct_qsat_salt * abo_q_sat(dmax(zT_s, 200.), slp);
        }
        if (l_wl) {                                 /* :331-340 */
            double zQns, zTau;
            update_qnsol_tau(zu, zT_s, zq_s, zt_zu, zq_zu, zus, zts, zqs, zUzu, zUbzu, slp, rad_lw, &zQns, &zTau, NULL);
            wl_ecmwf(wl, Qsw, zQns, zus, zSST);
            zT_s = zSST + wl[0];
            if (l_cs) zT_s = zT_s + zdT_cs;
            zq_s = rdct_qsat_salt * abo_q_sat(dmax(zT_s, 200.), slp);
        }
        zdt = zt_zu - zT_s;  zdt = fsign(dmax(fabs(zdt), 1.E-09), zdt);
        zdq = zq_zu - zq_s;  zdq = fsign(dmax(fabs(zdq), 1.E-12), zdq);
    }
    double zFq = zlog_zu - zlog_z0q - zpsi_h_u + zpsi_h_z0q;
    o->Cd = dmax(vkarmn2 / (zFm * zFm), Cx_min);
    o->Ch = dmax(vkarmn2 / (zFm * zFh), Cx_min);
    o->Ce = dmax(vkarmn2 / (zFm * zFq), Cx_min);
    o->t_zu = zt_zu; o->q_zu = zq_zu; o->Ubzu = zUbzu; o->T_s = zT_s; o->q_s = zq_s;
    /* optional outputs, mod_blk_ecmwf.f90:362-377 */
    {
        double zt0 = 1. / (zlog_zu - zlog_z0);
        double zt1 = vkarmn2 * zt0 / (zlog_zu - zlog_z0t);
        o->CdN = dmax(vkarmn2 * zt0 * zt0, Cx_min);
        o->ChN = dmax(zt1, Cx_min);
        o->CeN = dmax(zt1, Cx_min);
    }
    o->z0 = zz0; o->us = zus; o->L = 1. / z1oL; o->UN10 = zus / vkarmn * (zlog_10 - zlog_z0);
    o->dT_cs = zdT_cs; o->dT_wl = l_wl ? wl[0] : 0.; o->Hz_wl = l_wl ? wl[1] : 0.;
}

/* ---- NCAR: mod_blk_ncar.f90 -------------------------------------------------------- */
/* cd_n10_ncar_sclr :244-271 */
double abo_cd_n10_ncar(double zw)
{
    double zw6 = zw * zw * zw;
    zw6 = zw6 * zw6;
    double zgt33 = 0.5 + fsign(0.5, (zw - 33.));
    double r = 1.e-3 * ((1. - zgt33) * (2.7 / zw + 0.142 + zw / 13.09 - 3.14807E-10 * zw6) + zgt33 * 2.34);
    return dmax(r, Cx_min);
}
/* psi_m_ncar_sclr :333-363 */
double abo_psi_m_ncar(double zta)
{
    double zx2 = sqrt(fabs(1. - 16. * zta));
    zx2 = dmax(zx2, 1.);
    double zx = sqrt(zx2);
    double zpsi_unst = 2. * log((1. + zx) * 0.5) + log((1. + zx2) * 0.5) - 2. * atan(zx) + rpi * 0.5;
    double zpsi_stab = -5. * zta;
    double zstab = 0.5 + fsign(0.5, zta);
    return zstab * zpsi_stab + (1. - zstab) * zpsi_unst;
}
/* psi_h_ncar_sclr :379-407 */
double abo_psi_h_ncar(double zta)
{
    double zx2 = sqrt(fabs(1. - 16. * zta));
    zx2 = dmax(zx2, 1.);
    double zpsi_unst = 2. * log(0.5 * (1. + zx2));
    double zpsi_stab = -5. * zta;
    double zstab = 0.5 + fsign(0.5, zta);
    return zstab * zpsi_stab + (1. - zstab) * zpsi_unst;
}

/* turb_ncar :57-240 */
static void turb_ncar(double zt, double zu, double sst, double t_zt, double ssq, double q_zt, double U_zu,
                      int nb_iter, turb_out *o)
{
    int l_zt_equal_zu = (fabs(zu - zt) < 0.01);
    double Ubzu = dmax(0.5, U_zu);
    double zlog1 = log(zt / zu);
    double zlog2 = log(zu / 10.);
    double zstab = 0.5 + fsign(0.5, virt_temp(t_zt, q_zt) - virt_temp(sst, ssq));
    double zCdN = abo_cd_n10_ncar(Ubzu);
    double zsqrt_CdN = sqrt(zCdN);
    double Cd = zCdN;
    double Ce = dmax(1.e-3 * (34.6 * zsqrt_CdN), Cx_min);                                /* ce_n10 :321 */
    double Ch = dmax(1.e-3 * zsqrt_CdN * (18. * zstab + 32.7 * (1. - zstab)), Cx_min);   /* ch_n10 :301 */
    double zsqrt_Cd = zsqrt_CdN;
    double t_zu = dmax(t_zt, 180.);
    double q_zu = dmax(q_zt, 1.e-6);
    double d_us = 0., d_1oL = 0., d_un10 = 0., d_chn = 0., d_cen = 0.;
    for (int jit = 1; jit <= nb_iter; ++jit) {
        double zdt = t_zu - sst;
        double zdq = q_zu - ssq;
        double zus = zsqrt_Cd * Ubzu;
        double zts = Ch / zsqrt_Cd * zdt;
        double zqs = Ce / zsqrt_Cd * zdq;
        double z1oL = abo_one_on_l(t_zu, q_zu, zus, zts, zqs);
        double zeta_u = zu * z1oL;
        zeta_u = fsign(dmin(fabs(zeta_u), 10.), zeta_u);
        if (!l_zt_equal_zu) {
            double zeta_t = zt * z1oL;
            zeta_t = fsign(dmin(fabs(zeta_t), 10.), zeta_t);
            double ztmp = zlog1 + abo_psi_h_ncar(zeta_u) - abo_psi_h_ncar(zeta_t);
            t_zu = t_zt - zts / vkarmn * ztmp;
            q_zu = q_zt - zqs / vkarmn * ztmp;
            q_zu = dmax(0., q_zu);
        }
        double zpsi_m = abo_psi_m_ncar(zeta_u);
        double zUn10 = dmax(0.25, un10_from_cd(zu, Ubzu, Cd, zpsi_m));
        zCdN = abo_cd_n10_ncar(zUn10);
        zsqrt_CdN = sqrt(zCdN);
        double ztmp = 1. + zsqrt_CdN / vkarmn * (zlog2 - zpsi_m);
        Cd = dmax(zCdN / (ztmp * ztmp), Cx_min);
        zsqrt_Cd = sqrt(Cd);
        ztmp = (zlog2 - abo_psi_h_ncar(zeta_u)) / vkarmn / zsqrt_CdN;
        double ztmp2 = zsqrt_Cd / zsqrt_CdN;
        zstab = 0.5 + fsign(0.5, zeta_u);
        double zChN = 1.e-3 * zsqrt_CdN * (18. * zstab + 32.7 * (1. - zstab));
        double zCeN = 1.e-3 * (34.6 * zsqrt_CdN);
        Ch = dmax(zChN * ztmp2 / (1. + zChN * ztmp), Cx_min);
        Ce = dmax(zCeN * ztmp2 / (1. + zCeN * ztmp), Cx_min);
        d_us = zus; d_1oL = z1oL; d_un10 = zUn10; d_chn = zChN; d_cen = zCeN;
    }
    o->Cd = Cd; o->Ch = Ch; o->Ce = Ce; o->t_zu = t_zu; o->q_zu = q_zu; o->Ubzu = Ubzu;
    o->T_s = sst; o->q_s = ssq;
    /* optional outputs, mod_blk_ncar.f90:229-235 (z0_from_Cd without psi, mod_phymbl.f90:1349) */
    o->CdN = zCdN; o->CeN = d_cen; o->ChN = d_chn; o->UN10 = d_un10; o->L = 1. / d_1oL; o->us = d_us;
    o->z0 = dmin(zu * exp(-vkarmn / sqrt(zCdN)), z0_sea_max);
    o->dT_cs = 0.; o->dT_wl = 0.; o->Hz_wl = 0.;
}

/* ---- ANDREAS: mod_blk_andreas.f90 -------------------------------------------------- */
/* u_star_andreas_sclr :275-293 */
double abo_u_star_andreas(double pun10)
{
    double za = pun10 - 8.271;
    double zt = za + sqrt(0.12 * za * za + 0.181);
    return 0.239 + 0.0433 * zt;
}
/* psi_m_andreas :307-360 */
double abo_psi_m_andreas(double pzeta)
{
    const double zam = 5., zbm = 5. / 6.5, z1o3 = 1. / 3.;
    const double zsr3 = sqrt(3.);
    double zta = dmin(pzeta, 15.);
    double zx2 = sqrt(fabs(1. - 16. * zta));
    zx2 = dmax(zx2, 1.);
    double zx = sqrt(zx2);
    double zpsi_unst = 2. * log(fabs((1. + zx) * 0.5)) + log(fabs((1. + zx2) * 0.5)) - 2. * atan(zx) + rpi * 0.5;
    zx = pow(fabs(1. + zta), z1o3);
    double zbbm = pow(fabs((1. - zbm) / zbm), z1o3);
    double zpsi_stab = -3. * zam / zbm * (zx - 1.) + zam * zbbm / (2. * zbm) * (
                           2. * log(fabs((zx + zbbm) / (1. + zbbm)))
                           - log(fabs((zx * zx - zx * zbbm + zbbm * zbbm) / (1. - zbbm + zbbm * zbbm)))
                           + 2. * zsr3 * (atan((2. * zx - zbbm) / (zsr3 * zbbm)) - atan((2. - zbbm) / (zsr3 * zbbm))));
    double zstab = 0.5 + fsign(0.5, zta);
    return zstab * zpsi_stab + (1. - zstab) * zpsi_unst;
}
/* psi_h_andreas :363-410 */
double abo_psi_h_andreas(double pzeta)
{
    const double zah = 5., zbh = 5., zch = 3.;
    const double zbbh = sqrt(5.);
    double zta = dmin(pzeta, 15.);
    double zx2 = sqrt(fabs(1. - 16. * zta));
    zx2 = dmax(zx2, 1.);
    double zpsi_unst = 2. * log(0.5 * (1. + zx2));
    double zz = 2. * zta + zch;
    double zpsi_stab = -0.5 * zbh * log(fabs(1. + zch * zta + zta * zta))
                       + (-zah / zbbh + 0.5 * zbh * zch / zbbh)
                             * (log(fabs((zz - zbbh) / (zz + zbbh))) - log(fabs((zch - zbbh) / (zch + zbbh))));
    double zstab = 0.5 + fsign(0.5, zta);
    return zstab * zpsi_stab + (1. - zstab) * zpsi_unst;
}

/* turb_andreas :66-272 (whole-array statements are pointwise; evaluated here per cell) */
static void turb_andreas(double zt, double zu, double psst, double pt_zt, double pssq, double pq_zt, double pU_zu,
                         int nb_iter, turb_out *o)
{
    const double rRi_max = 0.15, rCs_min = 0.35E-3;
    int l_zt_equal_zu = (fabs(zu - zt) < 0.01);
    double pUbzu = dmax(0.25, pU_zu);
    double UN10 = pUbzu;
    double pCd = 1.1E-3, pCh = 1.1E-3, pCe = 1.1E-3;
    double pt_zu = pt_zt, pq_zu = pq_zt;
    double ztmp0 = sqrt(pCd);
    double t_star = pCh / ztmp0 * (pt_zu - psst);
    double q_star = pCe / ztmp0 * (pq_zu - pssq);
    double RiB = abo_ri_bulk(zu, psst, pt_zu, pssq, pq_zu, pUbzu);
    double u_star = 0., d_z0 = 0., d_zeta = 0.;
    for (int jit = 1; jit <= nb_iter; ++jit) {
        if (RiB < rRi_max) u_star = abo_u_star_andreas(UN10);
        else u_star = sqrt(Cx_min) * pUbzu;
        double zeta_u = zu * abo_one_on_l(pt_zu, pq_zu, u_star, t_star, q_star);
        ztmp0 = u_star / pUbzu;
        pCd = dmax(ztmp0 * ztmp0, Cx_min);
        double z0 = dmin(z0_from_cd_psi(zu, pCd, abo_psi_m_andreas(zeta_u)), z0_sea_max);
        d_z0 = z0; d_zeta = zeta_u;
        ztmp0 = z0 * u_star / abo_visc_air(pt_zu);
        double ztmp1 = abo_z0tq_lkb(1, ztmp0, z0);
        double ztmp2 = abo_z0tq_lkb(2, ztmp0, z0);
        ztmp0 = abo_psi_h_andreas(zeta_u);
        t_star = (pt_zu - psst) * vkarmn / (log(zu) - log(ztmp1) - ztmp0);
        q_star = (pq_zu - pssq) * vkarmn / (log(zu) - log(ztmp2) - ztmp0);
        if ((!l_zt_equal_zu) && (jit > 1)) {
            ztmp0 = zeta_u / zu * zt;
            ztmp0 = log(zt / zu) + abo_psi_h_andreas(zeta_u) - abo_psi_h_andreas(ztmp0);
            pt_zu = pt_zt - t_star / vkarmn * ztmp0;
            pq_zu = pq_zt - q_star / vkarmn * ztmp0;
            RiB = abo_ri_bulk(zu, psst, pt_zu, pssq, pq_zu, pUbzu);
        }
        /* UN10_from_ustar, mod_phymbl.f90:1498-1510 */
        UN10 = dmax(0.1, pUbzu - u_star / vkarmn * (log(zu / 10.) - abo_psi_m_andreas(zeta_u)));
    }
    ztmp0 = u_star / pUbzu;
    pCd = dmax(ztmp0 * ztmp0, Cx_min);
    double ztmp1 = pt_zu - psst;  ztmp1 = fsign(dmax(fabs(ztmp1), 1.E-6), ztmp1);
    double ztmp2 = pq_zu - pssq;  ztmp2 = fsign(dmax(fabs(ztmp2), 1.E-9), ztmp2);
    pCh = dmax(ztmp0 * t_star / ztmp1, rCs_min);
    pCe = dmax(ztmp0 * q_star / ztmp2, rCs_min);
    o->Cd = pCd; o->Ch = pCh; o->Ce = pCe; o->t_zu = pt_zu; o->q_zu = pq_zu; o->Ubzu = pUbzu;
    o->T_s = psst; o->q_s = pssq;
    /* optional outputs, mod_blk_andreas.f90:256-267 */
    ztmp0 = 1. / log(zu / d_z0);
    o->CdN = dmax(vkarmn2 * ztmp0 * ztmp0, Cx_min);
    ztmp1 = d_z0 * u_star / abo_visc_air(pt_zu);
    o->ChN = vkarmn2 * ztmp0 / log(zu / abo_z0tq_lkb(1, ztmp1, d_z0));
    o->CeN = vkarmn2 * ztmp0 / log(zu / abo_z0tq_lkb(2, ztmp1, d_z0));
    o->z0 = d_z0; o->us = u_star; o->L = zu / d_zeta;
    o->UN10 = pUbzu - u_star / vkarmn * (log(zu / 10.) - abo_psi_m_andreas(d_zeta));
    o->dT_cs = 0.; o->dT_wl = 0.; o->Hz_wl = 0.;
}

/* ---- aerobulk_compute: mod_aerobulk_compute.f90:22-213 ------------------------------ */
int abo_compute_diag(int algo, int jt, int nt, long n, double zt, double zu, int nb_iter,
                     int use_skin, int hum_type,
                     const double *sst, const double *t_zt, const double *hum_zt,
                     const double *u_zu, const double *v_zu, const double *slp,
                     const double *rad_sw, const double *rad_lw,
                     double *ql, double *qh, double *tau_x, double *tau_y, double *evap, double *t_s,
                     double *wl_state, int isecday_utc, const double *lon, double *diag)
{
    (void)nt;
    if (algo < ABO_COARE3P0 || algo > ABO_ANDREAS) return 2;
    int l_skin = (use_skin && (algo == ABO_COARE3P0 || algo == ABO_COARE3P6 || algo == ABO_ECMWF)) ? 3 : 0;  /* cs + wl */
    if (l_skin && (!rad_sw || !rad_lw || !wl_state)) return 2;
    int rc = 0;
    for (long k = 0; k < n; ++k) {
        /* :99-108 humidity conversion */
        double zQzt;
        if (hum_type == ABO_HUM_SH) zQzt = hum_zt[k];
        else if (hum_type == ABO_HUM_DP) zQzt = abo_q_air_dp(hum_zt[k], dmax(slp[k], 50000.));
        else zQzt = abo_q_air_rh(hum_zt[k], t_zt[k], dmax(slp[k], 50000.));
        double zWzu = sqrt(u_zu[k] * u_zu[k] + v_zu[k] * v_zu[k]);            /* :111 */
        double zSSQ = rdct_qsat_salt * abo_q_sat(sst[k], slp[k]);             /* :114 */
        double zThtzt = abo_theta_from_z_p0_t_q(zt, slp[k], t_zt[k], zQzt);   /* :118 */
        double wl[4] = {0., 0., 0., 0.};
        double Qsw = 0., rlw = 0.;
        if (l_skin) {
            Qsw = (1. - roce_alb0) * rad_sw[k];                               /* :135,146,161 */
            rlw = rad_lw[k];
            if (jt == 1) {  /* COARE3Px_INIT mod_blk_coare3p6.f90:80-88 ; ECMWF_INIT mod_blk_ecmwf.f90:399-405 */
                wl[0] = 0.;
                wl[1] = (algo == ABO_ECMWF) ? 3. : 20.;
                wl[2] = 0.;
                wl[3] = 0.;
            } else {
                for (int s = 0; s < 4; ++s) wl[s] = wl_state[(long)s * n + k];
            }
        }
        turb_out o;
        switch (algo) {
        case ABO_COARE3P0:
            turb_coare(0, zt, zu, sst[k], zThtzt, zSSQ, zQzt, zWzu, l_skin, nb_iter, Qsw, rlw, slp[k], wl,
                       isecday_utc, lon ? lon[k] : 0., &o);
            break;
        case ABO_COARE3P6:
            turb_coare(1, zt, zu, sst[k], zThtzt, zSSQ, zQzt, zWzu, l_skin, nb_iter, Qsw, rlw, slp[k], wl,
                       isecday_utc, lon ? lon[k] : 0., &o);
            break;
        case ABO_NCAR:
            turb_ncar(zt, zu, sst[k], zThtzt, zSSQ, zQzt, zWzu, nb_iter, &o);
            break;
        case ABO_ECMWF:
            turb_ecmwf(zt, zu, sst[k], zThtzt, zSSQ, zQzt, zWzu, l_skin, nb_iter, Qsw, rlw, slp[k], wl, &o);
            break;
        default:
            turb_andreas(zt, zu, sst[k], zThtzt, zSSQ, zQzt, zWzu, nb_iter, &o);
            break;
        }
        if (l_skin)
            for (int s = 0; s < 4; ++s) wl_state[(long)s * n + k] = wl[s];
        /* :184-185 BULK_FORMULA_VCTR */
        double zTaum, QH, QL, zEvap;
        bulk_formula(zu, o.T_s, o.q_s, o.t_zu, o.q_zu, o.Cd, o.Ch, o.Ce, zWzu, o.Ubzu, slp[k],
                     &zTaum, &QH, &QL, &zEvap);
        if (zTaum > 10.) rc = 1;                                              /* mod_phymbl.f90:1250 */
        qh[k] = QH;
        ql[k] = QL;
        /* :189-194 */
        tau_x[k] = 0.;
        tau_y[k] = 0.;
        if (zWzu > 1.E-3) {
            tau_x[k] = zTaum / zWzu * u_zu[k];
            tau_y[k] = zTaum / zWzu * v_zu[k];
        }
        if (t_s) t_s[k] = o.T_s;                                              /* :206 */
        if (evap) evap[k] = zEvap;                                            /* :208 */
        if (diag) {
            const double d[16] = {o.Cd, o.Ch, o.Ce, o.t_zu, o.q_zu, o.Ubzu, o.CdN, o.ChN, o.CeN, o.z0, o.us, o.L, o.UN10,
                                  o.dT_cs, o.dT_wl, o.Hz_wl};
            for (int s = 0; s < 16; ++s) diag[(long)s * n + k] = d[s];
        }
    }
    return rc;
}

int abo_compute(int algo, int jt, int nt, long n, double zt, double zu, int nb_iter,
                int use_skin, int hum_type,
                const double *sst, const double *t_zt, const double *hum_zt,
                const double *u_zu, const double *v_zu, const double *slp,
                const double *rad_sw, const double *rad_lw,
                double *ql, double *qh, double *tau_x, double *tau_y, double *evap, double *t_s,
                double *wl_state, int isecday_utc, const double *lon)
{
    return abo_compute_diag(algo, jt, nt, n, zt, zu, nb_iter, use_skin, hum_type, sst, t_zt, hum_zt, u_zu, v_zu, slp, rad_sw,
                            rad_lw, ql, qh, tau_x, tau_y, evap, t_s, wl_state, isecday_utc, lon, NULL);
}

/* ---- AEROBULK_INIT host checks: mod_aerobulk.f90:104-153 ---------------------------- */
/* ---- the TURB_* routines called directly (e.g. mod_blk_coare3p6.f90:123-131), as the reference's station drivers do
 * (tests/test_aerobulk_buoy_series_oce.f90:452-487): no pre-processing, no bulk formula. */
int abo_turb(int algo, int kt, long n, double zt, double zu, int nb_iter, int use_cs, int use_wl,
             double *T_s, const double *theta_zt, double *q_s, const double *q_zt, const double *U_zu,
             const double *Qsw, const double *rad_lw, const double *slp,
             double *wl_state, int isecday_utc, const double *lon, double *diag)
{
    if (algo < ABO_COARE3P0 || algo > ABO_ANDREAS || !diag) return 2;
    int l_skin = 0;
    if (algo == ABO_COARE3P0 || algo == ABO_COARE3P6 || algo == ABO_ECMWF) l_skin = (use_cs ? 1 : 0) | (use_wl ? 2 : 0);
    else if (use_cs || use_wl) return 2;
    if (l_skin && (!Qsw || !rad_lw || !slp)) return 2;
    if ((l_skin & 2) && !wl_state) return 2;
    for (long k = 0; k < n; ++k) {
        double wl[4] = {0., 0., 0., 0.};
        if (l_skin & 2) {
            if (kt == 1) wl[1] = (algo == ABO_ECMWF) ? 3. : 20.;
            else for (int s = 0; s < 4; ++s) wl[s] = wl_state[(long)s * n + k];
        }
        const double qsw = l_skin ? Qsw[k] : 0., rlw = l_skin ? rad_lw[k] : 0., p = l_skin ? slp[k] : 0.;
        turb_out o;
        switch (algo) {
        case ABO_COARE3P0:
        case ABO_COARE3P6:
            turb_coare(algo == ABO_COARE3P6, zt, zu, T_s[k], theta_zt[k], q_s[k], q_zt[k], U_zu[k], l_skin, nb_iter, qsw, rlw, p,
                       wl, isecday_utc, lon ? lon[k] : 0., &o);
            break;
        case ABO_NCAR: turb_ncar(zt, zu, T_s[k], theta_zt[k], q_s[k], q_zt[k], U_zu[k], nb_iter, &o); break;
        case ABO_ECMWF:
            turb_ecmwf(zt, zu, T_s[k], theta_zt[k], q_s[k], q_zt[k], U_zu[k], l_skin, nb_iter, qsw, rlw, p, wl, &o);
            break;
        default: turb_andreas(zt, zu, T_s[k], theta_zt[k], q_s[k], q_zt[k], U_zu[k], nb_iter, &o); break;
        }
        if (l_skin & 2)
            for (int s = 0; s < 4; ++s) wl_state[(long)s * n + k] = wl[s];
        if (l_skin) { T_s[k] = o.T_s; q_s[k] = o.q_s; }
        const double d[16] = {o.Cd, o.Ch, o.Ce, o.t_zu, o.q_zu, o.Ubzu, o.CdN, o.ChN, o.CeN, o.z0, o.us, o.L, o.UN10,
                              o.dT_cs, o.dT_wl, o.Hz_wl};
        for (int s = 0; s < 16; ++s) diag[(long)s * n + k] = d[s];
    }
    return 0;
}


/* ======================================================================================
 * Sea-ice bulk algorithms: src/ice/mod_blk_ice_nemo.f90, mod_blk_ice_an05.f90,
 * mod_blk_ice_lu12.f90, mod_blk_ice_lg15.f90, mod_cdn_form_ice.f90 (per-cell literal restatement)
 * ====================================================================================== */
static const double wspd_thrshld_ice = 0.2;   /* mod_const.f90:120 */
static const double rCd_ice = 1.4e-3;         /* mod_const.f90:118 */
static const double rz0_i_s_0 = 0.69e-3;      /* mod_blk_ice_lg15.f90:56, mod_blk_ice_lu12.f90:57 */
static const double rz0_i_f_0 = 4.54e-4;      /* mod_blk_ice_lg15.f90:57 */
static const double ralpha_0 = 0.2;           /* mod_blk_ice_lg15.f90:54 */
static const double rce10_i_0 = 3.46e-3;      /* mod_cdn_form_ice.f90:34 */
static const double rbeta_0_miz = 1.4;        /* mod_cdn_form_ice.f90:25 */

/* rough_leng_m, mod_blk_ice_an05.f90:232-255 */
static double an05_rough_leng_m(double pus, double pnua)
{
    double zus = dmax(pus, 1.E-9);
    double zz = (zus - 0.18) / 0.1;
    return 0.135 * pnua / zus + 0.035 * zus * zus / grav * (5. * exp(-zz * zz) + 1.);
}
/* rough_leng_tq, mod_blk_ice_an05.f90:257-312 */
static void an05_rough_leng_tq(double pz0, double pus, double pnua, double *z0t, double *z0q)
{
    double zz0 = pz0;
    double zus = dmax(pus, 1.E-9);
    double zre = dmax(zus * zz0 / pnua, 0.);
    double zsmoot = 0.5 + fsign(0.5, (0.135 - zre));
    double ztrans = 0.5 + fsign(0.5, (2.49999 - zre)) - zsmoot;
    double zrough = 0.5 + fsign(0.5, (zre - 2.5));
    double zlog = log(zre);
    double zlog2 = zlog * zlog;
    double zb0 = zsmoot * 1.25 + ztrans * 0.149 + zrough * 0.317;
    double zb1 = -ztrans * 0.550 - zrough * 0.565;
    double zb2 = -zrough * 0.183;
    *z0t = zz0 * exp(zb0 + zb1 * zlog + zb2 * zlog2);
    zb0 = zsmoot * 1.61 + ztrans * 0.351 + zrough * 0.396;
    zb1 = -ztrans * 0.628 - zrough * 0.512;
    zb2 = -zrough * 0.180;
    *z0q = zz0 * exp(zb0 + zb1 * zlog + zb2 * zlog2);
}
/* psi_m_ice / psi_h_ice, mod_blk_ice_an05.f90:316-405 */
double abo_psi_m_ice(double zta)
{
    double zx = pow(fabs(1. - 16. * zta), .25);
    double zpsi_u = log((1. + zx * zx) / 2.) + 2. * log((1. + zx) / 2.) - 2. * atan(zx) + 0.5 * rpi;
    double zpsi_s = -(0.7 * zta + 0.75 * (zta - 14.3) * exp(-0.35 * zta) + 10.7);
    double zstab = 0.5 + fsign(0.5, zta);
    return (1. - zstab) * zpsi_u + zstab * zpsi_s;
}
double abo_psi_h_ice(double zta)
{
    double zx = pow(fabs(1. - 16. * zta), .25);
    double zpsi_u = 2. * log((1. + zx * zx) / 2.);
    double zpsi_s = -(0.7 * zta + 0.75 * (zta - 14.3) * exp(-0.35 * zta) + 10.7);
    double zstab = 0.5 + fsign(0.5, zta);
    return (1. - zstab) * zpsi_u + zstab * zpsi_s;
}
/* f_m_louis_sclr / f_h_louis_sclr, mod_phymbl.f90:1419-1479 (rc_louis = 5 :150) */
static double f_louis(double pzu, double pRib, double pCxn, double pz0, double ra)
{
    const double rc2_louis = 25.;
    double zstab = 0.5 + fsign(0.5, pRib);
    double ztu = pRib / (1. + 3. * rc2_louis * pCxn * sqrt(fabs(-pRib * (pzu / pz0 + 1.))));
    double zts = pRib / sqrt(fabs(1. + pRib));
    return (1. - zstab) * (1. - ra * ztu) + zstab * 1. / (1. + ra * zts);
}
static double f_m_louis(double pzu, double pRib, double pCdn, double pz0) { return f_louis(pzu, pRib, pCdn, pz0, 10.); }
static double f_h_louis(double pzu, double pRib, double pChn, double pz0) { return f_louis(pzu, pRib, pChn, pz0, 15.); }
/* Cd_from_z0 (no psi), mod_phymbl.f90:1396-1414 ; z0_from_Cd (no psi) :1349 */
static double cd_from_z0(double pzu, double pz0) { double c = 1. / log(pzu / pz0); return vkarmn2 * c * c; }
static double z0_from_cd(double pzu, double pCd) { return pzu * exp(-vkarmn / sqrt(pCd)); }

typedef struct { double Cd, Ch, Ce, t_zu, q_zu, Ub, CdN, ChN, CeN, z0, us, L, UN10; } ice_out;

/* shared by NEMO and LU12: constant-in-stability coefficients, optional outputs of mod_blk_ice_nemo.f90:138-145 */
static void ice_const_cx(double zu, double Ts_i, double t_zt, double qs_i, double q_zt, double U_zu, double cd, ice_out *o)
{
    double Ub = dmax(U_zu, wspd_thrshld_ice);
    double t_zu = dmax(t_zt, 100.), q_zu = dmax(q_zt, 0.1e-6);
    double dt_zu = t_zu - Ts_i;  dt_zu = fsign(dmax(fabs(dt_zu), 1.E-6), dt_zu);
    double dq_zu = q_zu - qs_i;  dq_zu = fsign(dmax(fabs(dq_zu), 1.E-9), dq_zu);
    o->Cd = cd; o->Ch = cd; o->Ce = cd; o->t_zu = t_zu; o->q_zu = q_zu; o->Ub = Ub;
    o->CdN = cd; o->ChN = cd; o->CeN = cd;
    o->z0 = z0_from_cd(zu, cd);
    o->us = sqrt(cd) * Ub;
    o->L = 1. / abo_one_on_l(t_zu, q_zu, sqrt(cd) * Ub, cd / sqrt(cd) * dt_zu, cd / sqrt(cd) * dq_zu);
    o->UN10 = sqrt(cd) * Ub / vkarmn * log(10. / z0_from_cd(zu, cd));
}

/* turb_ice_an05, mod_blk_ice_an05.f90:41-228 */
static void turb_ice_an05(double zt, double zu, double Ts_i, double t_zt, double qs_i, double q_zt, double U_zu, int nb_iter,
                          ice_out *o)
{
    int l_zt_equal_zu = (fabs(zu - zt) < 0.01);
    double Ubzu = dmax(U_zu, wspd_thrshld_ice);
    double t_zu = dmax(t_zt, 100.), q_zu = dmax(q_zt, 0.1e-6);
    double dt_zu = t_zu - Ts_i;  dt_zu = fsign(dmax(fabs(dt_zu), 1.E-6), dt_zu);
    double dq_zu = q_zu - qs_i;  dq_zu = fsign(dmax(fabs(dq_zu), 1.E-9), dq_zu);
    double znu_a = abo_visc_air(t_zu);
    double z0 = 8.0E-4;
    double u_star = 0.035 * Ubzu * log(10. / z0) / log(zu / z0);
    z0 = an05_rough_leng_m(u_star, znu_a);
    for (int jit = 1; jit <= 2; ++jit) {
        u_star = dmax(Ubzu * vkarmn / (log(zu) - log(z0)), 1.E-9);
        z0 = an05_rough_leng_m(u_star, znu_a);
    }
    double z0t, z0q;
    an05_rough_leng_tq(z0, u_star, znu_a, &z0t, &z0q);
    double t_star = dt_zu * vkarmn / (log(zu / z0t));
    double q_star = dq_zu * vkarmn / (log(zu / z0q));
    for (int jit = 1; jit <= nb_iter; ++jit) {
        double ztmp0 = abo_one_on_l(t_zu, q_zu, u_star, t_star, q_star);
        ztmp0 = fsign(dmin(fabs(ztmp0), 200.), ztmp0);
        double zeta_u = zu * ztmp0;
        zeta_u = fsign(dmin(fabs(zeta_u), 50.0), zeta_u);
        double zeta_t = 0.;
        if (!l_zt_equal_zu) {
            zeta_t = zt * ztmp0;
            zeta_t = fsign(dmin(fabs(zeta_t), 50.0), zeta_t);
        }
        z0 = an05_rough_leng_m(u_star, znu_a);
        an05_rough_leng_tq(z0, u_star, znu_a, &z0t, &z0q);
        ztmp0 = abo_psi_h_ice(zeta_u);
        t_star = dt_zu * vkarmn / (log(zu) - log(z0t) - ztmp0);
        q_star = dq_zu * vkarmn / (log(zu) - log(z0q) - ztmp0);
        u_star = dmax(Ubzu * vkarmn / (log(zu) - log(z0) - abo_psi_m_ice(zeta_u)), 1.E-9);
        if (!l_zt_equal_zu) {
            double ztmp1 = log(zt / zu) + ztmp0 - abo_psi_h_ice(zeta_t);
            t_zu = t_zt - t_star / vkarmn * ztmp1;
            q_zu = q_zt - q_star / vkarmn * ztmp1;
            dt_zu = t_zu - Ts_i;  dt_zu = fsign(dmax(fabs(dt_zu), 1.E-6), dt_zu);
            dq_zu = q_zu - qs_i;  dq_zu = fsign(dmax(fabs(dq_zu), 1.E-9), dq_zu);
        }
    }
    double ztmp0 = u_star / Ubzu;
    o->Cd = ztmp0 * ztmp0;
    o->Ch = ztmp0 * t_star / dt_zu;
    o->Ce = ztmp0 * q_star / dq_zu;
    o->t_zu = t_zu; o->q_zu = q_zu; o->Ub = Ubzu;
    ztmp0 = 1. / log(zu / z0);
    o->CdN = vkarmn2 * ztmp0 * ztmp0;
    o->ChN = vkarmn2 * ztmp0 / log(zu / z0t);
    o->CeN = vkarmn2 * ztmp0 / log(zu / z0q);
    o->z0 = z0; o->us = u_star;
    o->L = 1. / abo_one_on_l(t_zu, q_zu, u_star, t_star, q_star);
    o->UN10 = u_star / vkarmn * log(10. / z0);
}

/* CdN_f_LG15_light (mod_cdn_form_ice.f90:272-307) assigns its WHOLE result array inside its cell loop, so every cell ends
 * up with the form drag of the LAST cell of the array: frice_last. */
double abo_cdn_f_lg15_light(double zu, double frice_last)
{
    double ztmp = 1. / rz0_i_f_0;
    double zrlog = log(10. * ztmp) / log(zu * ztmp);
    return rce10_i_0 * zrlog * zrlog * frice_last * pow(1. - frice_last, rbeta_0_miz);
}

/* turb_ice_lg15, mod_blk_ice_lg15.f90:68-307 (= the over-ice part of turb_ice_lg15_io, mod_blk_ice_lg15_io.f90:69-404) */
static void turb_ice_lg15(double zt, double zu, double Ts_i, double t_zt, double qs_i, double q_zt, double U_zu,
                          double frice_last, int nb_iter, ice_out *o)
{
    int l_zt_equal_zu = (fabs(zu - zt) < 0.01);
    double Ubzu = dmax(U_zu, wspd_thrshld_ice);
    double t_zu = dmax(t_zt, 100.), q_zu = dmax(q_zt, 0.1e-6);
    double dt_zu = t_zu - Ts_i;  dt_zu = fsign(dmax(fabs(dt_zu), 1.E-6), dt_zu);
    double dq_zu = q_zu - qs_i;  dq_zu = fsign(dmax(fabs(dq_zu), 1.E-9), dq_zu);
    double zz0_s = rz0_i_s_0;
    double zCdN_s = cd_from_z0(zu, zz0_s);
    double zChN_s = vkarmn2 / (log(zu / zz0_s) * log(zu / (ralpha_0 * zz0_s)));
    double zz0_f = rz0_i_f_0;
    double zCdN_f = abo_cdn_f_lg15_light(zu, frice_last);
    double zChN_f = zCdN_f / (1. + log(1. / ralpha_0) / vkarmn * sqrt(zCdN_f));
    double Cd_i = zCdN_s + zCdN_f;
    double Ch_i = zChN_s + zChN_f;
    double RiB = abo_ri_bulk(zt, Ts_i, t_zt, qs_i, q_zt, Ubzu);
    for (int jit = 1; jit <= nb_iter; ++jit) {
        double xtmp1, xtmp2;
        if (!l_zt_equal_zu) {
            xtmp1 = zCdN_s + zCdN_f;
            xtmp2 = zz0_s + zz0_f;
            xtmp1 = log(zt / zu) + f_h_louis(zu, RiB, xtmp1, xtmp2) - f_h_louis(zt, RiB, xtmp1, xtmp2);
            xtmp2 = dmax(Ubzu + (sqrt(Cd_i) * Ubzu) * xtmp1, wspd_thrshld_ice);
            xtmp2 = dmin(xtmp2, Ubzu);
        } else {
            xtmp2 = Ubzu;
        }
        RiB = abo_ri_bulk(zt, Ts_i, t_zt, qs_i, q_zt, xtmp2);
        Cd_i = zCdN_s * f_m_louis(zu, RiB, zCdN_s, zz0_s);
        Ch_i = zChN_s * f_h_louis(zu, RiB, zCdN_s, zz0_s);
        Cd_i = Cd_i + zCdN_f * f_m_louis(zu, RiB, zCdN_f, zz0_f);
        Ch_i = Ch_i + zChN_f * f_h_louis(zu, RiB, zCdN_f, zz0_f);
        if (!l_zt_equal_zu) {
            xtmp1 = zCdN_s + zCdN_f;
            xtmp2 = zz0_s + zz0_f;
            xtmp1 = log(zt / zu) + f_h_louis(zu, RiB, xtmp1, xtmp2) - f_h_louis(zt, RiB, xtmp1, xtmp2);
            xtmp2 = 1. / sqrt(Cd_i);
            t_zu = t_zt - (Ch_i * dt_zu * xtmp2) / vkarmn * xtmp1;
            q_zu = q_zt - (Ch_i * dq_zu * xtmp2) / vkarmn * xtmp1;
            q_zu = dmax(0., q_zu);
            dt_zu = t_zu - Ts_i;
            dq_zu = q_zu - qs_i;
            dt_zu = fsign(dmax(fabs(dt_zu), 1.E-6), dt_zu);
            dq_zu = fsign(dmax(fabs(dq_zu), 1.E-9), dq_zu);
        }
    }
    double Ce_i = Ch_i;
    o->Cd = Cd_i; o->Ch = Ch_i; o->Ce = Ce_i; o->t_zu = t_zu; o->q_zu = q_zu; o->Ub = Ubzu;
    o->CdN = zCdN_s + zCdN_f; o->ChN = zChN_s + zChN_f; o->CeN = zChN_s + zChN_f;
    o->z0 = z0_from_cd(zu, zCdN_s + zCdN_f);
    o->us = sqrt(Cd_i) * Ubzu;
    {
        double x = sqrt(Cd_i);
        o->L = 1. / abo_one_on_l(t_zu, q_zu, x * Ubzu, Ch_i * dt_zu / x, Ce_i * dq_zu / x);
    }
    o->UN10 = sqrt(Cd_i) * Ubzu / vkarmn * log(10. / z0_from_cd(zu, zCdN_s + zCdN_f));
}

/* turb_ice_easy, mod_blk_ice_easy.f90:44-209: Andreas-2005-type stability correction of PRESCRIBED neutral coefficients */
static void turb_ice_easy(double zt, double zu, double Ts_i, double t_zt, double qs_i, double q_zt, double U_zu, double CdN,
                          double ChN, double CeN, int nb_iter, ice_out *o)
{
    int l_zt_equal_zu = (fabs(zu - zt) < 0.01);
    double zsqrtCDN = sqrt(CdN), zlog1 = log(zt / zu), zlog2 = log(zu / 10.);
    double Ubzu = dmax(U_zu, wspd_thrshld_ice);
    double t_zu = dmax(t_zt, 100.), q_zu = dmax(q_zt, 0.1e-6);
    double Cd_i = CdN, Ch_i = ChN, Ce_i = CeN;
    double u_star = 0., t_star = 0., q_star = 0., zeta_u = 0.;
    for (int jit = 1; jit <= nb_iter; ++jit) {
        double dt_zu = t_zu - Ts_i, dq_zu = q_zu - qs_i;          /* no floor here (:147-148) */
        double ztmp0 = sqrt(Cd_i);
        u_star = ztmp0 * Ubzu;
        ztmp0 = 1. / dmax(ztmp0, 1.E-15);
        t_star = Ch_i * dt_zu * ztmp0;
        q_star = Ce_i * dq_zu * ztmp0;
        ztmp0 = abo_one_on_l(t_zu, q_zu, u_star, t_star, q_star);
        ztmp0 = fsign(dmin(fabs(ztmp0), 200.), ztmp0);
        zeta_u = zu * ztmp0;
        zeta_u = fsign(dmin(fabs(zeta_u), 50.0), zeta_u);
        double zeta_t = 0.;
        if (!l_zt_equal_zu) {
            zeta_t = zt * ztmp0;
            zeta_t = fsign(dmin(fabs(zeta_t), 50.0), zeta_t);
        }
        ztmp0 = 1. + zsqrtCDN / vkarmn * (zlog2 - abo_psi_m_ice(zeta_u));
        Cd_i = dmin(dmax(CdN / (ztmp0 * ztmp0), Cx_min), 1.9E-3);
        ztmp0 = (zlog2 - abo_psi_h_ice(zeta_u)) / vkarmn / zsqrtCDN;
        double ztmp1 = sqrt(Cd_i) / zsqrtCDN;
        Ch_i = dmin(dmax(ChN * ztmp1 / (1. + ChN * ztmp0), Cx_min), 1.9E-3);
        Ce_i = dmin(dmax(CeN * ztmp1 / (1. + CeN * ztmp0), Cx_min), 1.9E-3);
        if (!l_zt_equal_zu) {
            ztmp0 = abo_psi_h_ice(zeta_u) - abo_psi_h_ice(zeta_t) + zlog1;
            t_zu = t_zt - t_star / vkarmn * ztmp0;
            q_zu = dmax(0., q_zt - q_star / vkarmn * ztmp0);
        }
    }
    o->Cd = Cd_i; o->Ch = Ch_i; o->Ce = Ce_i; o->t_zu = t_zu; o->q_zu = q_zu; o->Ub = Ubzu;
    o->CdN = CdN; o->ChN = ChN; o->CeN = CeN;
    double psm = abo_psi_m_ice(zeta_u);
    o->z0 = z0_from_cd_psi(zu, Cd_i, psm);
    o->us = u_star;
    o->L = 1. / abo_one_on_l(t_zu, q_zu, u_star, t_star, q_star);
    o->UN10 = un10_from_cd(zu, Ubzu, Cd_i, psm);
}

/* TURB_ICE_<algo> over n cells.  ice_algo: 1 nemo, 2 an05, 3 lu12, 4 lg15.  frice (ice concentration) is read by lu12
 * (per cell, mod_cdn_form_ice.f90:147-191 with rMu_0 = rNu_0 = 1, rBeta_0 = 1.4) and lg15 (last cell only, see above).
 * diag: 13 planes Cd Ch Ce t_zu q_zu Ub CdN ChN CeN z0 u_star L UN10. */
int abo_turb_ice(int ice_algo, long n, double zt, double zu, int nb_iter, const double *Ts_i, const double *t_zt,
                 const double *qs_i, const double *q_zt, const double *U_zu, const double *frice, double *diag)
{
    if (ice_algo < 1 || ice_algo > 4 || !diag) return 2;
    if ((ice_algo == 3 || ice_algo == 4) && !frice) return 2;
    for (long k = 0; k < n; ++k) {
        ice_out o;
        switch (ice_algo) {
        case 1: ice_const_cx(zu, Ts_i[k], t_zt[k], qs_i[k], q_zt[k], U_zu[k], rCd_ice, &o); break;
        case 2: turb_ice_an05(zt, zu, Ts_i[k], t_zt[k], qs_i[k], q_zt[k], U_zu[k], nb_iter, &o); break;
        case 3: {
            const double rCe_0 = 2.23E-3, zcoef = 1. + 1. / (10. * rbeta_0_miz);  /* rNu_0 + 1/(10 rBeta_0): Fortran is case-blind, rBeta_0 IS rbeta_0 = 1.4 (:25) */
            double cdf = rCe_0 * pow(frice[k], 1. - 1.) * pow(1. - frice[k], zcoef);   /* CdN10_f_LU13 */
            ice_const_cx(zu, Ts_i[k], t_zt[k], qs_i[k], q_zt[k], U_zu[k], cd_from_z0(zu, rz0_i_s_0) + cdf, &o);
            break;
        }
        default: turb_ice_lg15(zt, zu, Ts_i[k], t_zt[k], qs_i[k], q_zt[k], U_zu[k], frice[n - 1], nb_iter, &o); break;
        }
        const double d[13] = {o.Cd, o.Ch, o.Ce, o.t_zu, o.q_zu, o.Ub, o.CdN, o.ChN, o.CeN, o.z0, o.us, o.L, o.UN10};
        for (int s = 0; s < 13; ++s) diag[(long)s * n + k] = d[s];
    }
    return 0;
}

/* ---- turb_neutral_10m, mod_blk_neutral_10m.f90:33-209: neutral 10 m coefficients from the neutral 10 m wind.
 * algo: coare3p0, coare3p6, ecmwf (nb_iter fixed-point passes on the Charnock + smooth-flow roughness), ncar (closed form);
 * the reference STOPs for andreas ("YET TO BE CODED").  Outputs CdN10 ChN10 CeN10 z0. */
int abo_turb_neutral_10m(int algo, long n, int nb_iter, const double *U_N10, double *CdN10, double *ChN10, double *CeN10,
                         double *pz0)
{
    const double zu = 10., rnu0_air = 1.5E-5;   /* mod_blk_neutral_10m.f90:28, mod_const.f90:89 */
    if (algo == ABO_ANDREAS || algo < ABO_COARE3P0 || algo > ABO_ANDREAS) return 2;
    for (long k = 0; k < n; ++k) {
        if (algo == ABO_NCAR) {
            double Ub = dmax(U_N10[k], 0.5);
            double cd = abo_cd_n10_ncar(Ub);
            double sq = sqrt(cd);
            CdN10[k] = cd;
            ChN10[k] = dmax(1.e-3 * sq * (18. * (Ub * 0.) + 32.7 * (1. - Ub * 0.)), Cx_min);   /* pstab = Ub*0. :174 */
            CeN10[k] = dmax(1.e-3 * (34.6 * sq), Cx_min);
            pz0[k] = dmin(dmax(10. * exp(-vkarmn / sqrt(cd)), 0.0001), 0.1);
            continue;
        }
        double Ub = dmax(U_N10[k], 0.1);
        double cd = 8.575E-5 * Ub + 0.657E-3;
        double u_star = 0., z0 = 0., ztmp0 = 0.;
        for (int jit = 1; jit <= nb_iter; ++jit) {
            u_star = Ub * sqrt(cd);
            double charn = algo == ABO_COARE3P6 ? abo_charn_coare3p6(Ub) : (algo == ABO_COARE3P0 ? abo_charn_coare3p0(Ub) : 0.018);
            z0 = charn * u_star * u_star / grav + 0.11 * rnu0_air / u_star;
            ztmp0 = log(zu / z0);
            cd = vkarmn2 / (ztmp0 * ztmp0);
        }
        double z0t, z0q;
        if (algo == ABO_COARE3P0) {
            double rr = z0 * u_star / rnu0_air;
            z0t = dmin(1.1E-4, 5.5E-5 * pow(rr, -0.6));
            z0q = z0t;
        } else if (algo == ABO_COARE3P6) {
            double rr = z0 * u_star / rnu0_air;
            z0t = dmin(1.6e-4, 5.8E-5 * pow(rr, -0.72));
            z0q = z0t;
        } else {
            double t = rnu0_air / u_star;
            z0t = 0.40 * t;
            z0q = 0.62 * t;
        }
        CdN10[k] = cd;
        ChN10[k] = vkarmn2 / (ztmp0 * log(zu / z0t));
        CeN10[k] = vkarmn2 / (ztmp0 * log(zu / z0q));
        pz0[k] = z0;
    }
    return 0;
}

int abo_turb_ice_easy(long n, double zt, double zu, int nb_iter, const double *Ts_i, const double *t_zt, const double *qs_i,
                      const double *q_zt, const double *U_zu, double CdN, double ChN, double CeN, double *diag)
{
    if (!diag || nb_iter < 1) return 2;
    for (long k = 0; k < n; ++k) {
        ice_out o;
        turb_ice_easy(zt, zu, Ts_i[k], t_zt[k], qs_i[k], q_zt[k], U_zu[k], CdN, ChN, CeN, nb_iter, &o);
        const double d[13] = {o.Cd, o.Ch, o.Ce, o.t_zu, o.q_zu, o.Ub, o.CdN, o.ChN, o.CeN, o.z0, o.us, o.L, o.UN10};
        for (int s = 0; s < 13; ++s) diag[(long)s * n + k] = d[s];
    }
    return 0;
}

static int check_unit(long n, const double *x, const double *x2, int wind_module, const unsigned char *mask,
                      double zmin, double zmax)
{
    /* check_unit_consistency, mod_phymbl.f90:1851-1954: masked mean, masked min/max */
    double s = 0., cnt = 0., mn = HUGE_VAL, mx = -HUGE_VAL;
    for (long k = 0; k < n; ++k) {
        double v = wind_module ? sqrt(x[k] * x[k] + x2[k] * x2[k]) : x[k];
        s += v * (double)mask[k];
        cnt += (double)mask[k];
        if (mask[k]) { if (v < mn) mn = v; if (v > mx) mx = v; }
    }
    double zmean = s / cnt;
    return (mx > zmax) || (mn < zmin) || (zmean < zmin) || (zmean > zmax);
}

int abo_init_checks(long n, const double *sst, const double *t_air, const double *hum,
                    const double *u, const double *v, const double *slp,
                    const double *rad_sw, const double *rad_lw,
                    int *hum_type_out, long *n_masked_out, int *bad_field)
{
    /* buffers: a byte mask, mod_aerobulk.f90:105-115 (ranges mod_const.f90:138-146) */
    unsigned char *mask = (unsigned char *)__builtin_malloc((size_t)n);
    long np = 0;
    for (long k = 0; k < n; ++k) {
        unsigned char m = 1;
        if ((sst[k] < 270.) || (sst[k] > 320.)) m = 0;
        if ((t_air[k] < 180.) || (t_air[k] > 330.)) m = 0;
        if ((slp[k] < 80000.) || (slp[k] > 110000.)) m = 0;
        if (sqrt(u[k] * u[k] + v[k] * v[k]) > 50.) m = 0;
        if (rad_sw && rad_lw) {
            if ((rad_sw[k] < 0.) || (rad_sw[k] > 1500.)) m = 0;
            if ((rad_lw[k] < 0.) || (rad_lw[k] > 750.)) m = 0;
        }
        mask[k] = m;
        np += m;
    }
    if (n_masked_out) *n_masked_out = n - np;
    if (np <= 0) { __builtin_free(mask); return -1; }

    /* type_of_humidity, mod_phymbl.f90:1957-2007 */
    double s = 0., cnt = 0., mn = HUGE_VAL, mx = -HUGE_VAL;
    for (long k = 0; k < n; ++k) {
        s += hum[k] * (double)mask[k];
        cnt += (double)mask[k];
        if (mask[k]) { if (hum[k] < mn) mn = hum[k]; if (hum[k] > mx) mx = hum[k]; }
    }
    double zmean = s / cnt;
    int ht;
    double hmin, hmax;
    if ((zmean >= 0.) && (zmean < 0.08) && (mn >= 0.) && (mx < 0.08)) { ht = ABO_HUM_SH; hmin = 0.; hmax = 0.08; }
    else if ((zmean >= 150.) && (zmean < 330.) && (mn >= 150.) && (mx < 330.)) { ht = ABO_HUM_DP; hmin = 150.; hmax = 330.; }
    else if ((zmean >= 0.) && (zmean <= 100.) && (mn >= 0.) && (mx <= 100.)) { ht = ABO_HUM_RH; hmin = 0.; hmax = 100.; }
    else { __builtin_free(mask); return -2; }
    if (hum_type_out) *hum_type_out = ht;

    /* mod_aerobulk.f90:143-153 */
    int bad = -1;
    if (check_unit(n, sst, NULL, 0, mask, 270., 320.)) bad = 0;
    else if (check_unit(n, t_air, NULL, 0, mask, 180., 330.)) bad = 1;
    else if (check_unit(n, slp, NULL, 0, mask, 80000., 110000.)) bad = 2;
    else if (check_unit(n, u, NULL, 0, mask, -50., 50.)) bad = 3;
    else if (check_unit(n, v, NULL, 0, mask, -50., 50.)) bad = 4;
    else if (check_unit(n, u, v, 1, mask, 0., 50.)) bad = 5;
    else if (check_unit(n, hum, NULL, 0, mask, hmin, hmax)) bad = 6;
    else if (rad_sw && rad_lw) {
        if (check_unit(n, rad_sw, NULL, 0, mask, 0., 1500.)) bad = 7;
        else if (check_unit(n, rad_lw, NULL, 0, mask, 0., 750.)) bad = 8;
    }
    __builtin_free(mask);
    if (bad >= 0) { if (bad_field) *bad_field = bad; return -3; }
    return 0;
}

/* ---- synthetic inputs, SURVEY.md §8d ------------------------------------------------ */
void abo_synth_fields(int ni, int nj, int j0, int nj_local,
                      double *sst, double *t_zt, double *q_zt, double *u, double *v,
                      double *slp, double *rad_sw, double *rad_lw)
{
    static const double A[7] = {0.6180339887498949, 0.5698402909980532, 0.8191725133961645, 0.4142135623730951,
                                0.2360679774997897, 0.3166247903553998, 0.1231056256176606};
    static const double B[7] = {0.7548776662466927, 0.3247179572447460, 0.6710436067037893, 0.7320508075688772,
                                0.6457513110645906, 0.6055512754639891, 0.3588989435406740};
    static const double C[7] = {0., 0.1, 0.2, 0.3, 0.4, 0.5, 0.6};
    (void)nj;
    for (int jl = 0; jl < nj_local; ++jl) {
        int j = j0 + jl + 1;
        for (int ii = 0; ii < ni; ++ii) {
            int i = ii + 1;
            long k = (long)jl * ni + ii;
            double r[7];
            for (int m = 0; m < 7; ++m) {
                double x = (double)i * A[m] + (double)j * B[m] + C[m];
                r[m] = x - floor(x);
            }
            sst[k] = 274.15 + 29. * r[0];
            t_zt[k] = sst[k] - 6. + 9. * r[1];
            slp[k] = 98000. + 5000. * r[2];
            q_zt[k] = (0.55 + 0.4 * r[3]) * abo_q_sat(t_zt[k], slp[k]);
            u[k] = -14. + 28. * r[4];
            v[k] = -14. + 28. * r[5];
            if (rad_sw) rad_sw[k] = 900. * r[6];
            if (rad_lw) rad_lw[k] = 250. + 200. * r[0];
        }
    }
}
