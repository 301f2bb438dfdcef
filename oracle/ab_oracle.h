/* oracle/ab_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C (fp64) restatement of the AeroBulk hot path `aerobulk_compute()` and
 * of the host-side AEROBULK_INIT checks.  It exists so that tests/, the
 * smoke() check and bench.py's `cpu_baseline` leg have a CPU checker that
 * travels to the GPU box.  NOTHING in the product (aerobulk_amd/, include/)
 * may include, link or call this file.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_vs_ref.py,
 * tests/test_golden.py) against (i) the unmodified reference Fortran compiled
 * into oracle/_ref/libaerobulk_ref.so, (ii) golden vectors generated from that
 * library (tests/golden/, generator tools/gen_golden.py) and (iii) the
 * reference's own captured example output doc/ex_ab.dat.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src/).
 */
#ifndef AB_ORACLE_H
#define AB_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* algorithm ids: same numbering as aerobulk.hpp:13-21 */
enum { ABO_COARE3P0 = 1, ABO_COARE3P6 = 2, ABO_NCAR = 3, ABO_ECMWF = 4, ABO_ANDREAS = 5 };
/* humidity types: mod_const.f90:27 */
enum { ABO_HUM_SH = 0, ABO_HUM_DP = 1, ABO_HUM_RH = 2 };

/* One time record of aerobulk_compute (mod_aerobulk_compute.f90:22-213).
 *   n        : number of cells (flat)
 *   jt, nt   : current record / number of records (WL state is initialised at jt==1)
 *   wl_state : 4*n doubles [dT_wl | Hz_wl | Qnt_ac | Tau_ac] persistent between records
 *              (mod_skin_coare.f90:31-36, mod_skin_ecmwf.f90:52-55); may be NULL when
 *              use_skin==0.  ECMWF uses only the first two planes.
 *   rad_sw, rad_lw, T_s may be NULL when use_skin==0 (T_s then is not written unless non-NULL).
 *   isecday_utc / lon : warm-layer solar time inputs; aerobulk_compute hard-wires 12 and 0
 *              (mod_aerobulk_compute.f90:126,136,146); lon==NULL means 0 everywhere.
 * Returns 0, or 1 if a wind stress > 10 N/m^2 was met (mod_phymbl.f90:1250-1253; the
 * reference STOPs there), or 2 on bad arguments.
 */
int abo_compute(int algo, int jt, int nt, long n, double zt, double zu, int nb_iter,
                int use_skin, int hum_type,
                const double *sst, const double *t_zt, const double *hum_zt,
                const double *u_zu, const double *v_zu, const double *slp,
                const double *rad_sw, const double *rad_lw,
                double *ql, double *qh, double *tau_x, double *tau_y, double *evap, double *t_s,
                double *wl_state, int isecday_utc, const double *lon);

/* Same, plus the diagnostics the TURB_* routines return (mandatory and OPTIONAL outputs, e.g. mod_blk_coare3p6.f90:207-230):
 * diag = 16 planes of n doubles: Cd Ch Ce t_zu q_zu Ubzu | CdN ChN CeN z0 u_star L UN10 | dT_cs dT_wl Hz_wl (NULL: skip). */
int abo_compute_diag(int algo, int jt, int nt, long n, double zt, double zu, int nb_iter,
                     int use_skin, int hum_type,
                     const double *sst, const double *t_zt, const double *hum_zt,
                     const double *u_zu, const double *v_zu, const double *slp,
                     const double *rad_sw, const double *rad_lw,
                     double *ql, double *qh, double *tau_x, double *tau_y, double *evap, double *t_s,
                     double *wl_state, int isecday_utc, const double *lon, double *diag);

/* TURB_<algo> called directly (mod_blk_coare3p6.f90:123-131, mod_blk_coare3p0.f90:54, mod_blk_ecmwf.f90:63,
 * mod_blk_ncar.f90:57, mod_blk_andreas.f90:66): T_s, q_s are INOUT (overwritten with the skin values when use_cs or use_wl),
 * theta_zt is the POTENTIAL temperature, Qsw the NET solar flux; kt==1 initialises the warm-layer state.
 * diag: the 16 planes of abo_compute_diag (mandatory outputs first). */
int abo_turb(int algo, int kt, long n, double zt, double zu, int nb_iter, int use_cs, int use_wl,
             double *T_s, const double *theta_zt, double *q_s, const double *q_zt, const double *U_zu,
             const double *Qsw, const double *rad_lw, const double *slp,
             double *wl_state, int isecday_utc, const double *lon, double *diag);

/* Sea-ice bulk algorithms (src/ice/): TURB_ICE_NEMO (1), TURB_ICE_AN05 (2), TURB_ICE_LU12 (3), TURB_ICE_LG15 (4), e.g.
 * mod_blk_ice_an05.f90:41-43.  Inputs as the TURB_* routines (Ts_i ice surface temperature, qs_i its saturation humidity,
 * t_zt POTENTIAL temperature); frice = ice concentration (lu12, lg15; may be NULL otherwise).
 * diag: 13 planes of n doubles: Cd Ch Ce t_zu q_zu Ub CdN ChN CeN z0 u_star L UN10. */
int abo_turb_ice(int ice_algo, long n, double zt, double zu, int nb_iter, const double *Ts_i, const double *t_zt,
                 const double *qs_i, const double *q_zt, const double *U_zu, const double *frice, double *diag);
/* TURB_ICE_EASY (mod_blk_ice_easy.f90:44-47): the neutral coefficients CdN, ChN, CeN are prescribed scalars. */
/* CdN_f_LG15_light (mod_cdn_form_ice.f90:272-307) for the concentration of the LAST cell: the value every cell receives
 * (TURB_ICE_LG15 / TURB_ICE_LG15_IO; the optional CdN_frm output of the latter, mod_blk_ice_lg15_io.f90:346) */
double abo_cdn_f_lg15_light(double zu, double frice_last);
int abo_turb_ice_easy(long n, double zt, double zu, int nb_iter, const double *Ts_i, const double *t_zt, const double *qs_i,
                      const double *q_zt, const double *U_zu, double CdN, double ChN, double CeN, double *diag);
double abo_psi_m_ice(double zeta);
double abo_psi_h_ice(double zeta);

/* turb_neutral_10m (mod_blk_neutral_10m.f90:33): neutral 10 m coefficients and roughness length from the neutral 10 m wind;
 * algo = coare3p0 | coare3p6 | ecmwf | ncar (the reference STOPs for andreas). */
int abo_turb_neutral_10m(int algo, long n, int nb_iter, const double *U_N10, double *CdN10, double *ChN10, double *CeN10,
                         double *pz0);

/* AEROBULK_INIT host checks (mod_aerobulk.f90:24-160): mask, humidity type, unit ranges.
 * Returns 0 ok; negative error codes:
 *  -1 whole domain masked, -2 humidity type unidentified, -3 unit-consistency failure
 *  (field index in *bad_field: 0 sst,1 t_air,2 slp,3 u10,4 v10,5 wnd,6 hum,7 rad_sw,8 rad_lw)
 * hum_type_out receives ABO_HUM_*, n_masked_out the number of masked cells.
 * NB the reference passes prsw=rad_lw (mod_aerobulk.f90:248): rad_sw is never range-checked;
 * callers reproduce that by passing rad_lw twice.
 */
int abo_init_checks(long n, const double *sst, const double *t_air, const double *hum,
                    const double *u, const double *v, const double *slp,
                    const double *rad_sw, const double *rad_lw,
                    int *hum_type_out, long *n_masked_out, int *bad_field);

/* scalar helpers exported for unit pinning against the reference module functions */
double abo_e_sat(double t);
double abo_q_sat(double t, double p);
double abo_theta_from_z_p0_t_q(double z, double slp, double ta, double qa);
double abo_rho_air(double ta, double qa, double p);
double abo_visc_air(double ta);
double abo_one_on_l(double tha, double qa, double us, double ts, double qs);
double abo_ri_bulk(double z, double sst, double tha, double ssq, double qa, double ub);
double abo_alpha_sw(double sst);
double abo_delta_skin_layer(double alpha, double qd, double ustar, int with_qlat, double qlat);
double abo_psi_m_coare(double zeta);
double abo_psi_h_coare(double zeta);
double abo_psi_m_ecmwf(double zeta);
double abo_psi_h_ecmwf(double zeta);
double abo_psi_m_ncar(double zeta);
double abo_psi_h_ncar(double zeta);
double abo_psi_m_andreas(double zeta);
double abo_psi_h_andreas(double zeta);
double abo_cd_n10_ncar(double w);
double abo_charn_coare3p0(double w);
double abo_charn_coare3p6(double w);
double abo_u_star_andreas(double un10);
double abo_z0tq_lkb(int iflag, double rer, double z0);
double abo_q_air_rh(double rh, double ta, double p);
double abo_q_air_dp(double dp, double p);
double abo_phi_takaya(double zeta);

/* synthetic quasi-random field generator of SURVEY.md §8d (1-based i,j; Ni fastest) */
void abo_synth_fields(int ni, int nj, int j0, int nj_local,
                      double *sst, double *t_zt, double *q_zt, double *u, double *v,
                      double *slp, double *rad_sw, double *rad_lw);

#ifdef __cplusplus
}
#endif
#endif
