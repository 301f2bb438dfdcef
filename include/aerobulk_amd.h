/* include/aerobulk_amd.h — C ABI of the MI355X-native bulk air-sea flux engine.
 *
 * Drop-in boundary for the AeroBulk hot path `aerobulk_compute()` and its callers
 * `AEROBULK_INIT / AEROBULK_MODEL / AEROBULK_BYE`.  Plain pointers and sizes only; no
 * torch / C++ types.  Implemented by aerobulk_amd/csrc (libaerobulk_amd.so, HIP, gfx950).
 *
 * Citations are file:line in the reference tree (brodeau/aerobulk, `src/` unless noted).
 *
 * Data layout: every field is a flat, contiguous array of `ni*nj` cells (Fortran column-major
 * (Ni,Nj), i fastest — exactly what the reference's C shim sees as an (m,1) array,
 * mod_aerobulk_cxx.f90:40-44).  Element type is `double` for an AB_F64 session and `float`
 * for an AB_F32 / AB_F32_STORAGE / AB_F32_MIXED session.  `mem` says where the caller's arrays live: AB_MEM_HOST (the
 * reference's calling convention; the library stages H2D/D2H) or AB_MEM_DEVICE (arrays are
 * already resident in HBM; nothing is copied and the call is asynchronous on `stream`).
 */
#ifndef AEROBULK_AMD_H
#define AEROBULK_AMD_H

#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums --------------------------------------------------------------------------- */
/* numbering of include/aerobulk.hpp:13-21 */
enum ab_algo { AB_ALGO_OTHER = 0, AB_ALGO_COARE3P0 = 1, AB_ALGO_COARE3P6 = 2, AB_ALGO_NCAR = 3,
               AB_ALGO_ECMWF = 4, AB_ALGO_ANDREAS = 5 };
/* ctype_humidity 'sh' | 'dp' | 'rh', mod_const.f90:27 */
enum ab_hum { AB_HUM_SH = 0, AB_HUM_DP = 1, AB_HUM_RH = 2 };
enum ab_mem { AB_MEM_HOST = 0, AB_MEM_DEVICE = 1 };
/* AB_F32: fp32 arrays AND fp32 arithmetic (fastest; errors up to 1e-3 where theta - T_s or q - q_s is small).
 * AB_F32_STORAGE: fp32 arrays (half the HBM footprint and traffic) with the fp64 arithmetic of AB_F64: results = the fp64 path
 * on the rounded inputs, rounded once.  AB_F32_MIXED: fp32 arrays; SST, theta, T_s, q, q_s, every difference of them and q_sat in
 * fp64, everything else (profile functions, roughness lengths, cool skin, warm layer) in fp32 with the hardware transcendentals:
 * within 1e-4 of the fp64 reference (the single-precision hazard mod_blk_ecmwf.f90:556-561 warns about is exactly those
 * differences) at about twice the speed of AB_F32_STORAGE.  BASELINE config 5 runs in this mode. */
enum ab_precision { AB_F64 = 0, AB_F32 = 1, AB_F32_STORAGE = 2, AB_F32_MIXED = 3 };

/* Error codes.  The reference has none: it prints and STOPs (mod_const.f90:238-278).  Each code
 * below names the reference condition it replaces; the Fortran host turns a non-zero code
 * back into the reference's message + STOP. */
enum ab_status {
    AB_OK = 0,
    AB_ERR_ALGO = 1,        /* unknown algorithm string, mod_aerobulk_compute.f90:173-175 */
    AB_ERR_SKIN_ALGO = 2,   /* skin scheme asked for ncar/andreas, mod_aerobulk.f90:69-70 */
    AB_ERR_SKIN_NORAD = 3,  /* skin scheme without rad_sw/rad_lw, mod_aerobulk.f90:72 */
    AB_ERR_JT = 4,          /* jt < 1, mod_aerobulk.f90:244 */
    AB_ERR_ALL_MASKED = 5,  /* whole domain masked, mod_aerobulk.f90:122 */
    AB_ERR_HUM_TYPE = 6,    /* humidity type not identified, mod_phymbl.f90:1996-2003 */
    AB_ERR_UNITS = 7,       /* check_unit_consistency failed, mod_phymbl.f90:1946-1950 */
    AB_ERR_TAU = 8,         /* wind stress > 10 N/m^2, mod_phymbl.f90:1250-1253 */
    AB_ERR_HIP = 9,         /* HIP runtime failure / no gfx950 device */
    AB_ERR_ARG = 10,        /* NULL pointer, size mismatch (mod_aerobulk.f90:87-95), bad enum */
    AB_ERR_STATE = 11,      /* call protocol violated (e.g. jt>1 before jt==1, mod_aerobulk.f90:246-267) */
    AB_ERR_NOCONV = 12      /* e_air's whole-array fixed point (mod_phymbl.f90:1706-1736) not converged after 200 sweeps: the reference loops for ever */
};

typedef struct ab_session ab_session; /* opaque; owns WL state + staging buffers on one GPU */

/* Summary of the AEROBULK_INIT checks (mod_aerobulk.f90:104-153), for the caller's banner. */
typedef struct ab_init_report {
    long n_cells;
    long n_masked;      /* cells failing the sanity ranges mod_const.f90:138-146 */
    int hum_type;       /* enum ab_hum detected by type_of_humidity, mod_phymbl.f90:1957-2007 */
    int bad_field;      /* on AB_ERR_UNITS: 0 sst,1 t_air,2 slp,3 u10,4 v10,5 wnd,6 hum,7 rad_sw,8 rad_lw */
    double bad_min, bad_max, bad_mean;
} ab_init_report;

/* ---- helpers -------------------------------------------------------------------------- */
/* 'coare3p0'|'coare3p6'|'ncar'|'ecmwf'|'andreas' -> enum ab_algo, 0 if unknown.
 * `len` < 0 means NUL-terminated.  Replaces SELECT CASE(TRIM(calgo)) mod_aerobulk_compute.f90:129-176
 * and algorithm_to_string, aerobulk.cpp:22-49 (inverse direction). */
int ab_algo_from_string(const char *calgo, int len);
const char *ab_algo_name(int algo);
const char *ab_strerror(int status);
/* Human-readable detail of the last failure on this thread (mirrors the text ctl_stop prints). */
const char *ab_last_error(void);
/* Number of usable gfx950 devices (0 if none / HIP unavailable). */
int ab_device_count(void);

/* ---- session: one aerobulk_model() time loop jt = 1..nt on one GPU, or sharded over several --- */
/* Replaces the module-global state of the reference (nitend, l_use_skin_schemes, ctype_humidity,
 * nb_iter: mod_const.f90:22-33; warm-layer arrays mod_skin_coare.f90:31-36, mod_skin_ecmwf.f90:52-55)
 * with per-handle state.  `device` is the HIP device ordinal (-1: current device), or AB_DEVICE_ALL: the grid is cut into
 * contiguous blocks of rows (j-blocks), one per visible GPU — see ab_session_create_sharded. */
#define AB_DEVICE_ALL (-2)
int ab_session_create(ab_session **out, int algo, long ni, long nj, int nt, int use_skin,
                      int precision, int device);
int ab_session_destroy(ab_session *s);
/* Row-block sharding inside the library, for the callers of AEROBULK_MODEL (mod_aerobulk.f90:250-262: whole (Ni,Nj) host
 * arrays): shard r owns the rows [j0_r, j0_r + nj_r) on devices[r], with its own streams, pinned-free staging and warm-layer
 * state.  Every entry point below takes such a session like any other: host arrays are cut by rows and the shards run
 * concurrently (each device moves its rows over its own PCIe link); AEROBULK_INIT's statistics — the one exchange of the
 * path — are reduced per shard and combined on the host; no other data crosses between devices (pointwise path, no halo).
 * Results are bit-identical to one unsharded session.  A device may appear several times (several shards on it).
 * AB_MEM_DEVICE arrays are accepted only if every shard lives on the device that holds them: a GPU-resident model owns one
 * session per GPU instead.  Process-global AEROBULK_MODEL / aerobulk_cxx_*: environment AEROBULK_AMD_DEVICES = all | N | list. */
int ab_session_create_sharded(ab_session **out, int algo, long ni, long nj, int nt, int use_skin,
                              int precision, const int *devices, int nshards);
/* The same with the caller's row counts: shard r owns nj_per_shard[r] >= 1 rows (their sum must be nj), in order.  For layouts in
 * which the shards are not peers: the rows of the GPU that receives the gathered fluxes of the others (ab_session_gather's root) never
 * cross a link, so it is given MORE rows than its peers — until its kernel takes as long as a peer's kernel + that peer's transfer
 * (bench.py: balanced_peer_rows). */
int ab_session_create_sharded_rows(ab_session **out, int algo, long ni, long nj, int nt, int use_skin,
                                   int precision, const int *devices, int nshards, const long *nj_per_shard);
int ab_session_shard_count(const ab_session *s);    /* 1 for an ordinary session */
int ab_session_shard_info(const ab_session *s, int shard, long *j0, long *nj_local, int *device);

/* AEROBULK_INIT (mod_aerobulk.f90:24-160) minus its banner: builds the sanity mask, detects the
 * humidity type, runs check_unit_consistency on every input.  rad_sw/rad_lw may be NULL.
 * NB the reference passes prsw=rad_lw (mod_aerobulk.f90:248) so rad_sw is never range-checked:
 * callers that want identical behaviour pass rad_lw for both (the Fortran/C++ hosts do).
 * Global reductions run on the GPU (one pass over the inputs).  AB_MEM_DEVICE: the reduction is enqueued on `stream`
 * (hipStream_t, NULL = default stream) behind whatever still produces the fields there, and the call returns when it is done. */
int ab_session_init(ab_session *s, const void *sst, const void *t_zt, const void *hum_zt,
                    const void *u_zu, const void *v_zu, const void *slp,
                    const void *rad_sw, const void *rad_lw, int mem, void *stream, ab_init_report *report);
/* The same in two steps, for a grid sharded over several GPUs/processes (AEROBULK_INIT's statistics are the one global
 * exchange of the path): every rank reduces its own cells with ab_session_init_stats(), the ranks combine
 * the AB_INIT_NSTATS doubles — stats[0..10] by SUM (count of unmasked cells, number of cells, 9 masked sums),
 * stats[11..19] by MIN, stats[20..28] by MAX; field order sst,t_air,slp,u10,v10,wnd,hum,rad_sw,rad_lw — and every rank
 * takes the decisions on the combined numbers with ab_session_init_apply().  ab_session_init == stats + apply. */
#define AB_INIT_NSTATS 29
int ab_session_init_stats(ab_session *s, const void *sst, const void *t_zt, const void *hum_zt,
                          const void *u_zu, const void *v_zu, const void *slp,
                          const void *rad_sw, const void *rad_lw, int mem, void *stream, double stats[AB_INIT_NSTATS]);
int ab_session_init_apply(ab_session *s, const double stats[AB_INIT_NSTATS], int have_rad, ab_init_report *report);
/* Skip the detection and force the humidity type (device-resident callers that already know). */
int ab_session_set_humidity(ab_session *s, int hum_type);

/* aerobulk_compute (mod_aerobulk_compute.f90:22-213) for time record jt.
 *   inputs : sst,t_zt,hum_zt,u_zu,v_zu,slp (+rad_sw,rad_lw when the session uses skin schemes)
 *   outputs: ql,qh,tau_x,tau_y (required); evap, t_s optional (NULL = not wanted)
 * WL state initialises at jt==1 and persists until jt==nt (mod_blk_coare3p6.f90:250,411).
 * AB_MEM_HOST: synchronous; returns AB_ERR_TAU if any cell exceeded 10 N/m^2.
 * AB_MEM_DEVICE: enqueued on `stream` (hipStream_t, NULL = default stream); call
 * ab_session_check() to synchronise and fetch the AB_ERR_TAU flag.  The calls of ONE session must be ordered among themselves (one
 * stream, or the caller's own dependencies between streams): they share the warm-layer state, the error flag and the tile counters
 * of the persistent kernel.  Independent streams take independent sessions.
 * ab_session_check also VERIFIES the persistent kernel's tile counters (zero after a launch that handed out every tile): AB_ERR_STATE if
 * a launch did not finish its queue (aborted, or two computes of one session overlapped) — the counters are re-armed, the fluxes of that
 * record must be recomputed; AB_ERR_HIP if the kernel found its argument block laid out otherwise than the host assumed. */
int ab_session_compute(ab_session *s, int jt, double zt, double zu, int niter,
                       const void *sst, const void *t_zt, const void *hum_zt,
                       const void *u_zu, const void *v_zu, const void *slp,
                       const void *rad_sw, const void *rad_lw,
                       void *ql, void *qh, void *tau_x, void *tau_y, void *evap, void *t_s,
                       int mem, void *stream);
int ab_session_check(ab_session *s);

/* ---- device-resident fields of a sharded session, and the gather of the fluxes ------------------------------------------------
 * The north_star's multi-GPU layout: the global grid is cut by row blocks across the GPUs of a node, every GPU computes its rows
 * from fields that already live in its HBM, and the output tau / Q_L / Q_H / E arrays are gathered over xGMI — "a trivial RCCL
 * gather", no halo (the path is pointwise).  Call site served: AEROBULK_MODEL's coupling call, mod_aerobulk.f90:250-262, in a
 * model whose fields are GPU-resident and sharded like the session (ab_session_shard_info gives each shard's rows and device).
 *
 * ab_session_compute_shards: ab_session_compute(AB_MEM_DEVICE) with ONE SET OF POINTERS PER SHARD: shard r's arrays hold its
 * rows only (ni * nj_r elements) on its device.  `streams` (hipStream_t per shard, or NULL / NULL entries = the default stream of
 * the shard's device): the kernels are enqueued there, asynchronously; ab_session_check() synchronises and collects AB_ERR_TAU.
 * An ordinary session is a one-shard session here.
 *
 * ab_session_gather: copies the shards' flux arrays into whole-grid arrays `dst` (ni * nj elements each) on the device of shard
 * `root_shard`; shard r's rows land at element offset ni * j0_r.  Members of `dst` that are NULL are not gathered (the north_star
 * names tau, Q_L, Q_H, E: five arrays; T_s usually stays where it was computed).  Shards on other devices travel by RCCL inside the
 * process: one communicator over the DISTINCT devices of the session (ncclCommInitAll, created at the first gather, kept), one
 * ncclGroup of per-peer ncclSend / ncclRecv per call — RCCL has no native gather — straight from the shard's array into its place
 * in `dst` (no packing, no staging copy); shards that share the root's device are device-to-device copies.  Each transfer is
 * ordered behind the work already enqueued on its shard's stream (the compute that produced the fluxes), and the receives are
 * enqueued on the root shard's stream: asynchronous unless `synchronize` != 0.  librccl.so is loaded on first use (dlopen);
 * a session whose shards all live on one device never touches it.
 * STATUS: the multi-device leg (ncclCommInitAll over distinct devices, ncclSend on the shard's stream / ncclRecv on the root's in one
 * group) is EXPERIMENTAL: this pool has one GPU per box, so it has only ever run as k shards on one device through a one-device
 * communicator (tests/test_gpu_sharded.py, `python bench.py --gpus N --devices 0,0,...`); no run on two devices has happened
 * (INTEGRATION.md).  With streams == NULL the transfers use each device's NULL stream.  On an RCCL error the call drains the root's
 * stream before it returns AB_ERR_HIP: dst then holds the rows of the shards on the root's device only. */
typedef struct ab_shard_arrays {
    const void *sst, *t_zt, *hum_zt, *u_zu, *v_zu, *slp, *rad_sw, *rad_lw;   /* inputs: the shard's rows, on the shard's device */
    void *ql, *qh, *tau_x, *tau_y, *evap, *t_s;                               /* outputs (evap, t_s may be NULL) */
} ab_shard_arrays;
typedef struct ab_flux_arrays {
    void *ql, *qh, *tau_x, *tau_y, *evap, *t_s;
} ab_flux_arrays;
int ab_session_compute_shards(ab_session *s, int jt, double zt, double zu, int niter, const ab_shard_arrays *shards,
                              void *const *streams);
int ab_session_gather(ab_session *s, int root_shard, const ab_shard_arrays *shards, const ab_flux_arrays *dst,
                      void *const *streams, int synchronize);

/* Lane regrouping of the flux kernel (default on).  A thread block owns a tile of 512-1280 consecutive cells, parks their
 * pre-processed inputs in LDS and sorts them so that each 64-lane wave works on cells that take the same branches (stable /
 * unstable stratification, warm layer gaining heat / idle): on spatially incoherent input this removes most of the SIMT
 * divergence.  Results are bit-identical with it on or off; off = natural order inside the tile.
 * Environment override for A/B measurements: AEROBULK_AMD_REGROUP=0. */
int ab_session_set_regroup(ab_session *s, int on);

/* Warm-layer solar-time inputs for the next ab_session_compute calls.  aerobulk_compute
 * hard-wires isecday_utc=12 and longitude 0 (mod_aerobulk_compute.f90:126,136,146), which is
 * the default; TURB_COARE3Px callers with real time/longitude (tests/test_aerobulk_buoy_series_oce.f90
 * :345-377) set them here.  `lon` (degrees East, same layout/mem/precision as the fields) may be NULL; the session keeps its
 * own copy (a device `lon` is copied on `stream`, behind its producer; the call returns when the copy is done). */
int ab_session_set_solar_time(ab_session *s, int isecday_utc, const void *lon, int mem, void *stream);

/* Optional per-cell diagnostics: what the TURB_* routines return besides the fluxes — their mandatory outputs Cd, Ch, Ce,
 * t_zu, q_zu, Ubzu and their OPTIONAL ones CdN, ChN, CeN, xz0, xu_star, xL, xUN10, pdT_cs, pdT_wl, pHz_wl
 * (mod_blk_coare3p6.f90:207-230,392-407; mod_blk_coare3p0.f90:337-352; mod_blk_ecmwf.f90:362-377; mod_blk_ncar.f90:229-235;
 * mod_blk_andreas.f90:256-267), as consumed by the reference's toy driver (src/tests/aerobulk_toy.F90:324-393).
 * Any member may be NULL (not wanted).  The arrays (same layout / precision / `mem` as the fields) are written by every
 * following ab_session_compute() until ab_session_set_diagnostics(s, NULL, 0) switches them off again; with diagnostics on,
 * a slightly heavier kernel instantiation runs (the default one does not carry them). */
typedef struct ab_diag {
    void *Cd, *Ch, *Ce, *t_zu, *q_zu, *Ubzu;                 /* transfer coefficients, theta/q adjusted to zu, bulk wind */
    void *CdN, *ChN, *CeN, *z0, *u_star, *L, *UN10;          /* neutral coefficients, roughness, u*, Obukhov length, UN10 */
    void *dT_cs, *dT_wl, *Hz_wl;                             /* skin: cool-skin / warm-layer increments, warm-layer depth */
} ab_diag;
int ab_session_set_diagnostics(ab_session *s, const ab_diag *d, int mem);

/* One call of a TURB_<algo> routine itself, for callers that own the pre-processing and the bulk formula (NEMO's sbcblk,
 * the reference's station drivers src/tests/test_aerobulk_buoy_series_oce.f90:452-487):
 *   TURB_COARE3P6( kt, zt, zu, T_s, t_zt, q_s, q_zt, U_zu, l_use_cs, l_use_wl, Cd, Ch, Ce, t_zu, q_zu, Ubzu,
 *                  Qsw, rad_lw, slp, pdT_cs, isecday_utc, plong, pdT_wl, pHz_wl, CdN, ChN, CeN, xz0, xu_star, xL, xUN10 )
 * (mod_blk_coare3p6.f90:123-131; same shape for turb_coare3p0 mod_blk_coare3p0.f90:54-59, turb_ecmwf mod_blk_ecmwf.f90:63-67,
 * turb_ncar mod_blk_ncar.f90:57-59 and turb_andreas mod_blk_andreas.f90:66-68, the last two without the skin arguments).
 *   T_s, q_s  : INOUT.  In: bulk SST [K] and its saturation humidity.  Out (use_cs or use_wl): skin temperature and
 *               0.98 q_sat(T_s, slp); untouched otherwise.
 *   theta_zt  : POTENTIAL air temperature at zt [K]; q_zt specific humidity [kg/kg]; U_zu scalar wind [m/s]
 *   Qsw       : NET solar flux (1-albedo) rad_sw; rad_lw; slp: required when use_cs or use_wl, else may be NULL
 *   Cd..Ubzu  : required outputs.  The OPTIONAL ones (CdN ... Hz_wl) are taken from ab_session_set_diagnostics
 *               (its first six members are ignored here).
 *   kt        : 1 initialises the warm-layer state (COARE3P6_INIT mod_blk_coare3p6.f90:250), which then persists in the
 *               session; solar time / longitude of WL_COARE come from ab_session_set_solar_time.
 *   nb_iter   : the reference reads the module variable nb_iter (mod_const.f90:33).
 * Same `mem` / stream rules as ab_session_compute.  The session's algorithm and precision apply; its use_skin flag does not. */
typedef struct ab_turb_fields {
    void *T_s;
    const void *theta_zt;
    void *q_s;
    const void *q_zt, *U_zu;
    const void *Qsw, *rad_lw, *slp;
    void *Cd, *Ch, *Ce, *t_zu, *q_zu, *Ubzu;
} ab_turb_fields;
int ab_session_turb(ab_session *s, int kt, double zt, double zu, int use_cs, int use_wl, int nb_iter,
                    const ab_turb_fields *f, int mem, void *stream);

/* The same on process-global state, for the Fortran modules mod_blk_coare3p6 / mod_blk_coare3p0 / mod_blk_ecmwf /
 * mod_blk_ncar / mod_blk_andreas of aerobulk_amd/fortran/mod_blk_turb.f90 (host fp64 arrays, ni*nj cells): one hidden
 * session per algorithm, re-created when the shape changes, warm-layer state re-initialised at kt == 1 — the module-level
 * SAVE arrays of the reference (mod_skin_coare.f90:31-36).  isecday_utc / lon: WL_COARE's solar time (lon may be NULL =
 * 0 deg everywhere; ignored by ECMWF).  `opt`: host arrays for the OPTIONAL outputs (may be NULL; first six members ignored). */
int ab_turb(int algo, int kt, double zt, double zu, int use_cs, int use_wl, int nb_iter, int isecday_utc, const double *lon,
            double *T_s, const double *theta_zt, double *q_s, const double *q_zt, const double *U_zu,
            const double *Qsw, const double *rad_lw, const double *slp,
            double *Cd, double *Ch, double *Ce, double *t_zu, double *q_zu, double *Ubzu,
            const ab_diag *opt, long ni, long nj);

/* Warm-layer state of the process-global TURB_<algo> session after the latest ab_turb() call, copied plane by plane into host arrays of
 * n = ni*nj doubles (a NULL plane is skipped; planes the scheme does not keep — Qnt_ac / Tau_ac for ECMWF — come back 0).  The reference
 * holds these as PUBLIC module arrays (mod_skin_coare.f90:31-36, mod_skin_ecmwf.f90:52-55) which callers read after TURB_*
 * (src/tests/test_aerobulk_buoy_series_oce.f90:16,463-464); aerobulk_amd/fortran/mod_blk_turb.f90 mirrors them with this call. */
int ab_turb_get_wl_state(int algo, double *dT_wl, double *Hz_wl, double *Qnt_ac, double *Tau_ac, long n);

/* TURB_NEUTRAL_10M( calgo, U_N10, CdN10, ChN10, CeN10, pz0 ) (mod_blk_neutral_10m.f90:33): neutral 10 m transfer coefficients
 * and roughness length from the neutral 10 m wind.  algo = AB_ALGO_COARE3P0 | COARE3P6 | ECMWF | NCAR (the reference STOPs for
 * andreas: AB_ERR_ALGO).  Stateless; n cells of `precision` in `mem`. */
int ab_turb_neutral_10m(int algo, int nb_iter, const void *U_N10, void *CdN10, void *ChN10, void *CeN10, void *z0, long n,
                        int precision, int mem, void *stream);

/* Sea-ice bulk algorithms (reference: src/ice/): TURB_ICE_NEMO mod_blk_ice_nemo.f90:36-38, TURB_ICE_AN05
 * mod_blk_ice_an05.f90:41-43, TURB_ICE_LU12 mod_blk_ice_lu12.f90:69-71, TURB_ICE_LG15 mod_blk_ice_lg15.f90:68-70, e.g.
 *   TURB_ICE_LG15( zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu, frice, Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu,
 *                  CdN, ChN, CeN, xz0, xu_star, xL, xUN10 )
 * Stateless (no session): n cells, arrays of `precision` (enum ab_precision) in `mem`; AB_MEM_DEVICE enqueues on `stream`.
 *   Ts_i, qs_i : ice surface temperature [K] and saturation humidity over ice; theta_zt POTENTIAL air temperature at zt
 *   frice      : ice concentration, required by LU12 and LG15 (NULL otherwise).  NB LG15 reproduces the reference, whose
 *                CdN_f_LG15_light (mod_cdn_form_ice.f90:304) gives every cell the form drag of the LAST cell of the array
 *   Cd..Ub     : required outputs; CdN..UN10 optional (NULL = not wanted) */
enum ab_ice_algo { AB_ICE_NEMO = 1, AB_ICE_AN05 = 2, AB_ICE_LU12 = 3, AB_ICE_LG15 = 4, AB_ICE_EASY = 5 };
typedef struct ab_ice_fields {
    const void *Ts_i, *theta_zt, *qs_i, *q_zt, *U_zu, *frice;
    void *Cd, *Ch, *Ce, *t_zu, *q_zu, *Ub;
    void *CdN, *ChN, *CeN, *z0, *u_star, *L, *UN10;
    void *CdN_frm;   /* AB_ICE_LG15 only: form-drag part of CdN, the CdN_frm of TURB_ICE_LG15_IO (mod_blk_ice_lg15_io.f90:71,346) */
} ab_ice_fields;
int ab_turb_ice(int ice_algo, double zt, double zu, int nb_iter, const ab_ice_fields *f, long n, int precision, int mem,
                void *stream);
int ab_ice_algo_from_string(const char *calgo);
/* TURB_ICE_EASY( zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu, CdN, ChN, CeN, Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu, xz0, xu_star, xL, xUN10 )
 * (mod_blk_ice_easy.f90:44-47): the neutral coefficients are prescribed SCALARS; f->CdN / ChN / CeN, if given, receive them back. */
int ab_turb_ice_easy(double zt, double zu, int nb_iter, double CdN, double ChN, double CeN, const ab_ice_fields *f, long n,
                     int precision, int mem, void *stream);

/* ---- the public helper functions of mod_phymbl -------------------------------------------------------------------------------
 * What callers of the reference import with `USE mod_phymbl` next to aerobulk_model (generics mod_phymbl.f90:33-139; e.g.
 * src/tests/example_call_aerobulk.f90:7,59 Theta_from_z_P0_T_q, aerobulk_toy.F90, test_cx_vs_wind.f90, the sea-ice drivers), as ONE
 * elementwise entry: function `fn` on n cells.  The Fortran module aerobulk_amd/fortran/mod_phymbl.f90 binds every `_sclr` / `_vctr`
 * specific of the reference to it; a scalar call is n = 1 (the library has no host arithmetic).
 *   in[0..n_in)   the function's ARRAY arguments in the order of the reference's dummy arguments with the scalars taken out;
 *                 an OPTIONAL array that is not passed is NULL (or beyond n_in)
 *   out[0..n_out) the results; NULL members are not written (at least one must be given)
 *   par[0]        the function's scalar REAL argument if it has one: pz / pzu, or pPref when in[2] is NULL (pot_temp, abs_temp);
 *                 par[1]: zu of AB_PH_FIRST_GUESS_COARE (par[0] = zt); else reserved (pass 0)
 *   flag          its LOGICAL / INTEGER argument: l_ice (0 / 1), iflag of z0tq_LKB (1 temperature, 2 humidity)
 *   mem           AB_MEM_HOST: arrays staged through HBM; AB_MEM_DEVICE: device arrays, enqueued on `stream` (hipStream_t).  Both
 *                 return when the results are complete.
 *   info          may be NULL; AB_PH_BULK_FORMULA: info[0] = index of the first cell whose wind stress exceeds 10 N/m^2 (-1: none),
 *                 info[1] = that stress; the call then returns AB_ERR_TAU with every output written (BULK_FORMULA_VCTR's STOP,
 *                 mod_phymbl.f90:1250-1253)
 * n == 0 is a no-op returning AB_OK (zero-size Fortran arrays).  AB_MEM_DEVICE work runs on the device that owns the first input array.
 * AB_PH_E_AIR / RHO_AIR_ADV / RH_AIR: AB_ERR_NOCONV if e_air's whole-array fixed point has not converged after 200 sweeps (the
 * reference would not return); the outputs then hold the last iterate.
 * fn (arrays in ; scalar ; arrays out), reference lines in mod_phymbl.f90: */
enum ab_phymbl_fn {
    AB_PH_POT_TEMP = 1,             /* pTa, pPz, [pPref] ; pPref ; theta                     :163-200 */
    AB_PH_ABS_TEMP = 2,             /* pThta, pPz, [pPref] ; pPref ; T                       :205-242 */
    AB_PH_VIRT_TEMP = 3,            /* pTa, pqa ; - ; T_v                                    :247-276 */
    AB_PH_PZ_FROM_P0_TZ_QZ = 4,     /* pslp, pTa, pqa ; pz, l_ice ; P(z)                     :283-337 */
    AB_PH_THETA_FROM_Z_P0_T_Q = 5,  /* pslp, pTa, pqa ; pz, l_ice ; theta                    :343-375 */
    AB_PH_T_FROM_Z_P0_THETA_Q = 6,  /* pslp, pThta, pqa ; pz, l_ice ; T                      :380-421 */
    AB_PH_RHO_AIR = 7,              /* pTa, pqa, pslp ; - ; rho                              :522-546 */
    AB_PH_VISC_AIR = 8,             /* pTa ; - ; nu                                          :549-574 */
    AB_PH_L_VAP = 9,                /* psst ; - ; L_v                                        :579-598 */
    AB_PH_CP_AIR = 10,              /* pqa ; - ; Cp                                          :603-622 */
    AB_PH_GAMMA_MOIST = 11,         /* pTa, pqa ; - ; moist lapse rate                       :627-661 */
    AB_PH_ONE_ON_L = 12,            /* pThta, pqa, pus, pts, pqs ; - ; 1/L                   :666-708 */
    AB_PH_RI_BULK = 13,             /* psst, pThta, pssq, pqa, pub, [pTa_layer, pqa_layer] ; pz ; Ri_b   :712-772 */
    AB_PH_E_SAT = 14,               /* pTa ; - ; e_s over water [Pa]                         :777-811 */
    AB_PH_E_SAT_ICE = 15,           /* pTa ; - ; e_s over ice                                :815-843 */
    AB_PH_DE_SAT_DT_ICE = 16,       /* pTa ; - ; d e_s,ice / dT                              :845-875 */
    AB_PH_Q_SAT = 17,               /* pTa, pslp ; l_ice ; q_s                               :881-921 */
    AB_PH_DQ_SAT_DT_ICE = 18,       /* pTa, pslp ; - ; d q_s,ice / dT                        :926-958 */
    AB_PH_Q_AIR_RH = 19,            /* prha [%], pTa, pslp ; - ; q                           :963-985 */
    AB_PH_Q_AIR_DP = 20,            /* da, slp ; - ; q                                       :990-1000 */
    AB_PH_RHO_AIR_ADV = 21,         /* pTa, pqa, pslp ; - ; rho (true virtual temperature)   :1008-1024 */
    AB_PH_Q_SAT_CRUDE = 22,         /* pts, prhoa ; - ; q_s                                  :1029-1038 */
    AB_PH_DRY_STATIC_ENERGY = 23,   /* pTa, pqa ; pz ; s                                     :1043-1054 */
    AB_PH_UPDATE_QNSOL_TAU = 24,    /* pts, pqs, pThta, pqa, pust, ptst, pqst, pwnd, pUb, pslp, prlw ; pzu ; pQns, pTau, Qlat   :1059-1144 */
    AB_PH_BULK_FORMULA = 25,        /* pts, pqs, pThta, pqa, pCd, pCh, pCe, pwnd, pUb, pslp ; pzu, l_ice ; pTau, pQsen, pQlat, pEvap, prhoa   :1149-1261 */
    AB_PH_ALPHA_SW = 26,            /* psst ; - ; alpha                                      :1267-1286 */
    AB_PH_QLW_NET = 27,             /* pdwlw, pts ; l_ice ; net long-wave                    :1291-1330 */
    AB_PH_Z0_FROM_CD = 28,          /* pCd, [ppsi] ; pzu ; z0                                :1335-1366 */
    AB_PH_Z0_FROM_USTAR = 29,       /* pus, puzu ; pzu ; z0                                  :1371-1391 */
    AB_PH_CD_FROM_Z0 = 30,          /* pz0, [ppsi] ; pzu ; Cd                                :1396-1414 */
    AB_PH_F_M_LOUIS = 31,           /* pRib, pCdn, pz0 ; pzu ; f_m                           :1419-1453 */
    AB_PH_F_H_LOUIS = 32,           /* pRib, pChn, pz0 ; pzu ; f_h                           :1458-1492 */
    AB_PH_UN10_FROM_USTAR = 33,     /* pUzu, pus, ppsi ; pzu ; UN10                          :1498-1510 */
    AB_PH_UN10_FROM_CDN = 34,       /* pUb, pCdn, ppsi ; pzu ; UN10                          :1515-1527 */
    AB_PH_UN10_FROM_CD = 35,        /* pUb, pCd, ppsi ; pzu ; UN10                           :1532-1558 */
    AB_PH_Z0TQ_LKB = 36,            /* pRer, pz0 ; iflag ; z0t | z0q                         :1635-1701 */
    AB_PH_E_AIR = 37,               /* pqa, pslp ; - ; e  (fixed point on the WHOLE array: SUM |change| <= 1e-6)   :1706-1736 */
    AB_PH_RH_AIR = 38,              /* pqa, pTa, pslp ; - ; RH [%]                           :1741-1753 */
    AB_PH_DELTA_SKIN_LAYER = 39,    /* palpha, pQd, pustar_a, [Qlat] ; - ; delta             :2010-2046 */
    /* the two PUBLIC helpers of src/ice/mod_blk_ice_an05.f90 (Andreas et al. 2005), used by src/ice/test_ice.f90:49,55 */
    AB_PH_ROUGH_LENG_M = 40,        /* pus, pnua ; - ; z0 over sea ice                       mod_blk_ice_an05.f90:232-255 */
    AB_PH_ROUGH_LENG_TQ = 41,       /* pz0, pus, pnua ; - ; z0t, z0q                         mod_blk_ice_an05.f90:257-312 */
    /* the PUBLIC functions of the algorithm modules: stability functions of zeta = z/L, Charnock parameters, NCAR's neutral
     * coefficients, ANDREAS' friction velocity (the device functions the flux kernels iterate with) */
    AB_PH_PSI_M_COARE = 42,         /* pzeta ; - ; psi_m                                     mod_common_coare.f90:217-302 */
    AB_PH_PSI_H_COARE = 43,         /* pzeta ; - ; psi_h                                     mod_common_coare.f90:305-392 */
    AB_PH_PSI_M_NCAR = 44,          /* pzeta ; - ; psi_m                                     mod_blk_ncar.f90:333-376 */
    AB_PH_PSI_H_NCAR = 45,          /* pzeta ; - ; psi_h                                     mod_blk_ncar.f90:379-420 */
    AB_PH_PSI_M_ECMWF = 46,         /* pzeta ; - ; psi_m                                     mod_blk_ecmwf.f90:441-495 */
    AB_PH_PSI_H_ECMWF = 47,         /* pzeta ; - ; psi_h                                     mod_blk_ecmwf.f90:498-548 */
    AB_PH_PSI_M_ANDREAS = 48,       /* pzeta ; - ; psi_m                                     mod_blk_andreas.f90:307-360 */
    AB_PH_PSI_H_ANDREAS = 49,       /* pzeta ; - ; psi_h                                     mod_blk_andreas.f90:363-410 */
    AB_PH_CHARN_COARE3P0 = 50,      /* pwnd ; - ; Charnock parameter                         mod_blk_coare3p0.f90:420-447 */
    AB_PH_CHARN_COARE3P6 = 51,      /* pwnd ; - ; Charnock parameter                         mod_blk_coare3p6.f90:417-445 */
    AB_PH_CD_N10_NCAR = 52,         /* pw10 ; - ; CdN10                                      mod_blk_ncar.f90:244-284 */
    AB_PH_CH_N10_NCAR = 53,         /* psqrtcdn10, pstab ; - ; ChN10                         mod_blk_ncar.f90:287-310 */
    AB_PH_CE_N10_NCAR = 54,         /* psqrtcdn10 ; - ; CeN10                                mod_blk_ncar.f90:313-330 */
    AB_PH_U_STAR_ANDREAS = 55,      /* pun10 ; - ; u*                                        mod_blk_andreas.f90:275-305 */
    AB_PH_FIRST_GUESS_COARE = 56,   /* psst, t_zt, pssq, q_zt, U_zu, pcharn ; zt, zu (par[0], par[1]) ; pus, pts, pqs, t_zu, q_zu, Ubzu, pz0
                                                                                              mod_common_coare.f90:33-214 */
    /* the skin schemes as mod_skin_coare / mod_skin_ecmwf export them: one call of the scheme per cell, on the caller's state arrays */
    AB_PH_CS_COARE = 57,            /* pQsw, pQnsol, pustar, pSST, pQlat ; - ; pdT_cs                mod_skin_coare.f90:48-93 */
    AB_PH_CS_ECMWF = 58,            /* pQsw, pQnsol, pustar, pSST ; - ; pdT_cs                       mod_skin_ecmwf.f90:68-110 */
    AB_PH_WL_COARE = 59,            /* pQsw, pQnsol, pTau, pSST, plon, dT_wl, Hz_wl, Qnt_ac, Tau_ac ; isd (par[0]), iwait (flag) ;
                                       dT_wl, Hz_wl, Qnt_ac, Tau_ac after the call (unchanged if iwait /= 0)   mod_skin_coare.f90:97-250 */
    AB_PH_WL_ECMWF = 60             /* pQsw, pQnsol, pustar, pSST, dT_wl, Hz_wl, [pustk] ; - ; dT_wl after the call  mod_skin_ecmwf.f90:113-230 */
};
int ab_phymbl(int fn, long n, const double *const *in, int n_in, double *const *out, int n_out, const double *par, int flag,
              int mem, void *stream, double *info);

/* Copy the persistent warm-layer state (planes dT_wl, Hz_wl, Qnt_ac, Tau_ac; ECMWF uses the
 * first two) to host doubles — diagnostics pdT_wl/pHz_wl of TURB_COARE3P6, mod_blk_coare3p6.f90:406-407. */
int ab_session_get_wl_state(ab_session *s, double *state4n);

/* Duration in ms of the most recent flux kernel launched by this session (HIP events on the
 * session's stream); < 0 if unavailable.  Synchronises. */
double ab_session_last_kernel_ms(ab_session *s);

/* Test/bench utility: fill device arrays with the deterministic quasi-random fields of SURVEY.md §8d
 * (rows j0..j0+nj_local-1, 0-based, of an ni-wide grid; rad_sw/rad_lw may be NULL).  `precision` is an
 * enum ab_precision; evaluation is always fp64. */
int ab_synth_fields_device(void *sst, void *t_zt, void *q_zt, void *u_zu, void *v_zu, void *slp,
                           void *rad_sw, void *rad_lw, long ni, long j0, long nj_local, int precision,
                           void *stream);

/* Box calibration for benchmarks (bench.py `calib`; aerobulk_amd/csrc/ab_calib.hip): a fixed device workload that depends on the
 * box alone, timed with HIP events on `stream` of `device` after one untimed launch.  AB_CALIB_FMA_F64: chains of v_fma_f64 on every
 * lane, four waves per SIMD -> *rate = fp64 TFLOP/s (78.6 at 2.4 GHz: the resource that binds the flux kernels);
 * AB_CALIB_HBM_COPY: dst = src over 1 GiB -> *rate = GB/s read + written.  *ms = the timed launch.  No counterpart in the
 * reference (measurement contract). */
enum ab_calib_workload { AB_CALIB_FMA_F64 = 0, AB_CALIB_HBM_COPY = 1 };
int ab_calibrate(int what, int device, void *stream, double *ms, double *rate);

/* Test hook: apply the engine's fp64 device math function `op` elementwise to host arrays (y may be NULL):
 * 0 div 1 rcp 2 sqrt 3 log 4 log10 5 exp 6 exp10 7 atan 8 cbrt 9 rcbrt 10 e_sat 11 pow 12 x^(-1/4) 13 e_sat through the
 * piecewise LDS table of the tiled flux kernels.  tests/test_gpu_math.py */
int ab_test_math(int op, const double *x, const double *y, double *out, long n);

/* ---- the reference's own entry points --------------------------------------------------- */
/* AEROBULK_MODEL (mod_aerobulk.f90:176-269) on a process-global session, with the reference's
 * call protocol: INIT at jt==1, BYE at jt==nt.  Non-reentrant exactly like the reference.
 * rad_sw/rad_lw/t_s may be NULL (the Fortran OPTIONALs); niter <= 0 keeps the current nb_iter
 * (sticky, default 5: mod_const.f90:33, mod_aerobulk.f90:236).  Host arrays, fp64.
 * Returns an ab_status; `report` (may be NULL) is filled when jt==1.
 * jt == 1 follows the reference's order (mod_aerobulk.f90:246-262): AEROBULK_INIT's checks first — on an AB_ERR_ALL_MASKED,
 * AB_ERR_HUM_TYPE or AB_ERR_UNITS the output arrays are untouched — then aerobulk_compute on the fields the checks have just staged
 * in HBM (one PCIe crossing).  AEROBULK_AMD_FUSED_INIT=1 opts into a one-pass first record on grids of 4 Mi cells and more (per shard,
 * every shard): the statistics ride on the pipelined pass of the compute, whose kernels start from the humidity type the first 2^20
 * cells indicate (about 10 ms less, once per run, on an ORCA12 grid).  Consequence of the opt-in: when an AEROBULK_INIT error is
 * returned the output arrays already hold fluxes computed with a GUESSED humidity type, and if the guess was wrong record 1 is
 * computed twice (a one-line notice on stderr). */
int ab_model(int jt, int nt, const char *calgo, int calgo_len, double zt, double zu,
             const double *sst, const double *t_zt, const double *hum_zt,
             const double *u_zu, const double *v_zu, const double *slp,
             double *ql, double *qh, double *tau_x, double *tau_y, double *evap,
             int niter, int use_skin, const double *rad_sw, const double *rad_lw, double *t_s,
             long ni, long nj, ab_init_report *report);

/* Exactly the two C symbols the reference exports from its Fortran shim
 * (mod_aerobulk_cxx.f90:29-33,66-69; prototypes aerobulk.cpp:5-19): everything by reference,
 * `calgo` is char[l+1], data are flat length-m vectors.  `l_skin` is read as ONE byte (C `bool`,
 * aerobulk.cpp:10); the reference's Fortran side reads a 4-byte LOGICAL there (ABI slip).
 * On error they print the reference's message and terminate the process, like `STOP`. */
void aerobulk_cxx_skin(const int *jt, const int *nt, const char *calgo, const double *zt, const double *zu,
                       const double *sst, const double *t_zt, const double *hum_zt, const double *u_zu,
                       const double *v_zu, const double *slp,
                       double *ql, double *qh, double *tau_x, double *tau_y, double *evap,
                       const int *niter, const bool *l_skin, const double *rad_sw, const double *rad_lw,
                       double *t_s, const int *l, const int *m);
void aerobulk_cxx_no_skin(const int *jt, const int *nt, const char *calgo, const double *zt, const double *zu,
                          const double *sst, const double *t_zt, const double *hum_zt, const double *u_zu,
                          const double *v_zu, const double *slp,
                          double *ql, double *qh, double *tau_x, double *tau_y, double *evap,
                          const int *niter, const int *l, const int *m);

#ifdef __cplusplus
}
#endif
#endif /* AEROBULK_AMD_H */
