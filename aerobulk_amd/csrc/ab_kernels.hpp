// ab_kernels.hpp — launch interface between the C-ABI runtime (ab_runtime.hip) and the
// device kernels (ab_kernels.hip).  Internal; not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

namespace ab {

// Wave-uniform description of one aerobulk_compute() call (mod_aerobulk_compute.f90:22-213).
struct FluxCall {
    // inputs (device pointers; element type follows `f32`)
    const void *sst, *t_zt, *hum, *u, *v, *slp, *rad_sw, *rad_lw, *lon;
    // outputs
    void *ql, *qh, *tau_x, *tau_y, *evap, *t_s;
    // warm-layer state planes dT_wl, Hz_wl, Qnt_ac, Tau_ac (mod_skin_coare.f90:31-36)
    void *wl[4];
    void *diag[16]; // optional TURB_* diagnostics (all nullptr: the lean kernels run); order of ab_diag in the C ABI
    int *flags;     // bit0: wind stress > 10 N/m^2 somewhere (mod_phymbl.f90:1250)
    long n;
    double zt, zu;
    int algo;       // enum ab_algo
    int skin;       // cool-skin + warm-layer
    int f32;        // element type of the arrays
    int compute64;  // with f32: 1 = fp64 arithmetic on fp32 arrays (AB_F32_STORAGE), 2 = fp64 anchors, fp32 elsewhere (AB_F32_MIXED)
    int nb_iter;
    int hum_type;   // enum ab_hum
    int wl_load;    // jt > 1: read state ; else initial values
    int wl_store;   // jt < nt: write state back
    int isecday;    // UTC seconds of day for WL_COARE (12 in aerobulk_compute)
    int regroup;    // 1: regroup the cells of a tile into like-behaved waves (default), 0: natural order
};

hipError_t launch_flux(const FluxCall &c, hipStream_t stream);

// One call of a TURB_<algo> routine (e.g. mod_blk_coare3p6.f90:123-131): no pre-processing, no bulk formula.
struct TurbCall {
    void *T_s, *q_s;                         // INOUT: overwritten with the skin values when skin != 0
    const void *theta_zt, *q_zt, *U_zu;      // potential temperature and specific humidity at zt, scalar wind at zu
    const void *qsw, *rad_lw, *slp, *lon;    // net solar, downwelling LW, SLP (skin != 0 only); lon: WL_COARE solar time
    void *out[16];                           // Cd Ch Ce t_zu q_zu Ubzu (required) | CdN..UN10, dT_cs dT_wl Hz_wl (nullptr: skip)
    void *wl[4];
    long n;
    double zt, zu;
    int algo;
    int skin;        // bit 0: l_use_cs, bit 1: l_use_wl
    int f32, nb_iter, wl_load, wl_store, isecday;
    int regroup;     // as FluxCall::regroup
};
hipError_t launch_turb(const TurbCall &c, hipStream_t stream);

// turb_neutral_10m (mod_blk_neutral_10m.f90:33): neutral 10 m coefficients from the neutral 10 m wind
hipError_t launch_neutral10(int algo, int nb_iter, const void *U_N10, void *CdN10, void *ChN10, void *CeN10, void *z0, long n,
                            int f32, hipStream_t stream);

// One call of a sea-ice TURB_ICE_<algo> routine (src/ice/mod_blk_ice_*.f90).
struct IceCall {
    const void *Ts_i, *theta_zt, *qs_i, *q_zt, *U_zu, *frice;   // frice: lu12 (per cell), lg15 (its LAST element only)
    void *out[14];                                              // Cd Ch Ce t_zu q_zu Ub (required) | CdN ChN CeN z0 u_star L UN10 | CdN_frm (LG15)
    long n;
    double zt, zu;
    int algo;        // enum ab_ice_algo
    int f32, nb_iter;
    double cxn[3];   // TURB_ICE_EASY: prescribed CdN, ChN, CeN
};
hipError_t launch_turb_ice(const IceCall &c, hipStream_t stream);

// AEROBULK_INIT statistics (mod_aerobulk.f90:104-153): per-block partial reductions.
//  fields order: 0 sst,1 t_air,2 slp,3 u,4 v,5 wnd,6 hum,7 rad_sw,8 rad_lw
//  partials layout per block: [count, then for each of 9 fields: sum, min, max] = 28 doubles
constexpr int kStatFields = 9;
constexpr int kStatStride = 1 + 3 * kStatFields;
constexpr int kStatBlocks = 2048;
hipError_t launch_init_stats(const void *sst, const void *t_air, const void *hum, const void *u, const void *v,
                             const void *slp, const void *rad_sw, const void *rad_lw, long n, int f32,
                             double *partials /* nblocks*kStatStride */, hipStream_t stream, int nblocks = kStatBlocks);

// Synthetic quasi-random fields of SURVEY.md §8d generated straight into HBM (bench utility):
// rows j0 .. j0+nj_local-1 (0-based) of an ni-wide grid.
hipError_t launch_synth(void *sst, void *t_zt, void *q_zt, void *u, void *v, void *slp, void *rad_sw,
                        void *rad_lw, long ni, long j0, long nj_local, int f32, hipStream_t stream);

// unit-test hook: apply fp64 device math function `op` elementwise (tests/test_gpu_math.py)
hipError_t launch_math_test(int op, const double *x, const double *y, double *o, long n, hipStream_t stream);

}  // namespace ab
