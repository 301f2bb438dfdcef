// ab_session.hpp — the session object behind the opaque `ab_session` of include/aerobulk_amd.h, shared by the leaf runtime
// (ab_runtime.hip: one device) and the row-block sharding layer (ab_sharded.hip: several devices, or several shards of one).
// Internal; not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/aerobulk_amd.h"

struct ab_session {
    int algo = 0, nt = 1, use_skin = 0, f32 = 0, compute64 = 0, device = 0;   // f32: fp32 arrays; compute64: 1 = ... with fp64 arithmetic (AB_F32_STORAGE), 2 = fp64 anchors + fp32 transcendentals (AB_F32_MIXED)
    long ni = 0, nj = 0, n = 0;
    size_t esz = 8;
    int hum_type = AB_HUM_SH;
    int last_jt = 0;
    int isecday = 12;                 // mod_aerobulk_compute.f90:136,146
    int regroup = 1;                  // lane regrouping of flux_kernel (ab_session_set_regroup)
    void *d_lon = nullptr;            // optional longitude field (device, session-owned copy)
    void *wl[4] = {nullptr, nullptr, nullptr, nullptr};
    int *d_flags = nullptr;
    double *d_partials = nullptr;
    double *d_fused = nullptr;        // partial rows of the statistics that ride on the pipelined first record (kept: no malloc / free per loop)
    size_t d_fused_rows = 0;
    void *stage_in[8] = {nullptr};    // device staging for AB_MEM_HOST callers
    void *stage_out[6] = {nullptr};
    void *diag_user[16] = {nullptr};  // caller's diagnostic arrays (ab_session_set_diagnostics), host or device
    void *diag_dev[16] = {nullptr};   // device staging when the caller's arrays are host memory
    int diag_mem = AB_MEM_DEVICE;
    bool diag_on = false;
    hipStream_t stream = nullptr;     // session stream for host-mem calls (kernels)
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;  // copy streams of the pipelined host path
    hipStream_t last_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool done_pending = false;        // a device-mode call is in flight on the caller's stream: destroy waits for ev1, recorded behind its
                                      // kernel (the caller may have destroyed its stream by then; the event is ours)
    bool timed = false;
    // AEROBULK_MODEL at jt == 1 (ab_model): the fields AEROBULK_INIT staged in HBM are the ones aerobulk_compute reads next.
    // staged_from[i] = the host array stage_in[i] was last filled from; reuse_staged = honour it in the next host compute
    const void *staged_from[8] = {nullptr};
    bool reuse_staged = false;
    // ---- row-block sharding: a parent session owns no device state, its shards do.  Shard r covers the rows
    // [shard_j0[r], shard_j0[r] + shard_njl[r]) of the (ni, nj) grid = the cells [ni*j0, ni*(j0+njl)) of every flat field.
    std::vector<ab_session *> shards;
    std::vector<long> shard_j0, shard_njl;
    bool sharded() const { return !shards.empty(); }
    void *pool = nullptr;             // ab_sharded.hip: persistent worker threads, one per shard (host-array calls)
    void *gather = nullptr;           // ab_sharded.hip: RCCL communicator over the session's distinct devices (created at the first gather)
};

namespace ab {
// The library switches the calling thread's current HIP device to the session's; callers (torch, a model's own HIP code) must find
// theirs unchanged afterwards: every public entry point that calls hipSetDevice holds one of these.
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
}  // namespace ab

namespace ab {
// last-error plumbing across the worker threads of the sharding layer (ab_last_error() is per thread)
void set_last_error(const std::string &msg);
// Row-block sharding layer (ab_sharded.hip); each function mirrors the public entry point of the same name and is what that
// entry point calls for a session with shards.
int sharded_create(ab_session **out, int algo, long ni, long nj, int nt, int use_skin, int precision, const int *devices, int nshards,
                   const long *nj_per_shard = nullptr);
int sharded_destroy(ab_session *s);
int sharded_init_stats(ab_session *s, const void *const in[8], int mem, void *stream, double stats[AB_INIT_NSTATS]);
int sharded_compute(ab_session *s, int jt, double zt, double zu, int niter, const void *const in[8], void *const out[6], int mem,
                    void *stream);
int sharded_turb(ab_session *s, int kt, double zt, double zu, int use_cs, int use_wl, int nb_iter, const ab_turb_fields *f, int mem,
                 void *stream);
int sharded_check(ab_session *s);
int sharded_set_solar_time(ab_session *s, int isecday_utc, const void *lon, int mem, void *stream);
int sharded_set_diagnostics(ab_session *s, const ab_diag *d, int mem);
int sharded_get_wl_state(ab_session *s, double *state4n);
double sharded_last_kernel_ms(ab_session *s);
int sharded_compute_shards(ab_session *s, int jt, double zt, double zu, int niter, const ab_shard_arrays *sh, void *const *streams);
int sharded_gather(ab_session *s, int root, const ab_shard_arrays *sh, const ab_flux_arrays *dst, void *const *streams, int synchronize);
// AEROBULK_MODEL at jt == 1 through a sharded session: AEROBULK_INIT's statistics ride on every shard's pipelined pass
int sharded_prepare_staging(ab_session *s, int with_rad, int with_ts);
int sharded_model_first_record(ab_session *s, double zt, double zu, int niter, const void *const in[8], void *const out[6], int have_rad,
                               ab_init_report *report);
// leaf side of it (ab_runtime.hip): one shard's fused pass (statistics returned, not applied), and its redo with another humidity type
struct FusedShard;
int leaf_fused_first_record(ab_session *leaf, double zt, double zu, int niter, const void *const in[8], void *const out[6], int have_rad,
                            double stats[AB_INIT_NSTATS], int *guess, FusedShard **keep);
int leaf_fused_redo(ab_session *leaf, FusedShard *keep);
int leaf_prepare_staging(ab_session *leaf, int with_rad, int with_ts);
void leaf_fused_release(FusedShard *keep);
}  // namespace ab
