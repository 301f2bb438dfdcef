// ab_sharded.hip — row-block sharding of one session over several GPUs (the GCM coupling call site
// AEROBULK_MODEL, mod_aerobulk.f90:250-262, hands over whole (Ni,Nj) host arrays).
//
// The path is pointwise: no halo, no collective on the data path.  A sharded session cuts the grid into contiguous blocks of
// rows (the second Fortran dimension: contiguous memory ranges of every flat field) and gives each block to a LEAF session
// (ab_runtime.hip) on its own device, with its own streams, staging buffers and warm-layer state.  A host-array call runs the
// shards concurrently, one host thread each, so that every device stages its rows over its own PCIe link; the only exchange
// is AEROBULK_INIT's statistics (29 doubles per shard, combined on the host by SUM / MIN / MAX exactly as the ranks of a
// multi-process run would all-reduce them).  Several shards may share one device (tests; results are bit-identical to the
// unsharded session either way, tests/test_gpu_sharded.py).  Device-resident callers normally own one session per GPU and
// do not need this layer; it accepts device arrays only when every shard lives on the device that holds them.
#include "ab_kernels.hpp"
#include "ab_session.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types, enumerators, prototypes only: librccl.so is dlopen()ed on first use, never linked

#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

namespace {

int sfail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    ab::set_last_error(buf);
    return code;
}

inline const void *off(const ab_session *s, const void *p, long j0) { return p ? (const char *)p + (size_t)j0 * s->ni * s->esz : nullptr; }
inline void *offw(const ab_session *s, void *p, long j0) { return p ? (char *)p + (size_t)j0 * s->ni * s->esz : nullptr; }

// One persistent worker thread per shard, created with the session (round 2 created and joined a std::thread per shard per CALL: per
// record of a time loop; visible at config-1 sizes, 19 us records).  run(job) hands job(r) to worker r and returns when all are done.
class ShardPool {
public:
    explicit ShardPool(int n) : n_(n), state_(n, 0)
    {
        for (int r = 0; r < n; ++r) th_.emplace_back([this, r] { loop(r); });
    }
    ~ShardPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void run(const std::function<void(int)> &job)
    {
        std::unique_lock<std::mutex> lk(m_);
        job_ = &job;
        pending_ = n_;
        for (auto &x : state_) x = 1;
        cv_.notify_all();
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    void loop(int r)
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [&] { return quit_ || state_[r] == 1; });
            if (quit_) return;
            state_[r] = 0;
            const std::function<void(int)> *job = job_;
            lk.unlock();
            (*job)(r);
            lk.lock();
            if (--pending_ == 0) done_.notify_all();
        }
    }
    int n_;
    std::vector<int> state_;
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *job_ = nullptr;
    int pending_ = 0;
    bool quit_ = false;
};

// Run fn(r) for every shard: concurrently (the shard's persistent worker) when the call blocks on transfers, in order otherwise.
// Returns the first non-zero status in shard order, with that shard's error text.
template <class F> int for_shards(ab_session *s, bool concurrent, F fn)
{
    const int n = (int)s->shards.size();
    std::vector<int> rc(n, AB_OK);
    std::vector<std::string> msg(n);
    ab::DeviceGuard dguard_;      // the leaf entry points switch the current device: the caller gets its own back
    if (concurrent && n > 1) {
        if (!s->pool) s->pool = new ShardPool(n);
        const std::function<void(int)> job = [&](int r) {
            rc[r] = fn(r);
            if (rc[r]) msg[r] = ab_last_error();
        };
        static_cast<ShardPool *>(s->pool)->run(job);
    } else {
        for (int r = 0; r < n; ++r) {
            rc[r] = fn(r);
            if (rc[r]) msg[r] = ab_last_error();
        }
    }
    // AB_ERR_TAU is the reference's "some cell exceeded 10 N/m^2": it must not hide a harder failure of another shard
    int first = AB_OK;
    for (int r = 0; r < n; ++r)
        if (rc[r] && (first == AB_OK || (first == AB_ERR_TAU && rc[r] != AB_ERR_TAU))) {
            first = rc[r];
            ab::set_last_error(msg[r]);
        }
    return first;
}

// device arrays are only meaningful to a shard that lives on the device holding them
int check_device_arrays(ab_session *s, const void *any)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, any) != hipSuccess) {
        (void)hipGetLastError();
        return sfail(AB_ERR_ARG, "sharded session: AB_MEM_DEVICE arrays are not device memory");
    }
    for (ab_session *c : s->shards)
        if (c->device != at.device)
            return sfail(AB_ERR_ARG,
                         "sharded session: device arrays live on GPU %d but a shard runs on GPU %d.  A sharded session takes host "
                         "arrays (every device stages its own rows); device-resident callers own one session per GPU",
                         at.device, c->device);
    return AB_OK;
}

}  // namespace

namespace ab {

void gather_release(ab_session *s);

int sharded_create(ab_session **out, int algo, long ni, long nj, int nt, int use_skin, int precision, const int *devices, int nshards,
                   const long *nj_per_shard)
{
    ab::DeviceGuard dguard_;
    ab_session *p = new ab_session;
    p->algo = algo; p->ni = ni; p->nj = nj; p->n = ni * nj; p->nt = nt; p->use_skin = use_skin ? 1 : 0;
    p->f32 = (precision != AB_F64); p->compute64 = (precision == AB_F32_STORAGE) ? 1 : (precision == AB_F32_MIXED ? 2 : 0); p->esz = p->f32 ? 4 : 8; p->device = devices ? devices[0] : 0;
    // contiguous j-blocks: the caller's row counts, else equal blocks with the first (nj mod nshards) one row taller
    long j0 = 0;
    for (int r = 0; r < nshards; ++r) {
        const long njl = nj_per_shard ? nj_per_shard[r] : nj / nshards + (r < nj % nshards ? 1 : 0);
        ab_session *c = nullptr;
        int rc = ab_session_create(&c, algo, ni, njl, nt, use_skin, precision, devices ? devices[r] : r);
        if (rc) {
            const std::string why = ab_last_error();
            sharded_destroy(p);
            set_last_error(why);
            return rc;
        }
        p->shards.push_back(c);
        p->shard_j0.push_back(j0);
        p->shard_njl.push_back(njl);
        j0 += njl;
    }
    p->regroup = p->shards[0]->regroup;
    *out = p;
    return AB_OK;
}

int sharded_destroy(ab_session *s)
{
    ab::DeviceGuard dguard_;
    delete static_cast<ShardPool *>(s->pool);
    s->pool = nullptr;
    gather_release(s);
    for (ab_session *c : s->shards) ab_session_destroy(c);
    s->shards.clear();
    delete s;
    return AB_OK;
}

int sharded_init_stats(ab_session *s, const void *const in[8], int mem, void *stream, double stats[AB_INIT_NSTATS])
{
    if (mem == AB_MEM_DEVICE) {
        int rc = check_device_arrays(s, in[0]);
        if (rc) return rc;
    }
    const int n = (int)s->shards.size();
    std::vector<double> part((size_t)n * AB_INIT_NSTATS);
    int rc = for_shards(s, mem == AB_MEM_HOST, [&](int r) {
        const long j0 = s->shard_j0[r];
        return ab_session_init_stats(s->shards[r], off(s, in[0], j0), off(s, in[1], j0), off(s, in[2], j0), off(s, in[3], j0),
                                     off(s, in[4], j0), off(s, in[5], j0), off(s, in[6], j0), off(s, in[7], j0), mem, stream,
                                     &part[(size_t)r * AB_INIT_NSTATS]);
    });
    if (rc) return rc;
    // the one exchange of the path: [0..10] SUM (count, cells, 9 sums), [11..19] MIN, [20..28] MAX (include/aerobulk_amd.h)
    for (int k = 0; k < AB_INIT_NSTATS; ++k) stats[k] = part[k];
    for (int r = 1; r < n; ++r) {
        const double *q = &part[(size_t)r * AB_INIT_NSTATS];
        for (int k = 0; k < 11; ++k) stats[k] += q[k];
        for (int k = 11; k < 20; ++k) stats[k] = q[k] < stats[k] ? q[k] : stats[k];
        for (int k = 20; k < 29; ++k) stats[k] = q[k] > stats[k] ? q[k] : stats[k];
    }
    return AB_OK;
}

int sharded_compute(ab_session *s, int jt, double zt, double zu, int niter, const void *const in[8], void *const out[6], int mem,
                    void *stream)
{
    if (mem == AB_MEM_DEVICE) {
        int rc = check_device_arrays(s, in[0]);
        if (rc) return rc;
    }
    int rc = for_shards(s, mem == AB_MEM_HOST, [&](int r) {
        const long j0 = s->shard_j0[r];
        return ab_session_compute(s->shards[r], jt, zt, zu, niter, off(s, in[0], j0), off(s, in[1], j0), off(s, in[2], j0),
                                  off(s, in[3], j0), off(s, in[4], j0), off(s, in[5], j0), off(s, in[6], j0), off(s, in[7], j0),
                                  offw(s, out[0], j0), offw(s, out[1], j0), offw(s, out[2], j0), offw(s, out[3], j0),
                                  offw(s, out[4], j0), offw(s, out[5], j0), mem, stream);
    });
    s->last_jt = jt;
    return rc;
}

int sharded_turb(ab_session *s, int kt, double zt, double zu, int use_cs, int use_wl, int nb_iter, const ab_turb_fields *f, int mem,
                 void *stream)
{
    if (mem == AB_MEM_DEVICE) {
        int rc = check_device_arrays(s, f->T_s);
        if (rc) return rc;
    }
    int rc = for_shards(s, mem == AB_MEM_HOST, [&](int r) {
        const long j0 = s->shard_j0[r];
        ab_turb_fields g;
        g.T_s = offw(s, f->T_s, j0); g.theta_zt = off(s, f->theta_zt, j0); g.q_s = offw(s, f->q_s, j0); g.q_zt = off(s, f->q_zt, j0);
        g.U_zu = off(s, f->U_zu, j0); g.Qsw = off(s, f->Qsw, j0); g.rad_lw = off(s, f->rad_lw, j0); g.slp = off(s, f->slp, j0);
        g.Cd = offw(s, f->Cd, j0); g.Ch = offw(s, f->Ch, j0); g.Ce = offw(s, f->Ce, j0); g.t_zu = offw(s, f->t_zu, j0);
        g.q_zu = offw(s, f->q_zu, j0); g.Ubzu = offw(s, f->Ubzu, j0);
        return ab_session_turb(s->shards[r], kt, zt, zu, use_cs, use_wl, nb_iter, &g, mem, stream);
    });
    s->last_jt = kt;
    return rc;
}

int sharded_check(ab_session *s)
{
    return for_shards(s, false, [&](int r) { return ab_session_check(s->shards[r]); });
}

int sharded_set_solar_time(ab_session *s, int isecday_utc, const void *lon, int mem, void *stream)
{
    if (lon && mem == AB_MEM_DEVICE) {
        int rc = check_device_arrays(s, lon);
        if (rc) return rc;
    }
    s->isecday = isecday_utc;
    return for_shards(s, false, [&](int r) {
        return ab_session_set_solar_time(s->shards[r], isecday_utc, off(s, lon, s->shard_j0[r]), mem, stream);
    });
}

int sharded_set_diagnostics(ab_session *s, const ab_diag *d, int mem)
{
    return for_shards(s, false, [&](int r) {
        if (!d) return ab_session_set_diagnostics(s->shards[r], nullptr, mem);
        const long j0 = s->shard_j0[r];
        ab_diag g;
        void *const src[16] = {d->Cd, d->Ch, d->Ce, d->t_zu, d->q_zu, d->Ubzu, d->CdN, d->ChN, d->CeN, d->z0, d->u_star, d->L,
                               d->UN10, d->dT_cs, d->dT_wl, d->Hz_wl};
        void **dst[16] = {&g.Cd, &g.Ch, &g.Ce, &g.t_zu, &g.q_zu, &g.Ubzu, &g.CdN, &g.ChN, &g.CeN, &g.z0, &g.u_star, &g.L,
                          &g.UN10, &g.dT_cs, &g.dT_wl, &g.Hz_wl};
        for (int i = 0; i < 16; ++i) *dst[i] = offw(s, src[i], j0);
        return ab_session_set_diagnostics(s->shards[r], &g, mem);
    });
}

int sharded_get_wl_state(ab_session *s, double *state4n)
{
    const size_t n = (size_t)s->n;
    return for_shards(s, false, [&](int r) {
        ab_session *c = s->shards[r];
        std::vector<double> tmp(4 * (size_t)c->n);
        int rc = ab_session_get_wl_state(c, tmp.data());
        if (rc) return rc;
        const size_t o = (size_t)s->shard_j0[r] * s->ni;
        for (int p = 0; p < 4; ++p) memcpy(state4n + p * n + o, tmp.data() + (size_t)p * c->n, sizeof(double) * (size_t)c->n);
        return (int)AB_OK;
    });
}

// shards on different devices run side by side (the slowest device counts), shards sharing a device one after the other
double sharded_last_kernel_ms(ab_session *s)
{
    double per_dev[64] = {0.};
    double worst = -1.;
    for (ab_session *c : s->shards) {
        const double ms = ab_session_last_kernel_ms(c);
        if (ms < 0.) return -1.;
        double &acc = per_dev[c->device & 63];
        acc += ms;
        worst = acc > worst ? acc : worst;
    }
    return worst;
}


// ------------------------------------------------------------------------------------------------
// AEROBULK_MODEL at jt == 1 (INIT + compute, mod_aerobulk.f90:246-262) through a sharded session: every shard runs its pipelined pass
// with AEROBULK_INIT's statistics riding on it (its own device, its own PCIe link, its own worker thread), starting from the humidity
// type ITS first chunk indicates; the statistics are combined here — the one exchange of the path — the decisions are taken on the
// whole domain, and a shard whose guess was wrong computes its rows again from the fields resident in its HBM (round 2: a statistics
// pass of their own and then the compute pass: 30.3 ms against 21.5 ms unsharded).
// staging buffers and the FIRST transfer of every shard's copy streams, one shard after the other (leaf_prepare_staging: why)
int sharded_prepare_staging(ab_session *s, int with_rad, int with_ts)
{
    return for_shards(s, false, [&](int r) { return leaf_prepare_staging(s->shards[r], with_rad, with_ts); });
}

int sharded_model_first_record(ab_session *s, double zt, double zu, int niter, const void *const in[8], void *const out[6], int have_rad,
                               ab_init_report *report)
{
    const int n = (int)s->shards.size();
    std::vector<double> part((size_t)n * AB_INIT_NSTATS);
    std::vector<int> guess(n, AB_HUM_SH);
    std::vector<FusedShard *> keep(n, nullptr);
    auto release = [&] { for (auto &k : keep) { leaf_fused_release(k); k = nullptr; } };
    int rc = sharded_prepare_staging(s, in[6] && in[7], out[5] != nullptr);
    if (rc) return rc;
    rc = for_shards(s, true, [&](int r) {
        const long j0 = s->shard_j0[r];
        const void *cin[8];
        void *cout[6];
        for (int i = 0; i < 8; ++i) cin[i] = off(s, in[i], j0);
        for (int i = 0; i < 6; ++i) cout[i] = offw(s, out[i], j0);
        return leaf_fused_first_record(s->shards[r], zt, zu, niter, cin, cout, have_rad, &part[(size_t)r * AB_INIT_NSTATS], &guess[r], &keep[r]);
    });
    if (rc) { release(); return rc; }
    double stats[AB_INIT_NSTATS];
    for (int k = 0; k < AB_INIT_NSTATS; ++k) stats[k] = part[k];
    for (int r = 1; r < n; ++r) {
        const double *q = &part[(size_t)r * AB_INIT_NSTATS];
        for (int k = 0; k < 11; ++k) stats[k] += q[k];
        for (int k = 11; k < 20; ++k) stats[k] = q[k] < stats[k] ? q[k] : stats[k];
        for (int k = 20; k < 29; ++k) stats[k] = q[k] > stats[k] ? q[k] : stats[k];
    }
    rc = ab_session_init_apply(s, stats, have_rad, report);      // sets hum_type on the parent and on every shard
    if (rc) { release(); return rc; }
    bool any = false;
    for (int r = 0; r < n; ++r) any = any || guess[r] != s->hum_type;
    if (any) {
        fprintf(stderr, "aerobulk_amd: AEROBULK_INIT: a shard's first 2^20 cells read as another humidity type than the whole domain (%d): "
                        "its rows of record 1 are computed again from the resident fields\n", s->hum_type);
        rc = for_shards(s, true, [&](int r) { return guess[r] != s->hum_type ? leaf_fused_redo(s->shards[r], keep[r]) : (int)AB_OK; });
    }
    release();
    s->last_jt = 1;
    if (rc) return rc;
    return sharded_check(s);
}

// ------------------------------------------------------------------------------------------------
// Device-resident fields, one set of pointers per shard: the kernels are enqueued on each shard's device, asynchronously
int sharded_compute_shards(ab_session *s, int jt, double zt, double zu, int niter, const ab_shard_arrays *sh, void *const *streams)
{
    int rc = for_shards(s, false, [&](int r) {
        const ab_shard_arrays &a = sh[r];
        return ab_session_compute(s->shards[r], jt, zt, zu, niter, a.sst, a.t_zt, a.hum_zt, a.u_zu, a.v_zu, a.slp, a.rad_sw, a.rad_lw, a.ql,
                                  a.qh, a.tau_x, a.tau_y, a.evap, a.t_s, AB_MEM_DEVICE, streams ? streams[r] : nullptr);
    });
    s->last_jt = jt;
    return rc;
}

// ------------------------------------------------------------------------------------------------
// The gather of the fluxes (include/aerobulk_amd.h: ab_session_gather).  RCCL inside the process, loaded on first use.
// Types, enumerators and prototypes come from RCCL's own header (so that ncclFloat64, ncclComm_t ... are what the library means by them);
// the library itself is not linked: a session on one device never needs it, and it is loaded when the first multi-device gather asks.
namespace {
struct Rccl {
    void *lib = nullptr;
    int version = 0;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    bool ok() const { return lib != nullptr; }
};
Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return;
        auto sym = [&](const char *n) { return dlsym(h, n); };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
        if (!(r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.GetVersion)) return;
        // the enumerators compiled in above are those of the header's major version: a library of another major is refused
        if (r.GetVersion(&r.version) != ncclSuccess || r.version / 10000 != NCCL_MAJOR) return;
        r.lib = h;
    });
    return r;
}
constexpr ncclDataType_t kNcclFloat = ncclFloat32, kNcclDouble = ncclFloat64;

// communicator over the DISTINCT devices of the session; rank_of_shard[r] = rank of shard r's device
struct GatherState {
    std::vector<int> devs;                 // rank -> device
    std::vector<int> rank_of_shard;
    std::vector<ncclComm_t> comm;          // per rank; empty when every shard lives on one device (no RCCL needed)
    std::vector<hipEvent_t> ev;            // per shard: "the shard's fluxes are ready" / "the local copy is done"
};
}  // namespace

void gather_release(ab_session *s)
{
    GatherState *g = static_cast<GatherState *>(s->gather);
    if (!g) return;
    for (ncclComm_t c : g->comm)
        if (c) (void)rccl().CommDestroy(c);
    for (size_t r = 0; r < g->ev.size(); ++r)
        if (g->ev[r]) {
            (void)hipSetDevice(s->shards[r]->device);
            (void)hipEventDestroy(g->ev[r]);
        }
    delete g;
    s->gather = nullptr;
}

int sharded_gather(ab_session *s, int root, const ab_shard_arrays *sh, const ab_flux_arrays *dst, void *const *streams, int synchronize)
{
    const int n = (int)s->shards.size();
    if (root < 0 || root >= n) return sfail(AB_ERR_ARG, "ab_session_gather: root shard %d of %d", root, n);
    ab::DeviceGuard dguard_;
    // AEROBULK_AMD_GATHER=rccl: every shard but the root's own goes through ncclSend / ncclRecv, also on the root's device (a rank
    // sending to itself): how the RCCL leg is exercised on a one-GPU box (tests/test_gpu_sharded.py)
    const char *ge = getenv("AEROBULK_AMD_GATHER");
    const bool force_rccl = ge && strcmp(ge, "rccl") == 0;
    auto local = [&](int r) { return r == root || (s->shards[r]->device == s->shards[root]->device && !force_rccl); };
    GatherState *g = static_cast<GatherState *>(s->gather);
    if (g && force_rccl && g->comm.empty()) {      // (created without RCCL earlier)
        gather_release(s);
        g = nullptr;
    }
    if (!g) {
        g = new GatherState;
        g->rank_of_shard.resize(n);
        for (int r = 0; r < n; ++r) {
            const int d = s->shards[r]->device;
            int k = 0;
            while (k < (int)g->devs.size() && g->devs[k] != d) ++k;
            if (k == (int)g->devs.size()) g->devs.push_back(d);
            g->rank_of_shard[r] = k;
        }
        g->ev.assign(n, nullptr);
        for (int r = 0; r < n; ++r) {
            if (hipSetDevice(s->shards[r]->device) != hipSuccess || hipEventCreateWithFlags(&g->ev[r], hipEventDisableTiming) != hipSuccess) {
                s->gather = g;
                gather_release(s);
                return sfail(AB_ERR_HIP, "ab_session_gather: cannot create events");
            }
        }
        if (g->devs.size() > 1 || force_rccl) {
            if (!rccl().ok()) {
                s->gather = g;
                gather_release(s);
                return sfail(AB_ERR_HIP, "ab_session_gather: librccl.so not found (the shards live on %zu devices)", g->devs.size());
            }
            g->comm.assign(g->devs.size(), nullptr);
            const int rc = rccl().CommInitAll(g->comm.data(), (int)g->devs.size(), g->devs.data());
            if (rc != 0) {
                g->comm.clear();
                s->gather = g;
                gather_release(s);
                return sfail(AB_ERR_HIP, "ab_session_gather: ncclCommInitAll over %zu devices: %s", g->devs.size(),
                             rccl().GetErrorString ? rccl().GetErrorString((ncclResult_t)rc) : "error");
            }
        }
        s->gather = g;
    }
    const int root_dev = s->shards[root]->device, root_rank = g->rank_of_shard[root];
    hipStream_t root_st = streams ? (hipStream_t)streams[root] : nullptr;
    void *d[6] = {dst->ql, dst->qh, dst->tau_x, dst->tau_y, dst->evap, dst->t_s};
    const ncclDataType_t dtype = s->esz == 4 ? kNcclFloat : kNcclDouble;
    auto src_of = [&](int r, int f) -> const void * {
        const void *p[6] = {sh[r].ql, sh[r].qh, sh[r].tau_x, sh[r].tau_y, sh[r].evap, sh[r].t_s};
        return p[f];
    };
    for (int r = 0; r < n; ++r)
        for (int f = 0; f < 6; ++f)
            if (d[f] && !src_of(r, f)) return sfail(AB_ERR_ARG, "ab_session_gather: field %d wanted but shard %d has no such output", f, r);
    // shards on the root's device: device-to-device copies on the root's stream, behind the shard's own stream
    for (int r = 0; r < n; ++r) {
        if (!local(r)) continue;
        hipStream_t st = streams ? (hipStream_t)streams[r] : nullptr;
        if (hipSetDevice(root_dev) != hipSuccess) return sfail(AB_ERR_HIP, "ab_session_gather: hipSetDevice(%d)", root_dev);
        if (st != root_st) {
            if (hipEventRecord(g->ev[r], st) != hipSuccess || hipStreamWaitEvent(root_st, g->ev[r], 0) != hipSuccess)
                return sfail(AB_ERR_HIP, "ab_session_gather: stream ordering failed");
        }
        const size_t cnt = (size_t)s->shard_njl[r] * s->ni, offb = (size_t)s->shard_j0[r] * s->ni * s->esz;
        for (int f = 0; f < 6; ++f) {
            if (!d[f]) continue;
            void *to = (char *)d[f] + offb;
            if (to == src_of(r, f)) continue;      // the shard computed straight into its place
            if (hipMemcpyAsync(to, src_of(r, f), cnt * s->esz, hipMemcpyDeviceToDevice, root_st) != hipSuccess)
                return sfail(AB_ERR_HIP, "ab_session_gather: device-to-device copy failed");
        }
    }
    // shards on other devices: one group of ncclSend (on the shard's stream) / ncclRecv (on the root's stream) per field and shard,
    // straight from the shard's array into its rows of dst
    if (!g->comm.empty()) {
        Rccl &R = rccl();
        int rc = R.GroupStart();
        for (int r = 0; r < n && rc == 0; ++r) {
            if (local(r)) continue;
            hipStream_t st = streams ? (hipStream_t)streams[r] : nullptr;
            const int rk = g->rank_of_shard[r];
            const size_t cnt = (size_t)s->shard_njl[r] * s->ni, offb = (size_t)s->shard_j0[r] * s->ni * s->esz;
            for (int f = 0; f < 6 && rc == 0; ++f) {
                if (!d[f]) continue;
                rc = R.Send(src_of(r, f), cnt, dtype, root_rank, g->comm[rk], st);
                if (rc == 0) rc = R.Recv((char *)d[f] + offb, cnt, dtype, rk, g->comm[root_rank], root_st);
            }
        }
        const int rc2 = R.GroupEnd();
        if (rc == 0) rc = rc2;
        if (rc != 0) {
            // the device-to-device copies of the local shards are already enqueued into dst: drain them, so that the caller does not
            // reuse or free the destination while they are in flight (round-3 advisory)
            if (hipSetDevice(root_dev) == hipSuccess) (void)hipStreamSynchronize(root_st);
            return sfail(AB_ERR_HIP, "ab_session_gather: RCCL: %s (dst holds the rows of the root's own device only)", R.GetErrorString ? R.GetErrorString((ncclResult_t)rc) : "error");
        }
    }
    if (synchronize) {
        if (hipSetDevice(root_dev) != hipSuccess || hipStreamSynchronize(root_st) != hipSuccess)
            return sfail(AB_ERR_HIP, "ab_session_gather: synchronisation failed");
        for (int r = 0; r < n; ++r) {
            if (local(r)) continue;
            if (hipSetDevice(s->shards[r]->device) != hipSuccess || hipStreamSynchronize(streams ? (hipStream_t)streams[r] : nullptr) != hipSuccess)
                return sfail(AB_ERR_HIP, "ab_session_gather: synchronisation failed");
        }
    }
    return AB_OK;
}

}  // namespace ab
