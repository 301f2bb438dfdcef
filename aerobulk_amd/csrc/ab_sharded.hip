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

#include <cstdarg>
#include <cstdio>
#include <cstring>
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

// Run fn(r) for every shard: concurrently (one thread per shard) when the call blocks on transfers, in order otherwise.
// Returns the first non-zero status in shard order, with that shard's error text.
template <class F> int for_shards(ab_session *s, bool concurrent, F fn)
{
    const int n = (int)s->shards.size();
    std::vector<int> rc(n, AB_OK);
    std::vector<std::string> msg(n);
    if (concurrent && n > 1) {
        std::vector<std::thread> th;
        for (int r = 0; r < n; ++r)
            th.emplace_back([&, r] {
                rc[r] = fn(r);
                if (rc[r]) msg[r] = ab_last_error();
            });
        for (auto &t : th) t.join();
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

int sharded_create(ab_session **out, int algo, long ni, long nj, int nt, int use_skin, int precision, const int *devices, int nshards)
{
    ab_session *p = new ab_session;
    p->algo = algo; p->ni = ni; p->nj = nj; p->n = ni * nj; p->nt = nt; p->use_skin = use_skin ? 1 : 0;
    p->f32 = (precision != AB_F64); p->compute64 = (precision == AB_F32_STORAGE) ? 1 : (precision == AB_F32_MIXED ? 2 : 0); p->esz = p->f32 ? 4 : 8; p->device = devices ? devices[0] : 0;
    // contiguous j-blocks, the first (nj mod nshards) one row taller
    long j0 = 0;
    for (int r = 0; r < nshards; ++r) {
        const long njl = nj / nshards + (r < nj % nshards ? 1 : 0);
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

}  // namespace ab
