// ab_runtime.hip — C-ABI runtime of the MI355X bulk-flux engine (include/aerobulk_amd.h).
//
// Owns what the reference keeps in Fortran module globals (mod_const.f90:22-33, warm-layer SAVE
// arrays mod_skin_coare.f90:31-36 / mod_skin_ecmwf.f90:52-55): here it is per-session state living
// in HBM.  There is NO CPU fallback: every entry point needs a gfx950 device and returns
// AB_ERR_HIP (with a message) when none is usable.
#include "../../include/aerobulk_amd.h"
#include "ab_kernels.hpp"
#include "ab_session.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

namespace ab {
struct FusedShard {
    FluxCall c;
    void *hout[6], *dout[6];
    size_t bytes;
};
}  // namespace ab

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define AB_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(AB_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Optional ROCTx ranges around the entry points that launch work, for `rocprofv3 --marker-trace` (the reference has no
// tracing).  Off unless AEROBULK_AMD_ROCTX=1; the marker library is looked up at run time, never linked.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *e = getenv("AEROBULK_AMD_ROCTX");
        if (!e || e[0] != '1') return;
        void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
struct Range {
    static Roctx &tx() { static Roctx r; return r; }
    bool on;
    explicit Range(const char *name) : on(tx().push != nullptr) { if (on) tx().push(name); }
    ~Range() { if (on) tx().pop(); }
};

}  // namespace
namespace ab {
void set_last_error(const std::string &msg) { g_err = msg; }
}  // namespace ab
namespace {

const char *kAlgoNames[6] = {"other", "coare3p0", "coare3p6", "ncar", "ecmwf", "andreas"};

bool algo_has_skin(int algo) { return algo == AB_ALGO_COARE3P0 || algo == AB_ALGO_COARE3P6 || algo == AB_ALGO_ECMWF; }

}  // namespace

extern "C" {

int ab_algo_from_string(const char *calgo, int len)
{
    if (!calgo) return 0;
    size_t l = len < 0 ? strlen(calgo) : (size_t)len;
    while (l > 0 && (calgo[l - 1] == ' ' || calgo[l - 1] == '\0')) --l;  // TRIM(calgo)
    for (int a = 1; a <= 5; ++a)
        if (strlen(kAlgoNames[a]) == l && strncmp(kAlgoNames[a], calgo, l) == 0) return a;
    return 0;
}

const char *ab_algo_name(int algo) { return (algo >= 0 && algo <= 5) ? kAlgoNames[algo] : "unknown"; }

const char *ab_strerror(int st)
{
    switch (st) {
    case AB_OK: return "ok";
    case AB_ERR_ALGO: return "unknown bulk algorithm";
    case AB_ERR_SKIN_ALGO: return "Only `COARE*` and `ECMWF` algorithms support cool-skin & warm/layer schemes";
    case AB_ERR_SKIN_NORAD: return "provide SW and LW rad. input if you want to use skin schemes";
    case AB_ERR_JT: return "jt < 1 !??";
    case AB_ERR_ALL_MASKED: return "the whole domain is masked! check unit consistency of input fields";
    case AB_ERR_HUM_TYPE: return "un-identified humidity type";
    case AB_ERR_UNITS: return "input field does not seem to be in the expected unit";
    case AB_ERR_TAU: return "wind stress too strong (> 10 N/m^2)";
    case AB_ERR_HIP: return "HIP runtime error / no usable gfx950 device";
    case AB_ERR_ARG: return "bad argument";
    case AB_ERR_STATE: return "call protocol violated";
    case AB_ERR_NOCONV: return "e_air fixed point not converged";
    default: return "unknown status";
    }
}

const char *ab_last_error(void) { return g_err.c_str(); }

int ab_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ab_session_create(ab_session **out, int algo, long ni, long nj, int nt, int use_skin, int precision, int device)
{
    if (!out) return fail(AB_ERR_ARG, "ab_session_create: out == NULL");
    *out = nullptr;
    if (algo < AB_ALGO_COARE3P0 || algo > AB_ALGO_ANDREAS)
        return fail(AB_ERR_ALGO, "bulk algorithm id %d is unknown!!!", algo);
    if (ni <= 0 || nj <= 0 || nt < 1) return fail(AB_ERR_ARG, "ab_session_create: bad shape %ld x %ld, nt=%d", ni, nj, nt);
    if (use_skin && !algo_has_skin(algo))
        return fail(AB_ERR_SKIN_ALGO, " AEROBULK_INIT => Only `COARE*` and `ECMWF` algorithms support cool-skin & warm/layer schemes");
    if (precision != AB_F64 && precision != AB_F32 && precision != AB_F32_STORAGE && precision != AB_F32_MIXED) return fail(AB_ERR_ARG, "bad precision %d", precision);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(AB_ERR_HIP, "no HIP device visible: this engine has no CPU fallback");
    if (device == AB_DEVICE_ALL) {   // row blocks over every visible device (one device: an ordinary session)
        if (ndev > 1 && nj >= ndev) return ab::sharded_create(out, algo, ni, nj, nt, use_skin, precision, nullptr, ndev);
        device = 0;
    }
    if (device < 0) AB_HIP(hipGetDevice(&device));
    if (device >= ndev) return fail(AB_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
    ab::DeviceGuard dguard_;      // the caller's current device is the caller's (round-2 advisory: torch allocated on the last shard's GPU afterwards)
    AB_HIP(hipSetDevice(device));
    ab_session *s = new ab_session;
    s->algo = algo; s->ni = ni; s->nj = nj; s->n = ni * nj; s->nt = nt; s->use_skin = use_skin ? 1 : 0;
    s->f32 = (precision != AB_F64); s->compute64 = (precision == AB_F32_STORAGE) ? 1 : (precision == AB_F32_MIXED ? 2 : 0); s->esz = s->f32 ? 4 : 8; s->device = device;
    if (const char *e = getenv("AEROBULK_AMD_REGROUP")) s->regroup = atoi(e) != 0;   // A/B measurements
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    chk(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    chk(hipStreamCreateWithFlags(&s->s_h2d, hipStreamNonBlocking));
    chk(hipStreamCreateWithFlags(&s->s_d2h, hipStreamNonBlocking));
    chk(hipEventCreate(&s->ev0));
    chk(hipEventCreate(&s->ev1));
    chk(hipMalloc((void **)&s->d_flags, 8 * sizeof(int)));   // [0] the error flag; [4..6] the tile counters of flux_kernel_cu (zero between launches: the kernel re-arms them)
    chk(hipMalloc((void **)&s->d_partials, sizeof(double) * ab::kStatBlocks * ab::kStatStride));
    if (e == hipSuccess) chk(hipMemset(s->d_flags, 0, 8 * sizeof(int)));
    if (s->use_skin && nt > 1) {  // persistent warm-layer planes; a single record keeps them in registers
        const int np = (algo == AB_ALGO_ECMWF) ? 2 : 4;
        for (int p = 0; p < np; ++p) chk(hipMalloc(&s->wl[p], s->esz * (size_t)s->n));
    }
    if (e != hipSuccess) {
        ab_session_destroy(s);
        return fail(AB_ERR_HIP, "ab_session_create: %s", hipGetErrorString(e));
    }
    *out = s;
    return AB_OK;
}

int ab_session_create_sharded(ab_session **out, int algo, long ni, long nj, int nt, int use_skin, int precision,
                              const int *devices, int nshards)
{
    if (!out) return fail(AB_ERR_ARG, "ab_session_create_sharded: out == NULL");
    *out = nullptr;
    if (nshards < 1 || !devices) return fail(AB_ERR_ARG, "ab_session_create_sharded: no device list");
    if (nshards > nj) return fail(AB_ERR_ARG, "ab_session_create_sharded: %d shards for %ld rows", nshards, nj);
    return ab::sharded_create(out, algo, ni, nj, nt, use_skin, precision, devices, nshards);
}

int ab_session_create_sharded_rows(ab_session **out, int algo, long ni, long nj, int nt, int use_skin, int precision,
                                   const int *devices, int nshards, const long *nj_per_shard)
{
    if (!out) return fail(AB_ERR_ARG, "ab_session_create_sharded_rows: out == NULL");
    *out = nullptr;
    if (nshards < 1 || !devices || !nj_per_shard) return fail(AB_ERR_ARG, "ab_session_create_sharded_rows: no device list / no row counts");
    long sum = 0;
    for (int r = 0; r < nshards; ++r) {
        if (nj_per_shard[r] < 1) return fail(AB_ERR_ARG, "ab_session_create_sharded_rows: shard %d has %ld rows", r, nj_per_shard[r]);
        sum += nj_per_shard[r];
    }
    if (sum != nj) return fail(AB_ERR_ARG, "ab_session_create_sharded_rows: the shards hold %ld rows, the grid has %ld", sum, nj);
    return ab::sharded_create(out, algo, ni, nj, nt, use_skin, precision, devices, nshards, nj_per_shard);
}

int ab_session_shard_count(const ab_session *s) { return s ? (s->sharded() ? (int)s->shards.size() : 1) : 0; }

int ab_session_shard_info(const ab_session *s, int shard, long *j0, long *nj_local, int *device)
{
    if (!s || shard < 0 || shard >= ab_session_shard_count(s)) return fail(AB_ERR_ARG, "ab_session_shard_info: bad shard %d", shard);
    const bool sh = s->sharded();
    if (j0) *j0 = sh ? s->shard_j0[shard] : 0;
    if (nj_local) *nj_local = sh ? s->shard_njl[shard] : s->nj;
    if (device) *device = sh ? s->shards[shard]->device : s->device;
    return AB_OK;
}

int ab_session_destroy(ab_session *s)
{
    if (!s) return AB_OK;
    if (s->sharded()) return ab::sharded_destroy(s);
    ab::DeviceGuard dguard_;
    (void)hipSetDevice(s->device);
    // device-mode calls run on the caller's stream and use session-owned buffers (WL state, flags, lon, diagnostics staging): wait
    // for the last of them before they are freed — through the session's own event, recorded behind that call: the caller's stream
    // may be gone by now (a model that tears its streams down first)
    if (s->done_pending && s->ev1) (void)hipEventSynchronize(s->ev1);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (auto &p : s->wl) if (p) (void)hipFree(p);
    for (auto &p : s->stage_in) if (p) (void)hipFree(p);
    for (auto &p : s->stage_out) if (p) (void)hipFree(p);
    for (auto &p : s->diag_dev) if (p) (void)hipFree(p);
    if (s->d_lon) (void)hipFree(s->d_lon);
    if (s->d_flags) (void)hipFree(s->d_flags);
    if (s->d_partials) (void)hipFree(s->d_partials);
    if (s->d_fused) (void)hipFree(s->d_fused);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->s_h2d) (void)hipStreamDestroy(s->s_h2d);
    if (s->s_d2h) (void)hipStreamDestroy(s->s_d2h);
    delete s;
    return AB_OK;
}

}  // extern "C"

namespace {

// copy caller host arrays into the session's device staging (allocated on first use)
int stage_inputs(ab_session *s, const void *const host[8], const void *dev[8])
{
    const size_t bytes = s->esz * (size_t)s->n;
    for (int i = 0; i < 8; ++i) {
        dev[i] = nullptr;
        if (!host[i]) continue;
        if (!s->stage_in[i]) AB_HIP(hipMalloc(&s->stage_in[i], bytes));
        AB_HIP(hipMemcpyAsync(s->stage_in[i], host[i], bytes, hipMemcpyHostToDevice, s->stream));
        dev[i] = s->stage_in[i];
    }
    return AB_OK;
}

// fold per-block partial rows of init_stats_kernel into [count, n_cells | 9 sums | 9 mins | 9 maxs] (AB_INIT_NSTATS layout)
void fold_partials(const double *part, int nrows, long ncells, double stats[AB_INIT_NSTATS])
{
    double *sum = stats + 2, *mn = stats + 2 + ab::kStatFields, *mx = stats + 2 + 2 * ab::kStatFields;
    stats[0] = 0.;
    stats[1] = (double)ncells;
    for (int f = 0; f < ab::kStatFields; ++f) { sum[f] = 0.; mn[f] = 1.e300; mx[f] = -1.e300; }
    for (int b = 0; b < nrows; ++b) {
        const double *p = &part[(size_t)b * ab::kStatStride];
        stats[0] += p[0];
        for (int f = 0; f < ab::kStatFields; ++f) {
            sum[f] += p[1 + 3 * f];
            if (p[2 + 3 * f] < mn[f]) mn[f] = p[2 + 3 * f];
            if (p[3 + 3 * f] > mx[f]) mx[f] = p[3 + 3 * f];
        }
    }
}

// type_of_humidity, mod_phymbl.f90:1984-2003; -1: not identifiable
int classify_humidity(double hmean, double hmin, double hmax)
{
    if ((hmean >= 0.) && (hmean < 0.08) && (hmin >= 0.) && (hmax < 0.08)) return AB_HUM_SH;
    if ((hmean >= 150.) && (hmean < 330.) && (hmin >= 150.) && (hmax < 330.)) return AB_HUM_DP;
    if ((hmean >= 0.) && (hmean <= 100.) && (hmin >= 0.) && (hmax <= 100.)) return AB_HUM_RH;
    return -1;
}

// AEROBULK_MODEL at jt == 1 on a large host grid: AEROBULK_INIT's statistics ride on the chunks of the pipelined pass below
// (one PCIe crossing of the inputs, overlapped with the return of the outputs).  The kernels need the humidity type, which is
// a GLOBAL decision (mod_aerobulk.f90:128-140) only known after the last chunk: they run with the type the FIRST chunk
// indicates, and the record is recomputed from the resident inputs in the (pathological) case the whole domain says otherwise.
struct FusedInit {
    double *d_part = nullptr;     // nch x kFusedBlocks partial rows
    int guess = AB_HUM_SH;
    int have_rad = 0;
    ab_init_report *report = nullptr;
    // a shard of a sharded session (ab::sharded_model_first_record): the verdict is GLOBAL, so the shard hands its statistics back
    // instead of applying them, and keeps what a redo with another humidity type needs
    bool defer = false;
    double stats[AB_INIT_NSTATS] = {};
    ab::FusedShard *keep = nullptr;
};
constexpr int kFusedBlocks = 128;

// Host calling convention for large grids: the record is cut into cell chunks and pipelined over three streams —
// H2D of chunk c+1 (this thread), kernel of chunk c, D2H of chunk c-1 (helper thread) — so that both PCIe directions
// and the kernel overlap.  `c` holds DEVICE staging pointers for the whole record; host_in/host_out are the caller's arrays.
constexpr long kPipeChunk = 1L << 20;        // cells per chunk (8 MiB per fp64 field)
constexpr long kPipeThreshold = 4L << 20;    // below this the plain path is as fast

hipError_t compute_host_pipelined(ab_session *s, const ab::FluxCall &c, const void *const host_in[8], void *const host_out[6],
                                  FusedInit *fi = nullptr)
{
    const long n = s->n;
    const int nch = (int)((n + kPipeChunk - 1) / kPipeChunk);
    const size_t esz = s->esz;
    const void *din[8] = {c.sst, c.t_zt, c.hum, c.u, c.v, c.slp, c.rad_sw, c.rad_lw};
    void *dout[6] = {c.ql, c.qh, c.tau_x, c.tau_y, c.evap, c.t_s};
    std::vector<hipEvent_t> kdone(nch, nullptr), h2d(nch, nullptr);
    hipError_t err = hipSuccess;
    for (int i = 0; i < nch && err == hipSuccess; ++i) {
        err = hipEventCreateWithFlags(&kdone[i], hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&h2d[i], hipEventDisableTiming);
    }
    std::atomic<int> launched{0};
    std::atomic<int> abort_flag{0};
    hipError_t err_d2h = hipSuccess;
    std::thread drain([&] {
        if (hipSetDevice(s->device) != hipSuccess) { err_d2h = hipErrorInvalidDevice; return; }
        for (int i = 0; i < nch; ++i) {
            while (launched.load(std::memory_order_acquire) <= i) {
                if (abort_flag.load()) return;
                std::this_thread::yield();
            }
            hipError_t e = hipStreamWaitEvent(s->s_d2h, kdone[i], 0);
            const long off = (long)i * kPipeChunk, cnt = (n - off < kPipeChunk) ? n - off : kPipeChunk;
            for (int f = 0; f < 6 && e == hipSuccess; ++f)
                if (host_out[f])
                    e = hipMemcpyAsync((char *)host_out[f] + off * esz, (const char *)dout[f] + off * esz, cnt * esz,
                                       hipMemcpyDeviceToHost, s->s_d2h);
            if (e != hipSuccess) { err_d2h = e; return; }
        }
        err_d2h = hipStreamSynchronize(s->s_d2h);
    });
    if (err == hipSuccess) err = hipEventRecord(s->ev0, s->stream);
    for (int i = 0; i < nch && err == hipSuccess; ++i) {
        const long off = (long)i * kPipeChunk, cnt = (n - off < kPipeChunk) ? n - off : kPipeChunk;
        for (int f = 0; f < 8 && err == hipSuccess; ++f)
            if (host_in[f])
                err = hipMemcpyAsync((char *)din[f] + off * esz, (const char *)host_in[f] + off * esz, cnt * esz,
                                     hipMemcpyHostToDevice, s->s_h2d);
        if (err == hipSuccess) err = hipEventRecord(h2d[i], s->s_h2d);
        if (err == hipSuccess) err = hipStreamWaitEvent(s->stream, h2d[i], 0);
        if (err != hipSuccess) break;
        if (fi) {   // AEROBULK_INIT statistics of this chunk (the reference checks rad_lw in both radiation slots, mod_aerobulk.f90:248)
            auto at = [&](const void *p) -> const void * { return p ? (const char *)p + off * esz : nullptr; };
            double *rows = fi->d_part + (size_t)i * kFusedBlocks * ab::kStatStride;
            err = ab::launch_init_stats(at(c.sst), at(c.t_zt), at(c.hum), at(c.u), at(c.v), at(c.slp), at(c.rad_lw), at(c.rad_lw), cnt,
                                        s->f32, rows, s->stream, kFusedBlocks);
            if (err == hipSuccess && i == 0) {   // the first chunk's verdict on the humidity type is the working assumption
                std::vector<double> part((size_t)kFusedBlocks * ab::kStatStride);
                err = hipMemcpyAsync(part.data(), rows, part.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream);
                if (err == hipSuccess) err = hipStreamSynchronize(s->stream);
                double st[AB_INIT_NSTATS];
                fold_partials(part.data(), kFusedBlocks, cnt, st);
                const int g = st[0] > 0. ? classify_humidity(st[2 + 6] / st[0], st[2 + ab::kStatFields + 6], st[2 + 2 * ab::kStatFields + 6]) : -1;
                fi->guess = g >= 0 ? g : AB_HUM_SH;
            }
            if (err != hipSuccess) break;
        }
        ab::FluxCall cc = c;
        if (fi) cc.hum_type = fi->guess;
        auto adv = [&](const void *p) -> const void * { return p ? (const char *)p + off * esz : nullptr; };
        auto advw = [&](void *p) -> void * { return p ? (char *)p + off * esz : nullptr; };
        cc.sst = adv(c.sst); cc.t_zt = adv(c.t_zt); cc.hum = adv(c.hum); cc.u = adv(c.u); cc.v = adv(c.v); cc.slp = adv(c.slp);
        cc.rad_sw = adv(c.rad_sw); cc.rad_lw = adv(c.rad_lw); cc.lon = adv(c.lon);
        cc.ql = advw(c.ql); cc.qh = advw(c.qh); cc.tau_x = advw(c.tau_x); cc.tau_y = advw(c.tau_y);
        cc.evap = advw(c.evap); cc.t_s = advw(c.t_s);
        for (int p = 0; p < 4; ++p) cc.wl[p] = advw(c.wl[p]);
        cc.n = cnt;
        err = ab::launch_flux(cc, s->stream);
        if (err == hipSuccess) err = hipEventRecord(kdone[i], s->stream);
        if (err == hipSuccess) launched.store(i + 1, std::memory_order_release);
    }
    if (err == hipSuccess) err = hipEventRecord(s->ev1, s->stream);
    if (err != hipSuccess) abort_flag.store(1);
    drain.join();
    if (err == hipSuccess) err = hipStreamSynchronize(s->stream);
    for (int i = 0; i < nch; ++i) {
        if (kdone[i]) (void)hipEventDestroy(kdone[i]);
        if (h2d[i]) (void)hipEventDestroy(h2d[i]);
    }
    return err != hipSuccess ? err : err_d2h;
}

}  // namespace

extern "C" {

int ab_session_set_humidity(ab_session *s, int hum_type)
{
    if (!s) return fail(AB_ERR_ARG, "NULL session");
    if (hum_type < AB_HUM_SH || hum_type > AB_HUM_RH) return fail(AB_ERR_HUM_TYPE, "humidty type %d is unknown!!!", hum_type);
    s->hum_type = hum_type;
    for (ab_session *c : s->shards) c->hum_type = hum_type;
    return AB_OK;
}

int ab_session_set_diagnostics(ab_session *s, const ab_diag *d, int mem)
{
    if (!s) return fail(AB_ERR_ARG, "NULL session");
    if (s->sharded()) return ab::sharded_set_diagnostics(s, d, mem);
    s->diag_on = false;
    for (auto &p : s->diag_user) p = nullptr;
    if (!d) return AB_OK;
    void *const ptrs[16] = {d->Cd, d->Ch, d->Ce, d->t_zu, d->q_zu, d->Ubzu, d->CdN, d->ChN, d->CeN, d->z0, d->u_star, d->L,
                            d->UN10, d->dT_cs, d->dT_wl, d->Hz_wl};
    for (int i = 0; i < 16; ++i) {
        s->diag_user[i] = ptrs[i];
        s->diag_on = s->diag_on || (ptrs[i] != nullptr);
    }
    s->diag_mem = mem;
    return AB_OK;
}

int ab_session_set_regroup(ab_session *s, int on)
{
    if (!s) return fail(AB_ERR_ARG, "NULL session");
    s->regroup = on ? 1 : 0;
    for (ab_session *c : s->shards) c->regroup = s->regroup;
    return AB_OK;
}

int ab_session_set_solar_time(ab_session *s, int isecday_utc, const void *lon, int mem, void *stream)
{
    if (!s) return fail(AB_ERR_ARG, "NULL session");
    if (s->sharded()) return ab::sharded_set_solar_time(s, isecday_utc, lon, mem, stream);
    ab::DeviceGuard dguard_;
    AB_HIP(hipSetDevice(s->device));
    s->isecday = isecday_utc;
    if (!lon) {
        if (s->d_lon) { (void)hipFree(s->d_lon); s->d_lon = nullptr; }
        return AB_OK;
    }
    const size_t bytes = s->esz * (size_t)s->n;
    if (!s->d_lon) AB_HIP(hipMalloc(&s->d_lon, bytes));
    // a device `lon` may still be being written on the caller's stream: the copy is ordered behind it, then drained (the
    // session keeps its own copy, so the caller may reuse the array at once)
    hipStream_t st = (mem == AB_MEM_DEVICE) ? (hipStream_t)stream : s->stream;
    AB_HIP(hipMemcpyAsync(s->d_lon, lon, bytes, mem == AB_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    AB_HIP(hipStreamSynchronize(st));
    return AB_OK;
}

int ab_session_init_stats(ab_session *s, const void *sst, const void *t_zt, const void *hum_zt, const void *u_zu,
                          const void *v_zu, const void *slp, const void *rad_sw, const void *rad_lw, int mem,
                          void *stream, double stats[AB_INIT_NSTATS])
{
    Range trace_("ab_session_init_stats");
    if (!s || !stats) return fail(AB_ERR_ARG, "NULL argument");
    if (!sst || !t_zt || !hum_zt || !u_zu || !v_zu || !slp) return fail(AB_ERR_ARG, "ab_session_init: NULL input field");
    const void *host[8] = {sst, t_zt, hum_zt, u_zu, v_zu, slp, rad_sw, rad_lw};
    if (s->sharded()) return ab::sharded_init_stats(s, host, mem, stream, stats);
    ab::DeviceGuard dguard_;
    AB_HIP(hipSetDevice(s->device));
    // device arrays: the reduction runs on the CALLER's stream, behind whatever is still producing the fields there (the
    // session's own stream is non-blocking and ordered with nothing)
    hipStream_t st = (mem == AB_MEM_DEVICE) ? (hipStream_t)stream : s->stream;
    for (auto &p : s->staged_from) p = nullptr;
    const void *dev[8];
    if (mem == AB_MEM_HOST) {
        // rad_sw == rad_lw is the reference's own call pattern (mod_aerobulk.f90:248): stage once
        if (rad_sw && rad_sw == rad_lw) host[6] = nullptr;
        int rc = stage_inputs(s, host, dev);
        if (rc) return rc;
        if (rad_sw && rad_sw == rad_lw) dev[6] = dev[7];
        for (int i = 0; i < 8; ++i) s->staged_from[i] = host[i];   // what each staging buffer now holds (ab_model reuses it)
    } else {
        for (int i = 0; i < 8; ++i) dev[i] = host[i];
    }
    const bool rad = dev[6] && dev[7];
    AB_HIP(ab::launch_init_stats(dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], rad ? dev[6] : nullptr,
                                 rad ? dev[7] : nullptr, s->n, s->f32, s->d_partials, st));
    std::vector<double> part((size_t)ab::kStatBlocks * ab::kStatStride);
    AB_HIP(hipMemcpyAsync(part.data(), s->d_partials, part.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    AB_HIP(hipStreamSynchronize(st));

    fold_partials(part.data(), ab::kStatBlocks, s->n, stats);
    return AB_OK;
}

int ab_session_init_apply(ab_session *s, const double stats[AB_INIT_NSTATS], int have_rad, ab_init_report *report)
{
    if (!s || !stats) return fail(AB_ERR_ARG, "NULL argument");
    const double cnt = stats[0];
    const double *sum = stats + 2, *mn = stats + 2 + ab::kStatFields, *mx = stats + 2 + 2 * ab::kStatFields;
    ab_init_report rep;
    memset(&rep, 0, sizeof rep);
    rep.n_cells = (long)stats[1];
    rep.n_masked = (long)(stats[1] - cnt);
    rep.hum_type = -1;
    rep.bad_field = -1;
    if (report) *report = rep;
    if (cnt <= 0.)  // mod_aerobulk.f90:121-123
        return fail(AB_ERR_ALL_MASKED, "the whole domain is masked!\n check unit consistency of input fields");

    // type_of_humidity, mod_phymbl.f90:1984-2003
    const double hmean = sum[6] / cnt, hmin = mn[6], hmax = mx[6];
    double hlo, hhi;
    rep.hum_type = classify_humidity(hmean, hmin, hmax);
    if (rep.hum_type == AB_HUM_SH) { hlo = 0.; hhi = 0.08; }
    else if (rep.hum_type == AB_HUM_DP) { hlo = 150.; hhi = 330.; }
    else if (rep.hum_type == AB_HUM_RH) { hlo = 0.; hhi = 100.; }
    else {
        if (report) *report = rep;
        return fail(AB_ERR_HUM_TYPE,
                    "ERROR: type_of_humidity()@mod_aerobulk_compute => un-identified humidity type!\n"
                    "   ==> we could not identify the humidity type based on the mean, min & max of the field:\n"
                    "     * mean = %g\n     * min  = %g\n     * max  = %g", hmean, hmin, hmax);
    }
    s->hum_type = rep.hum_type;
    for (ab_session *c : s->shards) c->hum_type = rep.hum_type;

    // check_unit_consistency x7(+2), mod_aerobulk.f90:143-153 ; ranges mod_phymbl.f90:1885-1940
    static const char *names[9] = {"sst", "t_air", "slp", "u10", "v10", "wnd", "hum", "rad_sw", "rad_lw"};
    static const char *units[9] = {"K", "K", "Pa", "m/s", "m/s", "m/s", "kg/kg", "W/m^2", "W/m^2"};
    const double lo[9] = {270., 180., 80000., -50., -50., 0., hlo, 0., 0.};
    const double hi[9] = {320., 330., 110000., 50., 50., 50., hhi, 1500., 750.};
    const int nf = have_rad ? 9 : 7;
    for (int f = 0; f < nf; ++f) {
        const double mean = sum[f] / cnt;
        if ((mx[f] > hi[f]) || (mn[f] < lo[f]) || (mean < lo[f]) || (mean > hi[f])) {
            rep.bad_field = f; rep.bad_min = mn[f]; rep.bad_max = mx[f]; rep.bad_mean = mean;
            if (report) *report = rep;
            return fail(AB_ERR_UNITS,
                        " *** ERROR (check_unit_consistency@mod_phymbl): field `%s` does not seem to be in %s !\n"
                        " min value = %10.3e max value = %10.3e mean value = %10.3e",
                        names[f], units[f], mn[f], mx[f], mean);
        }
    }
    if (report) *report = rep;
    return AB_OK;
}

int ab_session_init(ab_session *s, const void *sst, const void *t_zt, const void *hum_zt, const void *u_zu,
                    const void *v_zu, const void *slp, const void *rad_sw, const void *rad_lw, int mem,
                    void *stream, ab_init_report *report)
{
    double stats[AB_INIT_NSTATS];
    int rc = ab_session_init_stats(s, sst, t_zt, hum_zt, u_zu, v_zu, slp, rad_sw, rad_lw, mem, stream, stats);
    if (rc) return rc;
    return ab_session_init_apply(s, stats, (rad_sw && rad_lw) ? 1 : 0, report);
}

}  // extern "C"

// aerobulk_compute for one record; `fi` (AEROBULK_MODEL at jt == 1, large host grid, one device): AEROBULK_INIT's checks are
// taken along in the same pipelined pass instead of a pass of their own
static int compute_impl(ab_session *s, int jt, double zt, double zu, int niter, const void *sst, const void *t_zt,
                        const void *hum_zt, const void *u_zu, const void *v_zu, const void *slp, const void *rad_sw,
                        const void *rad_lw, void *ql, void *qh, void *tau_x, void *tau_y, void *evap, void *t_s,
                        int mem, void *stream, FusedInit *fi)
{
    Range trace_("ab_session_compute");
    if (!s) return fail(AB_ERR_ARG, "NULL session");
    if (jt < 1) return fail(AB_ERR_JT, "AEROBULK_MODEL => jt < 1 !??\n we are in a Fortran world here...");
    if (jt > s->nt) return fail(AB_ERR_STATE, "jt=%d > nt=%d", jt, s->nt);
    if (!sst || !t_zt || !hum_zt || !u_zu || !v_zu || !slp) return fail(AB_ERR_ARG, "ab_session_compute: NULL input field");
    if (!ql || !qh || !tau_x || !tau_y) return fail(AB_ERR_ARG, "ab_session_compute: NULL output field");
    if (s->use_skin && (!rad_sw || !rad_lw))
        return fail(AB_ERR_SKIN_NORAD, " AEROBULK_INIT => provide SW and LW rad. input if you want to use skin schemes");
    if (niter < 0) return fail(AB_ERR_ARG, "niter < 0");
    if (s->sharded()) {
        const void *in8[8] = {sst, t_zt, hum_zt, u_zu, v_zu, slp, rad_sw, rad_lw};
        void *out6[6] = {ql, qh, tau_x, tau_y, evap, t_s};
        return ab::sharded_compute(s, jt, zt, zu, niter, in8, out6, mem, stream);
    }
    if (s->use_skin && s->nt > 1 && jt > 1 && s->last_jt != jt - 1 && s->last_jt != jt)
        return fail(AB_ERR_STATE, "warm-layer state: record jt=%d requested after jt=%d", jt, s->last_jt);
    ab::DeviceGuard dguard_;
    AB_HIP(hipSetDevice(s->device));

    hipStream_t st = (mem == AB_MEM_HOST) ? s->stream : (hipStream_t)stream;
    const void *host_in[8] = {sst, t_zt, hum_zt, u_zu, v_zu, slp, s->use_skin ? rad_sw : nullptr,
                              s->use_skin ? rad_lw : nullptr};
    const void *din[8];
    void *hout[6] = {ql, qh, tau_x, tau_y, evap, t_s};
    if (mem != AB_MEM_HOST) s->reuse_staged = false;
    void *dout[6];
    const size_t bytes = s->esz * (size_t)s->n;
    const bool pipelined = (mem == AB_MEM_HOST) && (s->n >= kPipeThreshold) && !s->diag_on;
    if (mem == AB_MEM_HOST) {
        // AEROBULK_MODEL at jt == 1: the fields AEROBULK_INIT has just staged (ab_model marks them) do not cross PCIe again
        const void *copy_in[8];
        for (int i = 0; i < 8; ++i) {
            const bool resident = s->reuse_staged && host_in[i] && s->stage_in[i] && s->staged_from[i] == host_in[i];
            copy_in[i] = resident ? nullptr : host_in[i];
        }
        s->reuse_staged = false;
        for (auto &p : s->staged_from) p = nullptr;
        for (int i = 0; i < 8; ++i) {
            din[i] = nullptr;
            if (!host_in[i]) continue;
            if (!s->stage_in[i]) AB_HIP(hipMalloc(&s->stage_in[i], bytes));
            din[i] = s->stage_in[i];
            if (!pipelined && copy_in[i])   // the pipelined path issues its copies chunk by chunk below
                AB_HIP(hipMemcpyAsync(s->stage_in[i], copy_in[i], bytes, hipMemcpyHostToDevice, s->stream));
        }
        for (int i = 0; i < 8; ++i) host_in[i] = copy_in[i];   // from here on: the fields that still have to be copied
        for (int i = 0; i < 6; ++i) {
            dout[i] = nullptr;
            if (!hout[i]) continue;
            if (!s->stage_out[i]) AB_HIP(hipMalloc(&s->stage_out[i], bytes));
            dout[i] = s->stage_out[i];
        }
    } else {
        for (int i = 0; i < 8; ++i) din[i] = host_in[i];
        for (int i = 0; i < 6; ++i) dout[i] = hout[i];
    }

    ab::FluxCall c;
    memset(&c, 0, sizeof c);
    c.sst = din[0]; c.t_zt = din[1]; c.hum = din[2]; c.u = din[3]; c.v = din[4]; c.slp = din[5];
    c.rad_sw = din[6]; c.rad_lw = din[7]; c.lon = s->d_lon;
    c.ql = dout[0]; c.qh = dout[1]; c.tau_x = dout[2]; c.tau_y = dout[3]; c.evap = dout[4]; c.t_s = dout[5];
    for (int p = 0; p < 4; ++p) c.wl[p] = s->wl[p];
    if (s->diag_on) {
        for (int i = 0; i < 16; ++i) {
            if (!s->diag_user[i]) continue;
            if (s->diag_mem == AB_MEM_DEVICE) { c.diag[i] = s->diag_user[i]; continue; }
            if (!s->diag_dev[i]) AB_HIP(hipMalloc(&s->diag_dev[i], bytes));
            c.diag[i] = s->diag_dev[i];
        }
    }
    c.flags = s->d_flags;
    c.n = s->n; c.zt = zt; c.zu = zu;
    c.algo = s->algo; c.skin = s->use_skin; c.f32 = s->f32; c.compute64 = s->compute64;
    c.nb_iter = niter; c.hum_type = s->hum_type;
    c.wl_load = (s->use_skin && jt > 1 && s->wl[0]) ? 1 : 0;      // kt == nit000 initialises, mod_blk_coare3p6.f90:250
    c.wl_store = (s->use_skin && jt < s->nt && s->wl[0]) ? 1 : 0;  // freed at kt == nitend, :411
    c.isecday = s->isecday;
    c.regroup = s->regroup;

    if (pipelined) {
        const int nch = (int)((s->n + kPipeChunk - 1) / kPipeChunk);
        if (fi) {   // rows for the statistics: a session-owned buffer (allocated once; round 2 allocated and freed one per first record)
            const size_t rows = (size_t)nch * kFusedBlocks;
            if (s->d_fused_rows < rows) {
                if (s->d_fused) AB_HIP(hipFree(s->d_fused));
                s->d_fused = nullptr; s->d_fused_rows = 0;
                AB_HIP(hipMalloc((void **)&s->d_fused, sizeof(double) * rows * ab::kStatStride));
                s->d_fused_rows = rows;
            }
            fi->d_part = s->d_fused;
        }
        hipError_t e = compute_host_pipelined(s, c, host_in, hout, fi);
        s->timed = true;
        s->last_stream = st;
        s->last_jt = jt;
        if (fi) {
            // the global verdict: fold every chunk's rows, take AEROBULK_INIT's decisions (mod_aerobulk.f90:105-153)
            std::vector<double> part((size_t)nch * kFusedBlocks * ab::kStatStride);
            // (on the session's stream: a synchronous hipMemcpy goes through the device's NULL stream — from the worker threads of a
            // sharded session that left every later record 35 % slower, profiles/r3_host_path.txt)
            if (e == hipSuccess) e = hipMemcpyAsync(part.data(), fi->d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
            fi->d_part = nullptr;
            AB_HIP(e);
            double stats[AB_INIT_NSTATS];
            fold_partials(part.data(), nch * kFusedBlocks, s->n, stats);
            if (fi->defer) {   // a shard: the verdict is taken on the statistics of ALL shards (ab::sharded_model_first_record)
                memcpy(fi->stats, stats, sizeof stats);
                fi->keep = new ab::FusedShard;
                fi->keep->c = c;
                for (int i = 0; i < 6; ++i) { fi->keep->hout[i] = hout[i]; fi->keep->dout[i] = dout[i]; }
                fi->keep->bytes = bytes;
                return AB_OK;
            }
            int rc = ab_session_init_apply(s, stats, fi->have_rad, fi->report);
            if (rc) return rc;
            if (s->hum_type != fi->guess) {   // the first chunk misjudged the humidity type: redo the record from the resident inputs
                fprintf(stderr, "aerobulk_amd: AEROBULK_INIT: the first 2^20 cells read as humidity type %d, the whole domain as %d: "
                                "record 1 is computed again from the resident fields\n", fi->guess, s->hum_type);
                c.hum_type = s->hum_type;
                AB_HIP(hipMemsetAsync(s->d_flags, 0, sizeof(int), s->stream));   // whatever the misjudged pass flagged is void
                AB_HIP(ab::launch_flux(c, s->stream));
                for (int i = 0; i < 6; ++i)
                    if (hout[i]) AB_HIP(hipMemcpyAsync(hout[i], dout[i], bytes, hipMemcpyDeviceToHost, s->stream));
                AB_HIP(hipStreamSynchronize(s->stream));
            }
        } else {
            AB_HIP(e);
        }
        return ab_session_check(s);
    }
    AB_HIP(hipEventRecord(s->ev0, st));
    AB_HIP(ab::launch_flux(c, st));
    AB_HIP(hipEventRecord(s->ev1, st));
    s->timed = true;
    s->last_stream = st;
    s->last_jt = jt;

    if (s->diag_on && s->diag_mem == AB_MEM_HOST) {
        for (int i = 0; i < 16; ++i)
            if (s->diag_user[i]) AB_HIP(hipMemcpyAsync(s->diag_user[i], s->diag_dev[i], bytes, hipMemcpyDeviceToHost, st));
        if (mem != AB_MEM_HOST) AB_HIP(hipStreamSynchronize(st));
    }
    if (mem == AB_MEM_HOST) {
        for (int i = 0; i < 6; ++i)
            if (hout[i]) AB_HIP(hipMemcpyAsync(hout[i], dout[i], bytes, hipMemcpyDeviceToHost, st));
        return ab_session_check(s);
    }
    s->done_pending = true;      // ev1, recorded behind the kernel, is the last thing this call put on the caller's stream
    return AB_OK;
}

extern "C" {

int ab_session_compute(ab_session *s, int jt, double zt, double zu, int niter, const void *sst, const void *t_zt,
                       const void *hum_zt, const void *u_zu, const void *v_zu, const void *slp, const void *rad_sw,
                       const void *rad_lw, void *ql, void *qh, void *tau_x, void *tau_y, void *evap, void *t_s,
                       int mem, void *stream)
{
    return compute_impl(s, jt, zt, zu, niter, sst, t_zt, hum_zt, u_zu, v_zu, slp, rad_sw, rad_lw, ql, qh, tau_x, tau_y, evap, t_s, mem,
                        stream, nullptr);
}

int ab_session_turb(ab_session *s, int kt, double zt, double zu, int use_cs, int use_wl, int nb_iter,
                    const ab_turb_fields *f, int mem, void *stream)
{
    Range trace_("ab_session_turb");
    if (!s || !f) return fail(AB_ERR_ARG, "ab_session_turb: NULL argument");
    if (kt < 1) return fail(AB_ERR_JT, "TURB_%s => kt < 1 !??", ab_algo_name(s->algo));
    if (nb_iter < 0) return fail(AB_ERR_ARG, "nb_iter < 0");
    if (!f->T_s || !f->theta_zt || !f->q_s || !f->q_zt || !f->U_zu) return fail(AB_ERR_ARG, "ab_session_turb: NULL input field");
    if (!f->Cd || !f->Ch || !f->Ce || !f->t_zu || !f->q_zu || !f->Ubzu) return fail(AB_ERR_ARG, "ab_session_turb: NULL output field");
    const int skin = (use_cs ? 1 : 0) | (use_wl ? 2 : 0);
    if (skin && !algo_has_skin(s->algo))
        return fail(AB_ERR_SKIN_ALGO, "TURB_%s has no cool-skin / warm-layer scheme", ab_algo_name(s->algo));
    if (skin && (!f->Qsw || !f->rad_lw || !f->slp))   // mod_blk_coare3p6.f90:263-269
        return fail(AB_ERR_SKIN_NORAD, "you need to provide Qsw, rad_lw & slp to use cool-skin / warm-layer param!");
    if (s->compute64) return fail(AB_ERR_ARG, "AB_F32_STORAGE / AB_F32_MIXED sessions serve aerobulk_compute only (ab_session_compute)");
    if (s->sharded()) return ab::sharded_turb(s, kt, zt, zu, use_cs, use_wl, nb_iter, f, mem, stream);
    ab::DeviceGuard dguard_;
    AB_HIP(hipSetDevice(s->device));
    const size_t bytes = s->esz * (size_t)s->n;
    if (use_wl) {
        const int np = (s->algo == AB_ALGO_ECMWF) ? 2 : 4;
        if (kt > 1 && !s->wl[0]) return fail(AB_ERR_STATE, "warm-layer state: kt=%d requested before kt=1", kt);
        for (int p = 0; p < np; ++p)
            if (!s->wl[p]) AB_HIP(hipMalloc(&s->wl[p], bytes));
    }
    hipStream_t st = (mem == AB_MEM_HOST) ? s->stream : (hipStream_t)stream;
    const void *hin[8] = {f->T_s, f->theta_zt, f->q_s, f->q_zt, f->U_zu, skin ? f->Qsw : nullptr, skin ? f->rad_lw : nullptr,
                          skin ? f->slp : nullptr};
    void *hout[6] = {f->Cd, f->Ch, f->Ce, f->t_zu, f->q_zu, f->Ubzu};
    const void *din[8];
    void *dout[6];
    if (mem == AB_MEM_HOST) {
        int rc = stage_inputs(s, hin, din);
        if (rc) return rc;
        for (int i = 0; i < 6; ++i) {
            if (!s->stage_out[i]) AB_HIP(hipMalloc(&s->stage_out[i], bytes));
            dout[i] = s->stage_out[i];
        }
    } else {
        for (int i = 0; i < 8; ++i) din[i] = hin[i];
        for (int i = 0; i < 6; ++i) dout[i] = hout[i];
    }
    ab::TurbCall c;
    memset(&c, 0, sizeof c);
    c.T_s = const_cast<void *>(din[0]); c.theta_zt = din[1]; c.q_s = const_cast<void *>(din[2]); c.q_zt = din[3]; c.U_zu = din[4];
    c.qsw = din[5]; c.rad_lw = din[6]; c.slp = din[7]; c.lon = s->d_lon;
    for (int i = 0; i < 6; ++i) c.out[i] = dout[i];
    if (s->diag_on) {
        for (int i = 6; i < 16; ++i) {
            if (!s->diag_user[i]) continue;
            if (s->diag_mem == AB_MEM_DEVICE) { c.out[i] = s->diag_user[i]; continue; }
            if (!s->diag_dev[i]) AB_HIP(hipMalloc(&s->diag_dev[i], bytes));
            c.out[i] = s->diag_dev[i];
        }
    }
    for (int p = 0; p < 4; ++p) c.wl[p] = s->wl[p];
    c.n = s->n; c.zt = zt; c.zu = zu;
    c.algo = s->algo; c.skin = skin; c.f32 = s->f32; c.nb_iter = nb_iter;
    c.wl_load = (use_wl && kt > 1) ? 1 : 0;
    c.wl_store = use_wl ? 1 : 0;
    c.isecday = s->isecday;
    c.regroup = s->regroup;
    AB_HIP(hipEventRecord(s->ev0, st));
    AB_HIP(ab::launch_turb(c, st));
    AB_HIP(hipEventRecord(s->ev1, st));
    s->timed = true;
    s->last_stream = st;
    s->last_jt = kt;
    if (s->diag_on && s->diag_mem == AB_MEM_HOST) {
        for (int i = 6; i < 16; ++i)
            if (s->diag_user[i]) AB_HIP(hipMemcpyAsync(s->diag_user[i], s->diag_dev[i], bytes, hipMemcpyDeviceToHost, st));
        if (mem != AB_MEM_HOST) AB_HIP(hipStreamSynchronize(st));
    }
    if (mem == AB_MEM_HOST) {
        for (int i = 0; i < 6; ++i) AB_HIP(hipMemcpyAsync(hout[i], dout[i], bytes, hipMemcpyDeviceToHost, st));
        if (skin) {
            AB_HIP(hipMemcpyAsync(f->T_s, din[0], bytes, hipMemcpyDeviceToHost, st));
            AB_HIP(hipMemcpyAsync(f->q_s, din[2], bytes, hipMemcpyDeviceToHost, st));
        }
        AB_HIP(hipStreamSynchronize(st));
    } else {
        s->done_pending = true;  // (ev1: see ab_session_compute)
    }
    return AB_OK;
}

int ab_session_check(ab_session *s)
{
    if (!s) return fail(AB_ERR_ARG, "NULL session");
    if (s->sharded()) return ab::sharded_check(s);
    ab::DeviceGuard dguard_;
    AB_HIP(hipSetDevice(s->device));
    hipStream_t st = s->last_stream;
    int fl[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // [0] the error flag; [4..6] the tile counters of flux_kernel_cu
    AB_HIP(hipMemcpyAsync(fl, s->d_flags, sizeof fl, hipMemcpyDeviceToHost, st));
    AB_HIP(hipStreamSynchronize(st));
    const int flags = fl[0];
    // The persistent kernel's tile counters are zero between launches (its last team re-arms them).  Anything else here means that a
    // launch did not run to its end or that two launches of this session overlapped (the one-stream rule of ab_session_compute): tiles
    // of the records since the last check may have been skipped or computed twice.  Reported, and the counters put right again.
    const bool queue_bad = fl[4] != 0 || fl[5] != 0 || fl[6] != 0;
    if (flags || queue_bad) {
        AB_HIP(hipMemsetAsync(s->d_flags, 0, sizeof fl, st));
        AB_HIP(hipStreamSynchronize(st));
        if (queue_bad)
            return fail(AB_ERR_STATE, "the tile counters of the persistent flux kernel were left at %d / %d / %d: a launch of this session was cut "
                                      "short, or two of its launches overlapped (one stream per session); the results since the last check are not valid",
                        fl[4], fl[5], fl[6]);
        if (flags & 2)
            return fail(AB_ERR_HIP, "the flux kernel read other arguments from its kernarg segment than it was passed by value: this build of the "
                                    "library does not match the HIP runtime's argument layout (kernarg_at, ab_kernels.hip); results are not valid");
        if (flags & 1) return fail(AB_ERR_TAU, "BULK_FORMULA_VCTR()@mod_phymbl: wind stress too strong!\n => > 10 N/m^2 !");
    }
    return AB_OK;
}

int ab_session_get_wl_state(ab_session *s, double *state4n)
{
    if (!s || !state4n) return fail(AB_ERR_ARG, "NULL argument");
    if (s->sharded()) return ab::sharded_get_wl_state(s, state4n);
    if (!s->wl[0]) return fail(AB_ERR_STATE, "session keeps no persistent warm-layer state (no skin scheme or nt == 1)");
    ab::DeviceGuard dguard_;
    AB_HIP(hipSetDevice(s->device));
    AB_HIP(hipDeviceSynchronize());
    const size_t n = (size_t)s->n;
    for (int p = 0; p < 4; ++p) {
        double *dst = state4n + (size_t)p * n;
        if (!s->wl[p]) { for (size_t k = 0; k < n; ++k) dst[k] = 0.; continue; }
        if (!s->f32) {
            AB_HIP(hipMemcpy(dst, s->wl[p], n * sizeof(double), hipMemcpyDeviceToHost));
        } else {
            std::vector<float> tmp(n);
            AB_HIP(hipMemcpy(tmp.data(), s->wl[p], n * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < n; ++k) dst[k] = tmp[k];
        }
    }
    return AB_OK;
}

double ab_session_last_kernel_ms(ab_session *s)
{
    if (s && s->sharded()) return ab::sharded_last_kernel_ms(s);
    if (!s || !s->timed) return -1.;
    ab::DeviceGuard dguard_;
    if (hipSetDevice(s->device) != hipSuccess) return -1.;
    if (hipEventSynchronize(s->ev1) != hipSuccess) return -1.;
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, s->ev0, s->ev1) != hipSuccess) return -1.;
    return (double)ms;
}

/* bench / test utility: synthetic fields of SURVEY.md §8d written straight into device arrays */
int ab_synth_fields_device(void *sst, void *t_zt, void *q_zt, void *u, void *v, void *slp, void *rad_sw,
                           void *rad_lw, long ni, long j0, long nj_local, int precision, void *stream)
{
    AB_HIP(ab::launch_synth(sst, t_zt, q_zt, u, v, slp, rad_sw, rad_lw, ni, j0, nj_local, precision == AB_F32,
                            (hipStream_t)stream));
    return AB_OK;
}

/* test hook: elementwise fp64 device math (host arrays in/out) */
int ab_test_math(int op, const double *x, const double *y, double *out, long n)
{
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    const size_t bytes = sizeof(double) * (size_t)n;
    AB_HIP(hipMalloc((void **)&dx, bytes));
    AB_HIP(hipMalloc((void **)&dout, bytes));
    AB_HIP(hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice));
    if (y) {
        AB_HIP(hipMalloc((void **)&dy, bytes));
        AB_HIP(hipMemcpy(dy, y, bytes, hipMemcpyHostToDevice));
    }
    AB_HIP(ab::launch_math_test(op, dx, dy, dout, n, nullptr));
    AB_HIP(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dout);
    if (dy) (void)hipFree(dy);
    return AB_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// One shard of a sharded session at AEROBULK_MODEL's jt == 1 (ab_sharded.hip, sharded_model_first_record): the shard's pipelined
// pass with AEROBULK_INIT's statistics riding on it; the statistics are handed back (the verdict is global), `keep` holds what a
// redo needs.  Called from the shard's worker thread.
namespace ab {
// staging buffers of the host path, allocated from the calling thread, one shard after the other (sharded_model_first_record: measured —
// when the shards' worker threads allocate them concurrently inside their first pipelined pass, every record that follows moves
// over PCIe at 63 GB/s instead of 85, profiles/r3_host_path.txt)
int leaf_prepare_staging(ab_session *s, int with_rad, int with_ts)
{
    ab::DeviceGuard dguard_;
    AB_HIP(hipSetDevice(s->device));
    const size_t bytes = s->esz * (size_t)s->n;
    const bool first = !s->stage_in[0];
    for (int i = 0; i < 8; ++i)
        if ((i < 6 || with_rad) && !s->stage_in[i]) AB_HIP(hipMalloc(&s->stage_in[i], bytes));
    for (int i = 0; i < 6; ++i)
        if ((i < 5 || with_ts) && !s->stage_out[i]) AB_HIP(hipMalloc(&s->stage_out[i], bytes));
    if (first) {
        // The shard's copy streams carry their FIRST transfer here, alone — one chunk in, one chunk out, between the staging buffers
        // and a private host scratch (round 3 used the caller's own arrays: a record that failed afterwards left uninitialised device
        // memory in the caller's Q_L).  Measured on k shards of one device (profiles/r3_host_path.txt): when the shards' very first
        // transfers are the concurrent chunk pipelines of their workers, EVERY later record moves at 63 GB/s instead of 85 (20.7 ->
        // 27.8 ms per ORCA12 record, sticky for the life of the process); with each shard's first transfer made alone it does not happen.
        const size_t cb = bytes < (size_t)kPipeChunk * s->esz ? bytes : (size_t)kPipeChunk * s->esz;
        std::vector<char> scratch(cb, 0);
        AB_HIP(hipMemcpyAsync(s->stage_in[0], scratch.data(), cb, hipMemcpyHostToDevice, s->s_h2d));
        AB_HIP(hipStreamSynchronize(s->s_h2d));
        AB_HIP(hipMemsetAsync(s->stage_out[0], 0, cb, s->s_d2h));
        AB_HIP(hipMemcpyAsync(scratch.data(), s->stage_out[0], cb, hipMemcpyDeviceToHost, s->s_d2h));
        AB_HIP(hipStreamSynchronize(s->s_d2h));
    }
    return AB_OK;
}
int leaf_fused_first_record(ab_session *leaf, double zt, double zu, int niter, const void *const in[8], void *const out[6], int have_rad,
                            double stats[AB_INIT_NSTATS], int *guess, FusedShard **keep)
{
    FusedInit fi;
    fi.have_rad = have_rad;
    fi.defer = true;
    int rc = compute_impl(leaf, 1, zt, zu, niter, in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7], out[0], out[1], out[2], out[3],
                          out[4], out[5], AB_MEM_HOST, nullptr, &fi);
    if (rc) { delete fi.keep; return rc; }
    if (!fi.keep)   // the leaf took the plain path (fewer than kPipeThreshold cells, or diagnostics on): no statistics were taken
        return fail(AB_ERR_STATE, "fused first record asked of a shard of %ld cells that cannot pipeline it", leaf->n);
    memcpy(stats, fi.stats, sizeof fi.stats);
    *guess = fi.guess;
    *keep = fi.keep;
    return AB_OK;
}
int leaf_fused_redo(ab_session *s, FusedShard *k)
{
    ab::DeviceGuard dguard_;
    AB_HIP(hipSetDevice(s->device));
    k->c.hum_type = s->hum_type;
    AB_HIP(hipMemsetAsync(s->d_flags, 0, sizeof(int), s->stream));   // whatever the misjudged pass flagged is void
    AB_HIP(ab::launch_flux(k->c, s->stream));
    for (int i = 0; i < 6; ++i)
        if (k->hout[i]) AB_HIP(hipMemcpyAsync(k->hout[i], k->dout[i], k->bytes, hipMemcpyDeviceToHost, s->stream));
    AB_HIP(hipStreamSynchronize(s->stream));
    return AB_OK;
}
void leaf_fused_release(FusedShard *k) { delete k; }
}  // namespace ab

extern "C" {

// device-resident fields, one set of pointers per shard (include/aerobulk_amd.h)
int ab_session_compute_shards(ab_session *s, int jt, double zt, double zu, int niter, const ab_shard_arrays *sh, void *const *streams)
{
    if (!s || !sh) return fail(AB_ERR_ARG, "ab_session_compute_shards: NULL argument");
    if (s->sharded()) return ab::sharded_compute_shards(s, jt, zt, zu, niter, sh, streams);
    return ab_session_compute(s, jt, zt, zu, niter, sh[0].sst, sh[0].t_zt, sh[0].hum_zt, sh[0].u_zu, sh[0].v_zu, sh[0].slp, sh[0].rad_sw,
                              sh[0].rad_lw, sh[0].ql, sh[0].qh, sh[0].tau_x, sh[0].tau_y, sh[0].evap, sh[0].t_s, AB_MEM_DEVICE,
                              streams ? streams[0] : nullptr);
}

int ab_session_gather(ab_session *s, int root_shard, const ab_shard_arrays *sh, const ab_flux_arrays *dst, void *const *streams,
                      int synchronize)
{
    if (!s || !sh || !dst) return fail(AB_ERR_ARG, "ab_session_gather: NULL argument");
    if (s->sharded()) return ab::sharded_gather(s, root_shard, sh, dst, streams, synchronize);
    if (root_shard != 0) return fail(AB_ERR_ARG, "ab_session_gather: root shard %d of a one-shard session", root_shard);
    ab::DeviceGuard dguard_;
    AB_HIP(hipSetDevice(s->device));
    hipStream_t st = streams ? (hipStream_t)streams[0] : nullptr;
    const void *src[6] = {sh[0].ql, sh[0].qh, sh[0].tau_x, sh[0].tau_y, sh[0].evap, sh[0].t_s};
    void *d[6] = {dst->ql, dst->qh, dst->tau_x, dst->tau_y, dst->evap, dst->t_s};
    for (int f = 0; f < 6; ++f) {
        if (!d[f]) continue;
        if (!src[f]) return fail(AB_ERR_ARG, "ab_session_gather: field %d wanted but the shard has no such output", f);
        if (d[f] != src[f]) AB_HIP(hipMemcpyAsync(d[f], src[f], s->esz * (size_t)s->n, hipMemcpyDeviceToDevice, st));
    }
    if (synchronize) AB_HIP(hipStreamSynchronize(st));
    return AB_OK;
}

// ------------------------------------------------------------------------------------------------
// AEROBULK_MODEL on a process-global session (mod_aerobulk.f90:176-269) — non-reentrant by contract.
static ab_session *g_sess = nullptr;
static int g_nb_iter = 5;  // mod_const.f90:33 ; sticky once set through Niter (mod_aerobulk.f90:236)

// The process-global session of AEROBULK_MODEL.  AEROBULK_AMD_DEVICES (unset: the current device) spreads it over several
// GPUs by row blocks without a change to the calling Fortran / C++ code: "all", a count ("4": devices 0..3) or a list
// ("0,2,5"; a device may repeat: several shards on it).
static int create_model_session(ab_session **out, int algo, long ni, long nj, int nt, int use_skin)
{
    const char *e = getenv("AEROBULK_AMD_DEVICES");
    if (!e || !*e) return ab_session_create(out, algo, ni, nj, nt, use_skin, AB_F64, -1);
    if (strcmp(e, "all") == 0) return ab_session_create(out, algo, ni, nj, nt, use_skin, AB_F64, AB_DEVICE_ALL);
    std::vector<int> devs;
    if (strchr(e, ',')) {
        for (const char *p = e; *p;) {
            devs.push_back(atoi(p));
            const char *c = strchr(p, ',');
            if (!c) break;
            p = c + 1;
        }
    } else {
        for (int d = 0; d < atoi(e); ++d) devs.push_back(d);
    }
    if (devs.empty()) return fail(AB_ERR_ARG, "AEROBULK_AMD_DEVICES=%s: no device", e);
    if (devs.size() == 1) return ab_session_create(out, algo, ni, nj, nt, use_skin, AB_F64, devs[0]);
    return ab_session_create_sharded(out, algo, ni, nj, nt, use_skin, AB_F64, devs.data(), (int)devs.size());
}

int ab_model(int jt, int nt, const char *calgo, int calgo_len, double zt, double zu, const double *sst,
             const double *t_zt, const double *hum_zt, const double *u_zu, const double *v_zu, const double *slp,
             double *ql, double *qh, double *tau_x, double *tau_y, double *evap, int niter, int use_skin,
             const double *rad_sw, const double *rad_lw, double *t_s, long ni, long nj, ab_init_report *report)
{
    if (niter > 0) g_nb_iter = niter;                                        // :236
    const bool lsrad = rad_sw && rad_lw;                                     // :242
    if (jt < 1) return fail(AB_ERR_JT, "AEROBULK_MODEL => jt < 1 !??\n we are in a Fortran world here...");  // :244
    const int algo = ab_algo_from_string(calgo, calgo_len);
    if (jt == 1) {                                                           // AEROBULK_INIT :248 / :257
        if (use_skin) {                                                      // :67-79
            if (!(algo_has_skin(algo) || (calgo && (calgo_len < 0 ? strlen(calgo) : (size_t)calgo_len) >= 4 && strncmp(calgo, "coar", 4) == 0)))
                return fail(AB_ERR_SKIN_ALGO, " AEROBULK_INIT => Only `COARE*` and `ECMWF` algorithms support cool-skin & warm/layer schemes");
            if (!lsrad) return fail(AB_ERR_SKIN_NORAD, " AEROBULK_INIT => provide SW and LW rad. input if you want to use skin schemes");
        }
        if (algo == 0)                                                       // mod_aerobulk_compute.f90:173-175
            return fail(AB_ERR_ALGO, "ERROR: mod_aerobulk_compute.f90 => bulk algorithm %.*s is unknown!!!",
                        calgo_len < 0 ? (int)strlen(calgo ? calgo : "") : calgo_len, calgo ? calgo : "");
        const bool reuse = g_sess && g_sess->algo == algo && g_sess->ni == ni && g_sess->nj == nj && g_sess->nt == nt &&
                           g_sess->use_skin == (use_skin ? 1 : 0) && !g_sess->f32;
        if (!reuse) {
            if (g_sess) { ab_session_destroy(g_sess); g_sess = nullptr; }
            int rc = create_model_session(&g_sess, algo, ni, nj, nt, use_skin);
            if (rc) return rc;
        }
        g_sess->last_jt = 0;
        for (ab_session *c : g_sess->shards) c->last_jt = 0;
        // The reference's order (mod_aerobulk.f90:246-262) is the default: AEROBULK_INIT decides BEFORE anything is computed, so an
        // AB_ERR_ALL_MASKED / AB_ERR_HUM_TYPE / AB_ERR_UNITS leaves the caller's output arrays untouched (round 3 did the opposite by
        // default).  AEROBULK_AMD_FUSED_INIT=1 opts into the one-pass first record: the statistics ride on the pipelined pass of
        // aerobulk_compute (the inputs cross PCIe once while the outputs already travel back: about 21 ms instead of 31 ms for
        // record 1 of an ORCA12 grid, once per run) at the price documented in include/aerobulk_amd.h — outputs written before the verdict.
        const char *fe = getenv("AEROBULK_AMD_FUSED_INIT");
        const bool fused_ok = fe && fe[0] == '1' && !getenv("AEROBULK_AMD_NO_FUSED_INIT") && (!lsrad || use_skin);
        if (fused_ok && !g_sess->sharded() && g_sess->n >= kPipeThreshold && !g_sess->diag_on) {
            FusedInit fi;
            fi.have_rad = lsrad ? 1 : 0;
            fi.report = report;
            // rad present but l_use_skin false => no skin, T_s = sst (mod_aerobulk_compute.f90:132,206): handled by compute
            return compute_impl(g_sess, jt, zt, zu, g_nb_iter, sst, t_zt, hum_zt, u_zu, v_zu, slp, rad_sw, rad_lw, ql, qh, tau_x, tau_y,
                                evap, lsrad ? t_s : nullptr, AB_MEM_HOST, nullptr, &fi);
        }
        if (g_sess->sharded()) {
            // every shard's copy streams make their first transfer alone (leaf_prepare_staging: the concurrency trap)
            int rc = ab::sharded_prepare_staging(g_sess, lsrad && use_skin, lsrad && t_s);   // (radiation without l_use_skin is never read: mod_aerobulk_compute.f90:132; no staging for it)
            if (rc) return rc;
            // fused, sharded: every shard on its own device and PCIe link, the verdict on the combined statistics.  EVERY shard has
            // to qualify (the first shards are one row taller: the last may sit just under the threshold; round-3 advisory)
            bool all = fused_ok;
            for (ab_session *c : g_sess->shards) all = all && c->n >= kPipeThreshold && !c->diag_on;
            if (all) {
                const void *in8[8] = {sst, t_zt, hum_zt, u_zu, v_zu, slp, rad_sw, rad_lw};
                void *out6[6] = {ql, qh, tau_x, tau_y, evap, lsrad ? (void *)t_s : nullptr};
                return ab::sharded_model_first_record(g_sess, zt, zu, g_nb_iter, in8, out6, lsrad ? 1 : 0, report);
            }
        }
        // the reference hands rad_lw to BOTH prsw and prlw (mod_aerobulk.f90:248)
        int rc = ab_session_init(g_sess, sst, t_zt, hum_zt, u_zu, v_zu, slp, lsrad ? rad_lw : nullptr,
                                 lsrad ? rad_lw : nullptr, AB_MEM_HOST, nullptr, report);
        if (rc) return rc;
        // aerobulk_compute reads the very arrays AEROBULK_INIT has just staged in HBM: one PCIe crossing, not two
        g_sess->reuse_staged = true;
        for (ab_session *c : g_sess->shards) c->reuse_staged = true;
    } else {
        if (!g_sess) return fail(AB_ERR_STATE, "AEROBULK_MODEL called with jt=%d before jt=1", jt);
        // the reference keeps using the settings of jt == 1 silently; a caller that changes them mid-loop has a bug
        if (g_sess->algo != algo || g_sess->ni != ni || g_sess->nj != nj || g_sess->nt != nt ||
            g_sess->use_skin != (use_skin ? 1 : 0))
            return fail(AB_ERR_STATE, "AEROBULK_MODEL: algorithm, shape, Nt or l_use_skin changed between time records "
                                      "(jt=%d: %s %ldx%ld Nt=%d skin=%d, session: %s %ldx%ld Nt=%d skin=%d)", jt, ab_algo_name(algo), ni, nj,
                        nt, use_skin ? 1 : 0, ab_algo_name(g_sess->algo), g_sess->ni, g_sess->nj, g_sess->nt, g_sess->use_skin);
    }
    // rad present but l_use_skin false => no skin, T_s = sst (mod_aerobulk_compute.f90:132,206)
    return ab_session_compute(g_sess, jt, zt, zu, g_nb_iter, sst, t_zt, hum_zt, u_zu, v_zu, slp, rad_sw, rad_lw, ql, qh,
                              tau_x, tau_y, evap, lsrad ? t_s : nullptr, AB_MEM_HOST, nullptr);
}

// TURB_* on process-global state: one hidden session per algorithm (the reference's module SAVE arrays)
static ab_session *g_turb[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

int ab_turb(int algo, int kt, double zt, double zu, int use_cs, int use_wl, int nb_iter, int isecday_utc, const double *lon,
            double *T_s, const double *theta_zt, double *q_s, const double *q_zt, const double *U_zu, const double *Qsw,
            const double *rad_lw, const double *slp, double *Cd, double *Ch, double *Ce, double *t_zu, double *q_zu,
            double *Ubzu, const ab_diag *opt, long ni, long nj)
{
    if (algo < AB_ALGO_COARE3P0 || algo > AB_ALGO_ANDREAS) return fail(AB_ERR_ALGO, "bulk algorithm id %d is unknown!!!", algo);
    ab_session *&s = g_turb[algo];
    if (s && (s->ni != ni || s->nj != nj)) {
        if (kt > 1 && use_wl) return fail(AB_ERR_STATE, "TURB_%s: shape changed between time steps", ab_algo_name(algo));
        ab_session_destroy(s);
        s = nullptr;
    }
    if (!s) {
        int rc = ab_session_create(&s, algo, ni, nj, 1, 0, AB_F64, -1);
        if (rc) return rc;
    }
    int rc = ab_session_set_solar_time(s, isecday_utc, (use_wl && algo != AB_ALGO_ECMWF) ? lon : nullptr, AB_MEM_HOST, nullptr);
    if (rc) return rc;
    ab_diag d;
    memset(&d, 0, sizeof d);
    if (opt) {
        d = *opt;
        d.Cd = d.Ch = d.Ce = d.t_zu = d.q_zu = d.Ubzu = nullptr;
    }
    rc = ab_session_set_diagnostics(s, opt ? &d : nullptr, AB_MEM_HOST);
    if (rc) return rc;
    ab_turb_fields f;
    f.T_s = T_s; f.theta_zt = theta_zt; f.q_s = q_s; f.q_zt = q_zt; f.U_zu = U_zu;
    f.Qsw = Qsw; f.rad_lw = rad_lw; f.slp = slp;
    f.Cd = Cd; f.Ch = Ch; f.Ce = Ce; f.t_zu = t_zu; f.q_zu = q_zu; f.Ubzu = Ubzu;
    return ab_session_turb(s, kt, zt, zu, use_cs, use_wl, nb_iter, &f, AB_MEM_HOST, nullptr);
}

/* The warm-layer state of the process-global TURB_<algo> session after the latest ab_turb call, plane by plane (NULL: not wanted): what the
 * reference keeps as PUBLIC module arrays of mod_skin_coare / mod_skin_ecmwf (dT_wl, Hz_wl, Qnt_ac, Tau_ac: mod_skin_coare.f90:31-36) */
int ab_turb_get_wl_state(int algo, double *dT_wl, double *Hz_wl, double *Qnt_ac, double *Tau_ac, long n)
{
    if (algo < AB_ALGO_COARE3P0 || algo > AB_ALGO_ANDREAS) return fail(AB_ERR_ALGO, "bulk algorithm id %d is unknown!!!", algo);
    ab_session *s = g_turb[algo];
    if (!s || !s->wl[0]) return fail(AB_ERR_STATE, "TURB_%s has not been called with its warm layer on: no state", ab_algo_name(algo));
    if (n != s->n) return fail(AB_ERR_ARG, "ab_turb_get_wl_state: %ld cells asked for, the TURB_%s session holds %ld", n, ab_algo_name(algo), (long)s->n);
    ab::DeviceGuard dguard_;
    AB_HIP(hipSetDevice(s->device));
    AB_HIP(hipDeviceSynchronize());
    double *dst[4] = {dT_wl, Hz_wl, Qnt_ac, Tau_ac};
    for (int p = 0; p < 4; ++p) {
        if (!dst[p]) continue;
        if (!s->wl[p]) { for (long k = 0; k < n; ++k) dst[p][k] = 0.; continue; }
        AB_HIP(hipMemcpy(dst[p], s->wl[p], (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    }
    return AB_OK;
}

int ab_turb_neutral_10m(int algo, int nb_iter, const void *U_N10, void *CdN10, void *ChN10, void *CeN10, void *z0, long n,
                        int precision, int mem, void *stream)
{
    Range trace_("ab_turb_neutral_10m");
    if (algo < AB_ALGO_COARE3P0 || algo > AB_ALGO_ECMWF)   // andreas: "YET TO BE CODED" + STOP in the reference (:190-191)
        return fail(AB_ERR_ALGO, "ERROR: algorithm %s is not supported yet!", ab_algo_name(algo));
    if (!U_N10 || !CdN10 || !ChN10 || !CeN10 || !z0 || n <= 0 || nb_iter < 0) return fail(AB_ERR_ARG, "ab_turb_neutral_10m: bad argument");
    if (precision != AB_F64 && precision != AB_F32) return fail(AB_ERR_ARG, "bad precision %d", precision);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(AB_ERR_HIP, "no HIP device visible: this engine has no CPU fallback");
    const size_t bytes = (precision == AB_F32 ? 4 : 8) * (size_t)n;
    if (mem == AB_MEM_DEVICE) {
        AB_HIP(ab::launch_neutral10(algo, nb_iter, U_N10, CdN10, ChN10, CeN10, z0, n, precision == AB_F32, (hipStream_t)stream));
        return AB_OK;
    }
    void *d = nullptr;
    AB_HIP(hipMalloc(&d, 5 * bytes));
    char *b = (char *)d;
    hipError_t e = hipMemcpy(b, U_N10, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = ab::launch_neutral10(algo, nb_iter, b, b + bytes, b + 2 * bytes, b + 3 * bytes, b + 4 * bytes, n, precision == AB_F32, nullptr);
    void *outs[4] = {CdN10, ChN10, CeN10, z0};
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipMemcpy(outs[i], b + (1 + i) * bytes, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    AB_HIP(e);
    return AB_OK;
}

// ---- sea ice: stateless; host callers are staged through a grow-only scratch
int ab_ice_algo_from_string(const char *calgo)
{
    static const char *names[6] = {"", "nemo", "an05", "lu12", "lg15", "easy"};
    if (!calgo) return 0;
    for (int i = 1; i <= 5; ++i)
        if (strcmp(calgo, names[i]) == 0) return i;
    return 0;
}

static int turb_ice_impl(int ice_algo, double zt, double zu, int nb_iter, const double cxn[3], const ab_ice_fields *f, long n,
                         int precision, int mem, void *stream);

int ab_turb_ice(int ice_algo, double zt, double zu, int nb_iter, const ab_ice_fields *f, long n, int precision, int mem,
                void *stream)
{
    if (ice_algo < AB_ICE_NEMO || ice_algo > AB_ICE_LG15) return fail(AB_ERR_ALGO, "sea-ice algorithm id %d is unknown!!!", ice_algo);
    const double none[3] = {0., 0., 0.};
    return turb_ice_impl(ice_algo, zt, zu, nb_iter, none, f, n, precision, mem, stream);
}

int ab_turb_ice_easy(double zt, double zu, int nb_iter, double CdN, double ChN, double CeN, const ab_ice_fields *f, long n,
                     int precision, int mem, void *stream)
{
    if (!(CdN > 0.) || !(ChN > 0.) || !(CeN > 0.)) return fail(AB_ERR_ARG, "TURB_ICE_EASY: neutral coefficients must be > 0");
    if (nb_iter < 1) return fail(AB_ERR_ARG, "TURB_ICE_EASY: nb_iter < 1");
    const double cxn[3] = {CdN, ChN, CeN};
    return turb_ice_impl(AB_ICE_EASY, zt, zu, nb_iter, cxn, f, n, precision, mem, stream);
}

static int turb_ice_impl(int ice_algo, double zt, double zu, int nb_iter, const double cxn[3], const ab_ice_fields *f, long n,
                         int precision, int mem, void *stream)
{
    Range trace_("ab_turb_ice");
    if (!f) return fail(AB_ERR_ARG, "ab_turb_ice: NULL fields");
    if (n <= 0 || nb_iter < 0) return fail(AB_ERR_ARG, "ab_turb_ice: bad n / nb_iter");
    if (precision != AB_F64 && precision != AB_F32) return fail(AB_ERR_ARG, "bad precision %d", precision);
    if (!f->Ts_i || !f->theta_zt || !f->qs_i || !f->q_zt || !f->U_zu) return fail(AB_ERR_ARG, "ab_turb_ice: NULL input field");
    if (!f->Cd || !f->Ch || !f->Ce || !f->t_zu || !f->q_zu || !f->Ub) return fail(AB_ERR_ARG, "ab_turb_ice: NULL output field");
    if ((ice_algo == AB_ICE_LU12 || ice_algo == AB_ICE_LG15) && !f->frice)
        return fail(AB_ERR_ARG, "TURB_ICE_%s needs the ice concentration `frice`", ice_algo == AB_ICE_LU12 ? "LU12" : "LG15");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(AB_ERR_HIP, "no HIP device visible: this engine has no CPU fallback");
    const size_t esz = precision == AB_F32 ? 4 : 8, bytes = esz * (size_t)n;
    const void *hin[6] = {f->Ts_i, f->theta_zt, f->qs_i, f->q_zt, f->U_zu,
                          (ice_algo == AB_ICE_LU12 || ice_algo == AB_ICE_LG15) ? f->frice : nullptr};
    if (f->CdN_frm && ice_algo != AB_ICE_LG15) return fail(AB_ERR_ARG, "ab_turb_ice: CdN_frm is an output of TURB_ICE_LG15(_IO) only");
    void *hout[14] = {f->Cd, f->Ch, f->Ce, f->t_zu, f->q_zu, f->Ub, f->CdN, f->ChN, f->CeN, f->z0, f->u_star, f->L, f->UN10, f->CdN_frm};
    ab::IceCall c;
    memset(&c, 0, sizeof c);
    c.n = n; c.zt = zt; c.zu = zu; c.algo = ice_algo; c.f32 = precision == AB_F32; c.nb_iter = nb_iter;
    for (int i = 0; i < 3; ++i) c.cxn[i] = cxn[i];
    if (mem == AB_MEM_DEVICE) {
        c.Ts_i = hin[0]; c.theta_zt = hin[1]; c.qs_i = hin[2]; c.q_zt = hin[3]; c.U_zu = hin[4]; c.frice = hin[5];
        for (int i = 0; i < 14; ++i) c.out[i] = hout[i];
        AB_HIP(ab::launch_turb_ice(c, (hipStream_t)stream));
        return AB_OK;
    }
    static void *scratch = nullptr;        // 20 planes, grow-only, one host caller at a time (like the reference)
    static size_t scratch_bytes = 0;
    static int scratch_dev = -1;
    int dev = 0;
    AB_HIP(hipGetDevice(&dev));
    if (scratch_bytes < 20 * bytes || dev != scratch_dev) {
        if (scratch) {
            if (scratch_dev >= 0) (void)hipSetDevice(scratch_dev);
            (void)hipFree(scratch);
            (void)hipSetDevice(dev);
        }
        scratch = nullptr; scratch_bytes = 0;
        AB_HIP(hipMalloc(&scratch, 20 * bytes));
        scratch_bytes = 20 * bytes;
        scratch_dev = dev;
    }
    char *base = (char *)scratch;
    const void *din[6];
    for (int i = 0; i < 6; ++i) {
        din[i] = nullptr;
        if (!hin[i]) continue;
        AB_HIP(hipMemcpyAsync(base + i * bytes, hin[i], bytes, hipMemcpyHostToDevice, nullptr));
        din[i] = base + i * bytes;
    }
    c.Ts_i = din[0]; c.theta_zt = din[1]; c.qs_i = din[2]; c.q_zt = din[3]; c.U_zu = din[4]; c.frice = din[5];
    for (int i = 0; i < 14; ++i) c.out[i] = hout[i] ? base + (6 + i) * bytes : nullptr;
    AB_HIP(ab::launch_turb_ice(c, nullptr));
    for (int i = 0; i < 14; ++i)
        if (hout[i]) AB_HIP(hipMemcpyAsync(hout[i], c.out[i], bytes, hipMemcpyDeviceToHost, nullptr));
    AB_HIP(hipStreamSynchronize(nullptr));
    return AB_OK;
}

static void stop_like_fortran(int rc)
{
    // ctl_stop, mod_const.f90:255-276: banner + message on stdout, then STOP
    printf(" *** E R R O R :  \n %s\n\n", ab_last_error());
    fflush(stdout);
    exit(rc > 0 ? rc : 1);
}

static void print_init_banner(const char *calgo, int l, long m, int nt, int niter_now, int skin, const ab_init_report &r)
{
    // AEROBULK_INIT banner, mod_aerobulk.f90:59-64,75-78,97-102,118-120,140,157
    printf("\n ===================================================================\n");
    printf("                    ----- AeroBulk_init -----\n\n");
    printf("     *** Bulk parameterization to be used => \"%.*s\"\n", l, calgo);
    if (skin) printf("        ==> will use the Cool-skin & Warm-layer scheme of `%.*s` !\n", l, calgo);
    else printf("     *** Cool-skin & Warm-layer schemes will NOT be used!\n");
    printf("     *** Computational domain shape: Ni x Nj = %05ld x %05d\n", m, 1);
    printf("     *** Number of time records that will be treated: %d\n", nt);
    printf("     *** Number of iterations in bulk algos: nb_iter  = %d\n", niter_now);
    printf("     *** Filling the `mask` array...\n");
    if (r.n_masked == 0) printf("         ==> no points need to be masked! :)\n");
    else printf("         ==> number of points to mask: %ld (out of %ld)\n", r.n_masked, r.n_cells);
    static const char *hn[3] = {"specific humidity [kg/kg]", "dew-point temperature [K]", "relative humidity [%]"};
    if (r.hum_type >= 0 && r.hum_type <= 2) printf("     *** Type of prescribed air humidity  `%s`\n", hn[r.hum_type]);
    printf(" ===================================================================\n");
}

static void print_bye_banner(void)
{
    // AEROBULK_BYE, mod_aerobulk.f90:164-170
    printf(" ===================================================================\n");
    printf("                    ----- AeroBulk_bye -----\n");
    printf(" ===================================================================\n\n");
}

void aerobulk_cxx_skin(const int *jt, const int *nt, const char *calgo, const double *zt, const double *zu,
                       const double *sst, const double *t_zt, const double *hum_zt, const double *u_zu,
                       const double *v_zu, const double *slp, double *ql, double *qh, double *tau_x, double *tau_y,
                       double *evap, const int *niter, const bool *l_skin, const double *rad_sw, const double *rad_lw,
                       double *t_s, const int *l, const int *m)
{
    ab_init_report rep;
    memset(&rep, 0, sizeof rep);
    const int skin = *l_skin ? 1 : 0;  // one byte, as the C++ caller passes it (aerobulk.cpp:10,107)
    int rc = ab_model(*jt, *nt, calgo, *l, *zt, *zu, sst, t_zt, hum_zt, u_zu, v_zu, slp, ql, qh, tau_x, tau_y, evap,
                      *niter, skin, rad_sw, rad_lw, t_s, (long)*m, 1, &rep);
    if (*jt == 1 && (rc == AB_OK || rc == AB_ERR_TAU)) print_init_banner(calgo, *l, *m, *nt, g_nb_iter, skin, rep);
    if (rc != AB_OK) stop_like_fortran(rc);
    if (*jt == *nt) print_bye_banner();  // mod_aerobulk.f90:267
}

void aerobulk_cxx_no_skin(const int *jt, const int *nt, const char *calgo, const double *zt, const double *zu,
                          const double *sst, const double *t_zt, const double *hum_zt, const double *u_zu,
                          const double *v_zu, const double *slp, double *ql, double *qh, double *tau_x,
                          double *tau_y, double *evap, const int *niter, const int *l, const int *m)
{
    ab_init_report rep;
    memset(&rep, 0, sizeof rep);
    int rc = ab_model(*jt, *nt, calgo, *l, *zt, *zu, sst, t_zt, hum_zt, u_zu, v_zu, slp, ql, qh, tau_x, tau_y, evap,
                      *niter, 0, nullptr, nullptr, nullptr, (long)*m, 1, &rep);
    if (*jt == 1 && (rc == AB_OK || rc == AB_ERR_TAU)) print_init_banner(calgo, *l, *m, *nt, g_nb_iter, 0, rep);
    if (rc != AB_OK) stop_like_fortran(rc);
    if (*jt == *nt) print_bye_banner();
}

}  // extern "C"
