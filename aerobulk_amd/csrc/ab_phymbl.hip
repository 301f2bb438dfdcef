// ab_phymbl.hip — `ab_phymbl()`: the public functions of the reference's mod_phymbl (src/mod_phymbl.f90:33-139) on arrays, as HIP
// kernels (gfx950).  One lane per cell, coalesced streaming of the argument arrays, the function itself from ab_phymbl.hpp.
// HBM-bound (8 B per argument and result, a few dozen fp64 operations): nothing to tile.  Scalars are arrays of one cell: there is
// no host implementation of any of this in the library.
#include "../../include/aerobulk_amd.h"
#include "ab_kernels.hpp"
#include "ab_phymbl.hpp"
#include "ab_session.hpp"

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace ab {
void set_last_error(const std::string &msg);

constexpr int kPhMaxIn = 11, kPhMaxOut = 7, kPhBlock = 256;

struct PhArgs {
    const double *in[kPhMaxIn + 1];   // (+1: the e_air iterate the two-pass functions add)
    double *out[kPhMaxOut];
    double par[2];
    long n;
    int flag;
    unsigned present;
    double *block_sums;               // e_air: sum over the block of |new - old| (one per block, combined in order on the host)
    unsigned long long *first_bad;    // bulk_formula: smallest cell index with tau > 10 N/m^2 (mod_phymbl.f90:1250-1253)
};

template <int FN> __global__ void __launch_bounds__(kPhBlock) phymbl_kernel(const PhArgs a)
{
    math_tables_init<double>();
    constexpr PhShape sh = ph_shape(FN);
    constexpr int n_in = sh.n_in + ((FN == kPhEair || FN == kPhRhoAirAdv || FN == kPhRhAir) ? 1 : 0);
    const long k = (long)blockIdx.x * kPhBlock + threadIdx.x;
    double dsum = 0.;
    if (k < a.n) {
        double x[n_in], y[kPhMaxOut];
#pragma unroll
        for (int i = 0; i < n_in; ++i) x[i] = ((a.present >> i) & 1u) ? a.in[i][k] : 0.;
        ph_cell<FN, double>(x, a.present, a.par, a.flag, y);
#pragma unroll
        for (int i = 0; i < sh.n_out; ++i)
            if (a.out[i]) a.out[i][k] = y[i];
        if (FN == kPhEair) dsum = __builtin_fabs(y[0] - x[2]);
        if (FN == kPhBulkFormula && y[0] > 10.) atomicMin(a.first_bad, (unsigned long long)k);
    }
    if (FN == kPhEair) {   // SUM( ABS( ee - e_old ) ), mod_phymbl.f90:1730: fixed-order tree inside the block
        __shared__ double s_part[kPhBlock];
        s_part[threadIdx.x] = dsum;
        __syncthreads();
        for (int s = kPhBlock / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) s_part[threadIdx.x] += s_part[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.block_sums[blockIdx.x] = s_part[0];
    }
}

template <int FN> static hipError_t launch_one(const PhArgs &a, hipStream_t stream)
{
    const long nblk = (a.n + kPhBlock - 1) / kPhBlock;
    hipLaunchKernelGGL((phymbl_kernel<FN>), dim3((unsigned)nblk), dim3(kPhBlock), 0, stream, a);
    return hipGetLastError();
}

template <int FN = 1> static hipError_t launch_fn(int fn, const PhArgs &a, hipStream_t stream)
{
    if constexpr (FN >= kPhCount) {
        return hipErrorInvalidValue;
    } else {
        if (fn == FN) return launch_one<FN>(a, stream);
        return launch_fn<FN + 1>(fn, a, stream);
    }
}

// ONE cell handed over by value: the scalar specifics of the Fortran modules (every `_sclr` of mod_phymbl, CS_* / WL_* of the skin modules) are calls
// on one-cell host arrays, and the reference's sweep drivers issue them by the hundred thousand.  The general path costs such a call two to ten small
// PCIe copies around its kernel (34 us); here the arguments travel in the kernel's argument segment and the results are written straight into a
// pinned, device-mapped host buffer: one launch, one synchronisation.  Same ph_cell<FN>, same bits.
struct PhScalarArgs {
    double x[kPhMaxIn + 1];
    double par[2];
    double *out;                      // kPhMaxOut doubles, host memory mapped into the device's address space
    int flag;
    unsigned present;
};
template <int FN> __global__ void __launch_bounds__(kPhBlock) phymbl_scalar_kernel(const PhScalarArgs a)
{
    math_tables_init<double>();
    if (threadIdx.x != 0) return;
    constexpr PhShape sh = ph_shape(FN);
    double x[kPhMaxIn + 1], y[kPhMaxOut];
#pragma unroll
    for (int i = 0; i < kPhMaxIn + 1; ++i) x[i] = a.x[i];
    ph_cell<FN, double>(x, a.present, a.par, a.flag, y);
#pragma unroll
    for (int i = 0; i < sh.n_out; ++i) a.out[i] = y[i];
}
template <int FN = 1> static hipError_t launch_scalar(int fn, const PhScalarArgs &a, hipStream_t stream)
{
    if constexpr (FN >= kPhCount) {
        return hipErrorInvalidValue;
    } else {
        if (fn == FN) {
            hipLaunchKernelGGL((phymbl_scalar_kernel<FN>), dim3(1), dim3(kPhBlock), 0, stream, a);
            return hipGetLastError();
        }
        return launch_scalar<FN + 1>(fn, a, stream);
    }
}

// e_air (mod_phymbl.f90:1706-1736): e <- q/eps (p - (1 - eps) e) from e = q p/eps until the SUM over the array of |change| is
// <= 1e-6 — a whole-array criterion, so the sweeps are separate launches with the sum brought to the host in between.  `e` ends up in
// d_e (device, n doubles); d_tmp is a second buffer of the same size.
static hipError_t e_air_device(const double *d_q, const double *d_p, double *d_e, double *d_tmp, long n, hipStream_t stream, bool *converged)
{
    *converged = false;
    const long nblk = (n + kPhBlock - 1) / kPhBlock;
    double *d_sums = nullptr;
    hipError_t e = hipMalloc((void **)&d_sums, sizeof(double) * (size_t)nblk);
    if (e != hipSuccess) return e;
    std::vector<double> sums((size_t)nblk);
    PhArgs a;
    memset(&a, 0, sizeof a);
    a.n = n;
    a.block_sums = d_sums;
    // first iterate e_old = q p / eps: the sweep kernel with the iterate's slot absent reads 0 there, q/eps (p - 0)
    a.in[0] = d_q; a.in[1] = d_p; a.in[2] = nullptr;
    a.present = 3u;
    a.out[0] = d_e;
    e = launch_one<kPhEair>(a, stream);
    double *cur = d_e, *nxt = d_tmp;
    for (int sweep = 0; e == hipSuccess && sweep < 200; ++sweep) {
        a.in[2] = cur;
        a.present = 7u;
        a.out[0] = nxt;
        e = launch_one<kPhEair>(a, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sums.data(), d_sums, sizeof(double) * (size_t)nblk, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) break;
        double zdiff = 0.;
        for (long b = 0; b < nblk; ++b) zdiff += sums[(size_t)b];
        double *t = cur; cur = nxt; nxt = t;
        if (!(zdiff > 1.e-6)) { *converged = true; break; }     // DO WHILE ( zdiff > repsilon ): a NaN sum leaves the loop there too
    }
    if (e == hipSuccess && cur != d_e) e = hipMemcpyAsync(d_e, cur, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream);
    (void)hipFree(d_sums);
    return e;
}

}  // namespace ab

namespace {
int ph_fail(int code, const char *msg)
{
    ab::set_last_error(msg);
    return code;
}
int ph_hip_fail(hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "ab_phymbl: %s failed: %s", what, hipGetErrorString(e));
    ab::set_last_error(buf);
    return AB_ERR_HIP;
}
}  // namespace

extern "C" int ab_phymbl(int fn, long n, const double *const *in, int n_in, double *const *out, int n_out, const double *par,
                         int flag, int mem, void *stream, double *info)
{
    using namespace ab;
    const PhShape sh = ph_shape(fn);
    if (sh.n_out == 0) return ph_fail(AB_ERR_ARG, "ab_phymbl: unknown function id");
    if (n < 0 || !in || !out) return ph_fail(AB_ERR_ARG, "ab_phymbl: bad n / NULL argument tables");
    if (n_in < sh.n_in_required || n_in > kPhMaxIn || n_out < 1) return ph_fail(AB_ERR_ARG, "ab_phymbl: wrong number of arrays for this function");
    if (n_in > sh.n_in) n_in = sh.n_in;          // the tables may be longer than this function needs: the rest is ignored
    if (n_out > sh.n_out) n_out = sh.n_out;
    if (mem != AB_MEM_HOST && mem != AB_MEM_DEVICE) return ph_fail(AB_ERR_ARG, "ab_phymbl: bad mem");
    unsigned present = 0;
    for (int i = 0; i < n_in; ++i)
        if (in[i]) present |= 1u << i;
    if ((present & ((1u << sh.n_in_required) - 1u)) != ((1u << sh.n_in_required) - 1u)) return ph_fail(AB_ERR_ARG, "ab_phymbl: a required input array is NULL");
    bool any_out = false;
    for (int i = 0; i < n_out; ++i) any_out = any_out || out[i];
    if (!any_out) return ph_fail(AB_ERR_ARG, "ab_phymbl: no output array");
    if (info) { info[0] = -1.; info[1] = 0.; }
    if (n == 0) return AB_OK;                    // zero-size arrays: the reference's elemental loops run no trip
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return ph_fail(AB_ERR_HIP, "no HIP device visible: this engine has no CPU fallback");

    // AB_MEM_DEVICE: the work goes to the device that OWNS the caller's arrays (scratch, block sums, launches), not to whatever device
    // is current in the calling thread; the caller's device is restored on return (as ab_session_* do)
    ab::DeviceGuard dguard_;
    if (mem == AB_MEM_DEVICE) {
        const void *first = nullptr;
        for (int i = 0; i < n_in && !first; ++i) first = in[i];
        hipPointerAttribute_t at;
        if (first && hipPointerGetAttributes(&at, first) == hipSuccess && at.type == hipMemoryTypeDevice) {
            if (hipSetDevice(at.device) != hipSuccess) return ph_fail(AB_ERR_HIP, "ab_phymbl: hipSetDevice to the arrays' device failed");
        } else {
            (void)hipGetLastError();
            return ph_fail(AB_ERR_ARG, "ab_phymbl: AB_MEM_DEVICE arrays are not device memory");
        }
    }
    const bool two_pass = fn == kPhRhoAirAdv || fn == kPhRhAir || fn == kPhEair;
    const bool tau_check = fn == kPhBulkFormula;
    const size_t bytes = sizeof(double) * (size_t)n;
    hipStream_t s = (hipStream_t)(mem == AB_MEM_DEVICE ? stream : nullptr);
    hipError_t e = hipSuccess;
    bool e_air_converged = true;

    // ---- one cell of a host caller (not the two-pass functions, not BULK_FORMULA's stress check): by value, see phymbl_scalar_kernel
    if (n == 1 && mem == AB_MEM_HOST && !two_pass && !tau_check) {
        static std::mutex mu;
        static double *h_out[64] = {nullptr}, *d_out[64] = {nullptr};
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            std::lock_guard<std::mutex> lock(mu);         // one scalar call at a time per process: the buffer is read below, under the lock
            if (!h_out[dev]) {
                e = hipHostMalloc((void **)&h_out[dev], sizeof(double) * kPhMaxOut, hipHostMallocMapped);
                if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&d_out[dev], h_out[dev], 0);
                if (e != hipSuccess) { h_out[dev] = nullptr; return ph_hip_fail(e, "hipHostMalloc (scalar results)"); }
            }
            PhScalarArgs sa;
            memset(&sa, 0, sizeof sa);
            for (int i = 0; i < n_in; ++i)
                if (in[i]) sa.x[i] = in[i][0];
            sa.par[0] = par ? par[0] : 0.; sa.par[1] = par ? par[1] : 0.;
            sa.out = d_out[dev];
            sa.flag = flag;
            sa.present = present;
            e = launch_scalar<>(fn, sa, nullptr);
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) return ph_hip_fail(e, "scalar kernel");
            for (int i = 0; i < n_out; ++i)
                if (out[i]) out[i][0] = h_out[dev][i];
            return AB_OK;
        }
    }

    // device scratch: staged inputs / outputs of a host call, the e_air iterates, the wind-stress flag
    const int n_stage = mem == AB_MEM_HOST ? n_in + n_out : 0;
    const int n_extra = two_pass ? 2 : 0;
    char *scratch = nullptr;
    const size_t scratch_bytes = (size_t)(n_stage + n_extra) * bytes + 64;
    e = hipMalloc((void **)&scratch, scratch_bytes);
    if (e != hipSuccess) return ph_hip_fail(e, "hipMalloc");
    unsigned long long *d_bad = (unsigned long long *)(scratch + (size_t)(n_stage + n_extra) * bytes);

    PhArgs a;
    memset(&a, 0, sizeof a);
    a.n = n;
    a.par[0] = par ? par[0] : 0.;
    a.par[1] = par ? par[1] : 0.;
    a.flag = flag;
    a.present = present;
    a.first_bad = d_bad;
    for (int i = 0; i < n_in; ++i) {
        if (!in[i]) continue;
        if (mem == AB_MEM_HOST) {
            double *d = (double *)(scratch + (size_t)i * bytes);
            if (e == hipSuccess) e = hipMemcpyAsync(d, in[i], bytes, hipMemcpyHostToDevice, s);
            a.in[i] = d;
        } else {
            a.in[i] = in[i];
        }
    }
    for (int i = 0; i < n_out; ++i) {
        if (!out[i]) continue;
        a.out[i] = mem == AB_MEM_HOST ? (double *)(scratch + (size_t)(n_in + i) * bytes) : out[i];
    }
    if (tau_check && e == hipSuccess) e = hipMemsetAsync(d_bad, 0xff, sizeof(unsigned long long), s);

    if (e == hipSuccess) {
        if (two_pass) {
            double *d_e = (double *)(scratch + (size_t)n_stage * bytes), *d_tmp = d_e + n;
            // e_air( pqa, pslp ): argument positions differ per function
            const double *d_q = fn == kPhRhoAirAdv ? a.in[1] : a.in[0];
            const double *d_p = fn == kPhEair ? a.in[1] : a.in[2];
            e = e_air_device(d_q, d_p, d_e, d_tmp, n, s, &e_air_converged);
            if (e == hipSuccess) {
                if (fn == kPhEair) {
                    e = hipMemcpyAsync(a.out[0], d_e, bytes, hipMemcpyDeviceToDevice, s);
                } else {
                    a.in[3] = d_e;
                    a.present |= 8u;
                    e = launch_fn<>(fn, a, s);
                }
            }
        } else {
            e = launch_fn<>(fn, a, s);
        }
    }
    unsigned long long bad = ~0ull;
    if (mem == AB_MEM_HOST) {
        for (int i = 0; i < n_out && e == hipSuccess; ++i)
            if (out[i]) e = hipMemcpyAsync(out[i], a.out[i], bytes, hipMemcpyDeviceToHost, s);
    }
    if (tau_check && e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, s);
    // the scratch is freed below: every path synchronises (a device caller's arrays are complete on return; helper calls are not
    // the hot path — ab_session_compute is the asynchronous entry)
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    double bad_tau = 0.;
    if (tau_check && e == hipSuccess && bad != ~0ull && a.out[0])
        e = hipMemcpy(&bad_tau, a.out[0] + bad, sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(scratch);
    if (e != hipSuccess) return ph_hip_fail(e, "kernel / copy");
    if (!e_air_converged)      // the reference would sweep for ever; the last iterate is in the output arrays
        return ph_fail(AB_ERR_NOCONV, "e_air()@mod_phymbl: SUM(ABS(ee - e_old)) still above 1e-6 after 200 sweeps over the array");
    if (tau_check && bad != ~0ull) {
        if (info) { info[0] = (double)bad; info[1] = bad_tau; }
        char buf[160];
        snprintf(buf, sizeof buf, "BULK_FORMULA_VCTR()@mod_phymbl: wind stress too strong! => %8.2f N/m^2 ! At cell %llu", bad_tau, bad);
        ab::set_last_error(buf);
        return AB_ERR_TAU;
    }
    return AB_OK;
}
