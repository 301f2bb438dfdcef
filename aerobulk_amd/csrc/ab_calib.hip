// ab_calib.hip — `ab_calibrate()`: two fixed device workloads that tell a slow BOX from a slow KERNEL (bench.py `calib`).
//
// Leases of one MI355X pool differ by a few per cent in what the same binary delivers (power cap, clocks, temperature); a benchmark
// line taken on one lease cannot be compared with a line of another round unless both carry a number that depends on the box alone.
//   AB_CALIB_FMA_F64 : every lane of 4 waves per SIMD runs eight independent chains of v_fma_f64 (inline assembly: the compiler has
//                      nothing to fold).  The time is set by the fp64 VALU issue rate alone — the resource that binds the flux
//                      kernels.  66 TFLOP/s on the leases of round 6 (84 % of the guide's 78.6: the chains do not
//                      reach the peak issue rate, which does not matter — the workload is FIXED, its ratio between boxes is the figure).
//   AB_CALIB_HBM_COPY: one coalesced pass dst[i] = src[i] over 1 GiB (grid-stride, 16-byte accesses): bytes read + written per
//                      second, the resource that binds the sea-ice and helper kernels.
// Nothing of the flux path is involved; the kernels are timed with HIP events on the caller's stream.
#include "../../include/aerobulk_amd.h"
#include "ab_session.hpp"

#include <hip/hip_runtime.h>
#include <string>

namespace ab {
void set_last_error(const std::string &msg);

__global__ void __launch_bounds__(256) calib_fma_f64(double *out, int n)
{
    double x0 = 1.0 + threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const double a = 0.999999, b = 1e-7;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void __launch_bounds__(256) calib_copy(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n2)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) dst[i] = src[i];
}

static int calib_fail(const char *what, hipError_t e)
{
    set_last_error(std::string("ab_calibrate: ") + what + ": " + hipGetErrorString(e));
    return AB_ERR_HIP;
}
#define CAL_HIP(call)                                   \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) return calib_fail(#call, e_); \
    } while (0)
}  // namespace ab

extern "C" int ab_calibrate(int what, int device, void *stream_, double *ms_out, double *rate_out)
{
    using namespace ab;
    if (what != AB_CALIB_FMA_F64 && what != AB_CALIB_HBM_COPY) {
        set_last_error("ab_calibrate: unknown workload");
        return AB_ERR_ARG;
    }
    DeviceGuard guard_;
    CAL_HIP(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)stream_;
    hipDeviceProp_t prop;
    CAL_HIP(hipGetDeviceProperties(&prop, device));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CAL_HIP(hipEventCreate(&e0));
    CAL_HIP(hipEventCreate(&e1));
    float ms = 0.f;
    double rate = 0.;
    int rc = AB_OK;
    if (what == AB_CALIB_FMA_F64) {
        // 16 blocks of four waves per CU = four rounds of four waves per SIMD; n = 1500 x 32 FMAs per wave: ~1.5 ms
        const int blocks = cus * 16, n = 1500;
        double *out = nullptr;
        CAL_HIP(hipMalloc((void **)&out, sizeof(double) * (size_t)blocks * 256));
        hipLaunchKernelGGL(calib_fma_f64, dim3(blocks), dim3(256), 0, stream, out, n);      // warm: code object load, clocks
        CAL_HIP(hipEventRecord(e0, stream));
        hipLaunchKernelGGL(calib_fma_f64, dim3(blocks), dim3(256), 0, stream, out, n);
        CAL_HIP(hipEventRecord(e1, stream));
        hipError_t e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        (void)hipFree(out);
        if (e != hipSuccess) rc = calib_fail("fma kernel", e);
        rate = 2.0 * 64.0 * (double)blocks * 4.0 * (double)n * 32.0 / (ms * 1e-3) / 1e12;   // TFLOP/s
    } else {
        const size_t bytes = (size_t)1 << 30, n2 = bytes / sizeof(double2);
        double2 *src = nullptr, *dst = nullptr;
        CAL_HIP(hipMalloc((void **)&src, bytes));
        hipError_t e = hipMalloc((void **)&dst, bytes);
        if (e != hipSuccess) { (void)hipFree(src); return calib_fail("hipMalloc", e); }
        (void)hipMemsetAsync(src, 0, bytes, stream);
        const int blocks = cus * 32;
        hipLaunchKernelGGL(calib_copy, dim3(blocks), dim3(256), 0, stream, src, dst, n2);
        CAL_HIP(hipEventRecord(e0, stream));
        hipLaunchKernelGGL(calib_copy, dim3(blocks), dim3(256), 0, stream, src, dst, n2);
        CAL_HIP(hipEventRecord(e1, stream));
        e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        (void)hipFree(src); (void)hipFree(dst);
        if (e != hipSuccess) rc = calib_fail("copy kernel", e);
        rate = 2.0 * (double)bytes / (ms * 1e-3) / 1e9;                                      // GB/s, read + written
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (ms_out) *ms_out = ms;
    if (rate_out) *rate_out = rate;
    return rc;
}
