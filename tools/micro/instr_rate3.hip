// v_cndmask_b32 reading VCC: when is it slow?  (instr_rate2 measured 20 cycles for a bare chain, 4 with an SGPR-pair mask)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/instr_rate3.hip -o build/var/instr_rate3 && build/var/instr_rate3
#include <hip/hip_runtime.h>
#include <cstdio>
#define KERNEL(name, NI, ASM)                                                                              \
    __global__ void __launch_bounds__(256) name(float *out, int n)                                         \
    {                                                                                                       \
        float i0 = threadIdx.x + 1.f, i1 = i0 + 1.f;  double d0 = i0, d1 = i1;                                                       \
        float o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0, o5 = 0, o6 = 0, o7 = 0;                               \
        for (int i = 0; i < n; ++i) {                                                                       \
            _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                   \
                asm volatile(ASM : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3), "=v"(o4), "=v"(o5), "=v"(o6), "=v"(o7) : "v"(i0), "v"(i1), "v"(d0), "v"(d1) : "vcc", "s20", "s21"); \
        }                                                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = o0 + o1 + o2 + o3 + o4 + o5 + o6 + o7;                        \
    }
#define C8V "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n"
#define M8 "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %9\n v_mov_b32 %5, %9\n v_mov_b32 %6, %9\n v_mov_b32 %7, %9\n"
KERNEL(k_mov8, 8, M8)
KERNEL(k_bare8, 8, C8V)
KERNEL(k_vcmp_then8, 9, "v_cmp_lt_f32 vcc, %8, %9\n" C8V)
KERNEL(k_smov_then8, 8, "s_mov_b64 vcc, exec\n" C8V)
KERNEL(k_sand_then8, 8, "s_and_b64 vcc, exec, s[20:21]\n" C8V)
KERNEL(k_vcmps_smov_then8, 9, "v_cmp_lt_f32 s[20:21], %8, %9\n s_and_b64 vcc, exec, s[20:21]\n" C8V)
KERNEL(k_alt, 16, "v_cmp_lt_f32 vcc, %8, %9\n v_cndmask_b32 %0, %8, %9, vcc\n v_cmp_gt_f32 vcc, %8, %9\n v_cndmask_b32 %1, %8, %9, vcc\n v_cmp_lt_f32 vcc, %8, %9\n v_cndmask_b32 %2, %8, %9, vcc\n v_cmp_gt_f32 vcc, %8, %9\n v_cndmask_b32 %3, %8, %9, vcc\n"
              "v_cmp_lt_f32 vcc, %8, %9\n v_cndmask_b32 %4, %8, %9, vcc\n v_cmp_gt_f32 vcc, %8, %9\n v_cndmask_b32 %5, %8, %9, vcc\n v_cmp_lt_f32 vcc, %8, %9\n v_cndmask_b32 %6, %8, %9, vcc\n v_cmp_gt_f32 vcc, %8, %9\n v_cndmask_b32 %7, %8, %9, vcc\n")
KERNEL(k_alt64, 16, "v_cmp_lt_f64 vcc, %10, %11\n v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cmp_gt_f64 vcc, %10, %11\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
              "v_cmp_lt_f64 vcc, %10, %11\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cmp_gt_f64 vcc, %10, %11\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n")
int main()
{
    const int blocks = 256 * 4 * 8, n = 2000;
    float *out;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    struct K { const char *name; void (*fn)(float *, int); int ni; } ks[] = {
        {"8 v_mov_b32", k_mov8, 8}, {"8 v_cndmask vcc (vcc never written)", k_bare8, 8}, {"v_cmp_f32 vcc + 8 v_cndmask vcc", k_vcmp_then8, 9},
        {"s_mov_b64 vcc + 8 v_cndmask vcc", k_smov_then8, 8}, {"s_and_b64 vcc + 8 v_cndmask vcc", k_sand_then8, 8},
        {"v_cmp sdst, s_and vcc + 8 v_cndmask", k_vcmps_smov_then8, 9}, {"8 x (v_cmp_f32 vcc, v_cndmask vcc)", k_alt, 16},
        {"4 x (v_cmp_f64 vcc, 2 v_cndmask vcc)", k_alt64, 12}};
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (auto &k : ks) {
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, n);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, n);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double cyc = ms * 1e-3 * 2.4e9 / ((double)blocks * 4 / 1024 * n * 4);   // cycles per asm block per wave
            if (rep == 1) printf("%-42s %8.3f ms  %6.1f cycles per block of %d instructions\n", k.name, ms, cyc, k.ni);
        }
    return 0;
}
