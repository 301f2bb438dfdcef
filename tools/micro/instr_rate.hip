// Issue cost of single VALU instructions on gfx950, relative to v_fma_f64: each kernel runs 8 independent chains of one
// instruction per lane, 4 waves per SIMD resident, so the time is set by the issue rate of that instruction alone.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/instr_rate.hip -o build/var/instr_rate && build/var/instr_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define KERNEL(name, decl, init, body)                                                                      \
    __global__ void __launch_bounds__(256) name(double *out, int n)                                         \
    {                                                                                                       \
        decl;                                                                                               \
        init;                                                                                               \
        for (int i = 0; i < n; ++i) {                                                                       \
            body; body; body; body;                                                                         \
        }                                                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = (double)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);             \
    }
#define D8 double x0, x1, x2, x3, x4, x5, x6, x7
#define F8 float x0, x1, x2, x3, x4, x5, x6, x7
#define INIT x0 = 1.0 + threadIdx.x * 1e-3; x1 = x0 + 1; x2 = x0 + 2; x3 = x0 + 3; x4 = x0 + 4; x5 = x0 + 5; x6 = x0 + 6; x7 = x0 + 7
#define EACH(op) asm volatile(op " %0, %0" : "+v"(x0)); asm volatile(op " %0, %0" : "+v"(x1)); asm volatile(op " %0, %0" : "+v"(x2)); \
                 asm volatile(op " %0, %0" : "+v"(x3)); asm volatile(op " %0, %0" : "+v"(x4)); asm volatile(op " %0, %0" : "+v"(x5)); \
                 asm volatile(op " %0, %0" : "+v"(x6)); asm volatile(op " %0, %0" : "+v"(x7))
#define EACH3(op) asm volatile(op " %0, %0, %0, %0" : "+v"(x0)); asm volatile(op " %0, %0, %0, %0" : "+v"(x1)); asm volatile(op " %0, %0, %0, %0" : "+v"(x2)); \
                  asm volatile(op " %0, %0, %0, %0" : "+v"(x3)); asm volatile(op " %0, %0, %0, %0" : "+v"(x4)); asm volatile(op " %0, %0, %0, %0" : "+v"(x5)); \
                  asm volatile(op " %0, %0, %0, %0" : "+v"(x6)); asm volatile(op " %0, %0, %0, %0" : "+v"(x7))
#define EACH2(op) asm volatile(op " %0, %0, %0" : "+v"(x0)); asm volatile(op " %0, %0, %0" : "+v"(x1)); asm volatile(op " %0, %0, %0" : "+v"(x2)); \
                  asm volatile(op " %0, %0, %0" : "+v"(x3)); asm volatile(op " %0, %0, %0" : "+v"(x4)); asm volatile(op " %0, %0, %0" : "+v"(x5)); \
                  asm volatile(op " %0, %0, %0" : "+v"(x6)); asm volatile(op " %0, %0, %0" : "+v"(x7))

KERNEL(k_fma64, D8, INIT, EACH3("v_fma_f64"))
KERNEL(k_mul64, D8, INIT, EACH2("v_mul_f64"))
KERNEL(k_add64, D8, INIT, EACH2("v_add_f64"))
KERNEL(k_rcp64, D8, INIT, EACH("v_rcp_f64"))
KERNEL(k_rsq64, D8, INIT, EACH("v_rsq_f64"))
KERNEL(k_sqrt64, D8, INIT, EACH("v_sqrt_f64"))
KERNEL(k_rndne64, D8, INIT, EACH("v_rndne_f64"))
KERNEL(k_frexpm64, D8, INIT, EACH("v_frexp_mant_f64"))
KERNEL(k_fma32, F8, INIT, EACH3("v_fma_f32"))
KERNEL(k_rcp32, F8, INIT, EACH("v_rcp_f32"))
KERNEL(k_rsq32, F8, INIT, EACH("v_rsq_f32"))
KERNEL(k_exp32, F8, INIT, EACH("v_exp_f32"))
KERNEL(k_log32, F8, INIT, EACH("v_log_f32"))
KERNEL(k_mov32, F8, INIT, EACH("v_mov_b32"))
KERNEL(k_pkfma32, D8, INIT, EACH3("v_pk_fma_f32"))
KERNEL(k_pkmul32, D8, INIT, EACH2("v_pk_mul_f32"))

int main()
{
    const int blocks = 256 * 4 * 8, n = 2000;   // 4 blocks of 4 waves per CU, 8 rounds
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    struct K { const char *name; void (*fn)(double *, int); } ks[] = {
        {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64}, {"v_add_f64", k_add64}, {"v_rcp_f64", k_rcp64}, {"v_rsq_f64", k_rsq64},
        {"v_sqrt_f64", k_sqrt64}, {"v_rndne_f64", k_rndne64}, {"v_frexp_mant_f64", k_frexpm64}, {"v_fma_f32", k_fma32},
        {"v_rcp_f32", k_rcp32}, {"v_rsq_f32", k_rsq32}, {"v_exp_f32", k_exp32}, {"v_log_f32", k_log32}, {"v_mov_b32", k_mov32},
        {"v_pk_fma_f32", k_pkfma32}, {"v_pk_mul_f32", k_pkmul32}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double base = 0;
    for (int rep = 0; rep < 2; ++rep)
        for (auto &k : ks) {
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, n);   // warm
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs * n*32 instructions
            const double cyc = ms * 1e-3 * 2.4e9 / ((double)blocks * 4 / 1024 * n * 32);
            if (rep == 1) {
                if (!base) base = ms;
                printf("%-18s %8.3f ms  %5.2f x v_fma_f64   (~%.1f cycles per wave-instruction at 2.4 GHz)\n", k.name, ms, ms / base, cyc);
            }
        }
    return 0;
}
