// Issue cost of the VALU instructions the PMC categories do not name (conversions, fp64 compare / max / ldexp, selects, 64-bit
// moves and integer ops), relative to v_fma_f64.  Eight independent instructions per loop body and lane, 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/instr_rate2.hip -o build/var/instr_rate2 && build/var/instr_rate2
#include <hip/hip_runtime.h>
#include <cstdio>

// OT/IT: C types of destination / source registers; the asm string names %0 (dst) and %1 (src)
#define KERNEL(name, OT, IT, ASM)                                                                           \
    __global__ void __launch_bounds__(256) name(double *out, int n)                                         \
    {                                                                                                       \
        IT i0 = (IT)(threadIdx.x + 1), i1 = i0 + (IT)1, i2 = i0 + (IT)2, i3 = i0 + (IT)3;                  \
        OT o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0, o5 = 0, o6 = 0, o7 = 0;                                  \
        for (int i = 0; i < n; ++i) {                                                                       \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                 \
                asm volatile(ASM : "=v"(o0) : "v"(i0)); asm volatile(ASM : "=v"(o1) : "v"(i1));             \
                asm volatile(ASM : "=v"(o2) : "v"(i2)); asm volatile(ASM : "=v"(o3) : "v"(i3));             \
                asm volatile(ASM : "=v"(o4) : "v"(i0)); asm volatile(ASM : "=v"(o5) : "v"(i1));             \
                asm volatile(ASM : "=v"(o6) : "v"(i2)); asm volatile(ASM : "=v"(o7) : "v"(i3));             \
            }                                                                                               \
        }                                                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = (double)o0 + (double)o1 + (double)o2 + (double)o3 + (double)o4 + (double)o5 + (double)o6 + (double)o7; \
    }

KERNEL(k_fma64, double, double, "v_fma_f64 %0, %1, %1, %1")
KERNEL(k_max64, double, double, "v_max_f64 %0, %1, %1")
KERNEL(k_ldexp64, double, double, "v_ldexp_f64 %0, %1, 3")
KERNEL(k_cmp64, float, double, "v_cmp_lt_f64 vcc, %1, %1\n v_mov_b32 %0, 0")   // pair: subtract the v_mov_b32 below
KERNEL(k_cmp32, float, float, "v_cmp_lt_f32 vcc, %1, %1\n v_mov_b32 %0, 0")
KERNEL(k_mov32, float, float, "v_mov_b32 %0, %1")
KERNEL(k_mov64, double, double, "v_mov_b64 %0, %1")
KERNEL(k_cnd32, float, float, "v_cndmask_b32 %0, %1, %1, vcc")
KERNEL(k_cnd32s, float, float, "v_cndmask_b32_e64 %0, %1, %1, s[20:21]")
KERNEL(k_cnd32z, float, float, "v_cndmask_b32 %0, 0, %1, vcc")
KERNEL(k_cnd32w, float, float, "v_cmp_lt_f32 vcc, %1, %1\n v_cndmask_b32 %0, 0, %1, vcc")
KERNEL(k_cmp64s, float, double, "v_cmp_lt_f64 s[20:21], %1, %1\n v_mov_b32 %0, 0")
KERNEL(k_cmpx64, float, double, "v_cmp_class_f64 vcc, %1, 3\n v_mov_b32 %0, 0")
KERNEL(k_ashr32, int, int, "v_ashrrev_i32 %0, 3, %1")
KERNEL(k_or_b32, int, int, "v_or_b32 %0, %1, %1")
KERNEL(k_sub_u32, int, int, "v_sub_u32 %0, %1, %1")
KERNEL(k_max_f32, float, float, "v_max_f32 %0, %1, %1")
KERNEL(k_mul_f32, float, float, "v_mul_f32 %0, %1, %1")
KERNEL(k_lshl_or, int, int, "v_lshl_or_b32 %0, %1, 3, %1")
KERNEL(k_and_or, int, int, "v_and_or_b32 %0, %1, %1, %1")
KERNEL(k_add3, int, int, "v_add3_u32 %0, %1, %1, %1")
KERNEL(k_perm, int, int, "v_perm_b32 %0, %1, %1, %1")
KERNEL(k_cvt_f64_i32, double, int, "v_cvt_f64_i32 %0, %1")
KERNEL(k_cvt_i32_f64, int, double, "v_cvt_i32_f64 %0, %1")
KERNEL(k_cvt_f32_f64, float, double, "v_cvt_f32_f64 %0, %1")
KERNEL(k_cvt_f64_f32, double, float, "v_cvt_f64_f32 %0, %1")
KERNEL(k_cvt_f32_i32, float, int, "v_cvt_f32_i32 %0, %1")
KERNEL(k_frexp_exp64, int, double, "v_frexp_exp_i32_f64 %0, %1")
KERNEL(k_trunc64, double, double, "v_trunc_f64 %0, %1")
KERNEL(k_floor64, double, double, "v_floor_f64 %0, %1")
KERNEL(k_add_u32, int, int, "v_add_u32 %0, %1, %1")
KERNEL(k_and_b32, int, int, "v_and_b32 %0, %1, %1")
KERNEL(k_lshl_b32, int, int, "v_lshlrev_b32 %0, 3, %1")
KERNEL(k_bfi_b32, int, int, "v_bfi_b32 %0, %1, %1, %1")
KERNEL(k_mul_lo_u32, int, int, "v_mul_lo_u32 %0, %1, %1")
KERNEL(k_mul_u24, int, int, "v_mul_u32_u24 %0, %1, %1")
KERNEL(k_max_i32, int, int, "v_max_i32 %0, %1, %1")
KERNEL(k_min_i32, int, int, "v_min_i32 %0, %1, %1")
KERNEL(k_max_u32, int, int, "v_max_u32 %0, %1, %1")
KERNEL(k_min_f32, float, float, "v_min_f32 %0, %1, %1")
KERNEL(k_med3_f32, float, float, "v_med3_f32 %0, %1, %1, %1")
KERNEL(k_add_f32, float, float, "v_add_f32 %0, %1, %1")
KERNEL(k_max_f32_abs, float, float, "v_max_f32_e64 %0, |%1|, %1")
KERNEL(k_cvt_i32_f32, int, float, "v_cvt_i32_f32 %0, %1")
KERNEL(k_floor_f32, float, float, "v_floor_f32 %0, %1")
KERNEL(k_fract_f32, float, float, "v_fract_f32 %0, %1")
KERNEL(k_xor_b32, int, int, "v_xor_b32 %0, %1, %1")
KERNEL(k_lshr_b32, int, int, "v_lshrrev_b32 %0, 3, %1")
KERNEL(k_cmp_i32, float, int, "v_cmp_gt_i32 vcc, %1, %1\n v_mov_b32 %0, 0")
KERNEL(k_ldexp_f32, float, float, "v_ldexp_f32 %0, %1, 3")
KERNEL(k_sqrt_f32, float, float, "v_sqrt_f32 %0, %1")
KERNEL(k_lshl_b64, long long, long long, "v_lshlrev_b64 %0, 3, %1")
KERNEL(k_lshl_add_u64, long long, long long, "v_lshl_add_u64 %0, %1, 3, %1")
KERNEL(k_mad_u64_u32, long long, int, "v_mad_u64_u32 %0, vcc, %1, %1, 0")
KERNEL(k_readlane, float, float, "v_readlane_b32 s20, %1, 3\n v_mov_b32 %0, 0")
KERNEL(k_fma32, float, float, "v_fma_f32 %0, %1, %1, %1")
KERNEL(k_fmamk64, double, double, "v_fma_f64 %0, %1, %1, s[20:21]")
KERNEL(k_addlit64, double, double, "v_add_f64 %0, %1, 0.5")
KERNEL(k_rcp64, double, double, "v_rcp_f64 %0, %1")

int main()
{
    const int blocks = 256 * 4 * 8, n = 2000;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    struct K { const char *name; void (*fn)(double *, int); } ks[] = {
        {"v_fma_f64", k_fma64}, {"v_max_f64", k_max64}, {"v_ldexp_f64", k_ldexp64}, {"v_cmp_lt_f64 + v_mov_b32", k_cmp64},
        {"v_cmp_lt_f32 + v_mov_b32", k_cmp32}, {"v_mov_b32", k_mov32}, {"v_mov_b64", k_mov64}, {"v_cndmask_b32", k_cnd32},
        {"v_cndmask_b32_e64 sgpr mask", k_cnd32s}, {"v_cndmask_b32 0,v,vcc", k_cnd32z}, {"v_cmp_lt_f32 + v_cndmask_b32", k_cnd32w},
        {"v_cmp_lt_f64 sdst + v_mov_b32", k_cmp64s}, {"v_cmp_class_f64 + v_mov_b32", k_cmpx64}, {"v_ashrrev_i32", k_ashr32}, {"v_or_b32", k_or_b32},
        {"v_sub_u32", k_sub_u32}, {"v_max_f32", k_max_f32}, {"v_mul_f32", k_mul_f32}, {"v_lshl_or_b32", k_lshl_or}, {"v_and_or_b32", k_and_or},
        {"v_add3_u32", k_add3}, {"v_perm_b32", k_perm},
        {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_cvt_f32_f64", k_cvt_f32_f64},
        {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_i32", k_cvt_f32_i32}, {"v_frexp_exp_i32_f64", k_frexp_exp64},
        {"v_trunc_f64", k_trunc64}, {"v_floor_f64", k_floor64}, {"v_add_u32", k_add_u32}, {"v_and_b32", k_and_b32},
        {"v_lshlrev_b32", k_lshl_b32}, {"v_bfi_b32", k_bfi_b32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_u32_u24", k_mul_u24},
        {"v_max_i32", k_max_i32}, {"v_min_i32", k_min_i32}, {"v_max_u32", k_max_u32}, {"v_min_f32", k_min_f32}, {"v_med3_f32", k_med3_f32},
        {"v_floor_f32", k_floor_f32}, {"v_fract_f32", k_fract_f32}, {"v_xor_b32", k_xor_b32}, {"v_lshrrev_b32", k_lshr_b32},
        {"v_cmp_gt_i32 + v_mov_b32", k_cmp_i32}, {"v_ldexp_f32", k_ldexp_f32}, {"v_sqrt_f32", k_sqrt_f32},
        {"v_lshlrev_b64", k_lshl_b64}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_mad_u64_u32", k_mad_u64_u32},
        {"v_readlane_b32 + v_mov_b32", k_readlane}, {"v_fma_f32", k_fma32}, {"v_fma_f64 (sgpr src)", k_fmamk64},
        {"v_add_f64 (inline const)", k_addlit64}, {"v_rcp_f64", k_rcp64}};
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
            const double cyc = ms * 1e-3 * 2.4e9 / ((double)blocks * 4 / 1024 * n * 32);
            if (rep == 1) {
                if (!base) base = ms;
                printf("%-28s %8.3f ms  %5.2f x v_fma_f64   (~%.1f cycles per wave-instruction at 2.4 GHz)\n", k.name, ms, ms / base, cyc);
            }
        }
    return 0;
}
