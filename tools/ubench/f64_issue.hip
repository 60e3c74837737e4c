// tools/ubench/f64_issue.hip -- issue cost of the f64 VALU instructions the iteration kernels are made of (gfx950).
// (the binary is git-ignored: build it in the container, it travels to the GPU box with the snapshot)
// Each kernel runs REPS x 8 independent copies of one instruction per wave, W waves per SIMD; cycles per wave instruction =
// elapsed s_memrealtime-free estimate: time / (REPS * 8 * W) * clock.  Build: hipcc -O3 --offload-arch=gfx950 -o f64_issue f64_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REPS 4096
#define OP1(name, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void name(double *out, double seed)                                 \
    {                                                                                                     \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < REPS; i++) {                                                                  \
            asm volatile(ASM(%0) "\n" ASM(%1) "\n" ASM(%2) "\n" ASM(%3) "\n" ASM(%4) "\n" ASM(%5) "\n" ASM(%6) "\n" ASM(%7)     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));                     \
        }                                                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                      \
    }
#define A_RCP(x) "v_rcp_f64 " #x ", " #x
#define A_RSQ(x) "v_rsq_f64 " #x ", " #x
#define A_SQRT(x) "v_sqrt_f64 " #x ", " #x
#define A_FMA(x) "v_fma_f64 " #x ", " #x ", " #x ", " #x
#define A_MUL(x) "v_mul_f64 " #x ", " #x ", " #x
#define A_ADD(x) "v_add_f64 " #x ", " #x ", " #x
#define A_MOV(x) "v_mov_b64 " #x ", " #x
#define A_MAX(x) "v_max_f64 " #x ", " #x ", " #x
#define A_FMA32(x) "v_fma_f32 " #x ", " #x ", " #x ", " #x
#define A_PKFMA32(x) "v_pk_fma_f32 " #x ", " #x ", " #x ", " #x
#define A_RCP32(x) "v_rcp_f32 " #x ", " #x
#define A_RSQ32(x) "v_rsq_f32 " #x ", " #x
#define A_CVT(x) "v_cvt_f32_f64 " #x ", " #x
#define A_CVTB(x) "v_cvt_f64_f32 " #x ", " #x
#define A_CND(x) "v_cndmask_b32 " #x ", " #x ", " #x ", vcc"
#define A_DPP(x) "v_mov_b32_dpp " #x ", " #x " row_shr:1 row_mask:0xf bank_mask:0xf"
#define A_DPPW(x) "v_mov_b32_dpp " #x ", " #x " wave_shr:1 row_mask:0xf bank_mask:0xf"
#define A_CNDS(x) "v_cndmask_b32_e64 " #x ", " #x ", " #x ", s[8:9]"
#define A_CND0(x) "v_cndmask_b32 " #x ", 0, " #x ", vcc"
#define A_CNDK(x) "v_cndmask_b32_e64 " #x ", 0, " #x ", s[8:9]"
#define A_BFI(x) "v_bfi_b32 " #x ", " #x ", " #x ", " #x
#define A_AND(x) "v_and_b32 " #x ", " #x ", " #x
#define A_MOV32(x) "v_mov_b32 " #x ", " #x
#define A_ADD32(x) "v_add_u32 " #x ", " #x ", " #x
#define A_CMPCND(x) "v_cmp_lt_f32 vcc, 0, " #x "\nv_cndmask_b32 " #x ", " #x ", " #x ", vcc"
#define A_MED3(x) "v_med3_f32 " #x ", " #x ", " #x ", " #x
#define A_DIVS(x) "v_div_scale_f64 " #x ", vcc, " #x ", " #x ", " #x
#define A_DIVF(x) "v_div_fmas_f64 " #x ", " #x ", " #x ", " #x
#define A_DIVX(x) "v_div_fixup_f64 " #x ", " #x ", " #x ", " #x
#define A_LDEXP(x) "v_ldexp_f64 " #x ", " #x ", 1"
#define A_CMP(x) "v_cmp_lt_f64 vcc, " #x ", " #x
#define OP1F(name, ASM)                                                                                  \
    __global__ __launch_bounds__(256) void name(double *out, double seed)                                 \
    {                                                                                                     \
        float a0 = (float) seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < REPS; i++) {                                                                  \
            asm volatile(ASM(%0) "\n" ASM(%1) "\n" ASM(%2) "\n" ASM(%3) "\n" ASM(%4) "\n" ASM(%5) "\n" ASM(%6) "\n" ASM(%7)     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));                     \
        }                                                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                      \
    }
// a conversion pair f64 -> f32 -> f64 (two instructions per chain link; the table halves it)
__global__ __launch_bounds__(256) void k_cvt(double *out, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0, f1, f2, f3, f4, f5, f6, f7;
    for (int i = 0; i < REPS / 2; i++) {
        asm volatile("v_cvt_f32_f64 %8, %0\nv_cvt_f32_f64 %9, %1\nv_cvt_f32_f64 %10, %2\nv_cvt_f32_f64 %11, %3\n"
                     "v_cvt_f32_f64 %12, %4\nv_cvt_f32_f64 %13, %5\nv_cvt_f32_f64 %14, %6\nv_cvt_f32_f64 %15, %7\n"
                     "v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\n"
                     "v_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(f0), "=&v"(f1), "=&v"(f2),
                       "=&v"(f3), "=&v"(f4), "=&v"(f5), "=&v"(f6), "=&v"(f7));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
OP1(k_rcp, A_RCP) OP1(k_rsq, A_RSQ) OP1(k_sqrt, A_SQRT) OP1(k_fma, A_FMA) OP1(k_mul, A_MUL) OP1(k_add, A_ADD) OP1(k_mov, A_MOV)
OP1(k_max, A_MAX) OP1F(k_fma32, A_FMA32) OP1(k_pkfma32, A_PKFMA32) OP1F(k_rcp32, A_RCP32) OP1F(k_rsq32, A_RSQ32) OP1F(k_cnd, A_CND) OP1F(k_dpp, A_DPP) OP1F(k_dppw, A_DPPW) OP1F(k_cnds, A_CNDS) OP1F(k_cnd0, A_CND0) OP1F(k_cndk, A_CNDK)
OP1F(k_bfi, A_BFI) OP1F(k_and, A_AND) OP1F(k_mov32, A_MOV32) OP1F(k_add32, A_ADD32) OP1F(k_cmpcnd, A_CMPCND) OP1F(k_med3, A_MED3) OP1(k_divs, A_DIVS) OP1(k_divf, A_DIVF) OP1(k_divx, A_DIVX)
OP1(k_ldexp, A_LDEXP) OP1(k_cmp, A_CMP)
typedef void (*kern_t)(double *, double);
int main()
{
    double *out;
    hipMalloc(&out, 256 * 16384 * sizeof(double));
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);   // kHz
    struct { const char *name; kern_t k; } ks[] = {
        {"v_fma_f64", k_fma}, {"v_mul_f64", k_mul}, {"v_add_f64", k_add}, {"v_max_f64", k_max}, {"v_mov_b64", k_mov},
        {"v_rcp_f64", k_rcp}, {"v_rsq_f64", k_rsq}, {"v_sqrt_f64", k_sqrt}, {"v_div_scale_f64", k_divs}, {"v_div_fmas_f64", k_divf},
        {"v_div_fixup_f64", k_divx}, {"v_ldexp_f64", k_ldexp}, {"v_cmp_lt_f64", k_cmp}, {"v_cndmask_b32", k_cnd},
        {"v_mov_b32_dpp row_shr", k_dpp}, {"v_mov_b32_dpp wave_shr", k_dppw}, {"v_cndmask_b32 sgpr mask", k_cnds}, {"v_cndmask_b32 0,x,vcc", k_cnd0}, {"v_cndmask_b32 0,x,sgpr", k_cndk},
        {"v_bfi_b32", k_bfi}, {"v_and_b32", k_and}, {"v_mov_b32", k_mov32}, {"v_add_u32", k_add32}, {"v_cmp_f32 + v_cndmask (2 instr)", k_cmpcnd}, {"v_med3_f32", k_med3}, {"v_cvt_f32_f64 / f64_f32", k_cvt}, {"v_fma_f32", k_fma32},
        {"v_pk_fma_f32", k_pkfma32}, {"v_rcp_f32", k_rcp32}, {"v_rsq_f32", k_rsq32}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("clock %d kHz; cycles per wave instruction (8 independent chains per wave) at 1 / 2 / 4 waves per SIMD\n", clk);
    for (auto &k : ks) {
        printf("%-34s", k.name);
        for (int w : {1, 2, 4}) {
            const int blocks = 256 * w;          // 256 CUs x w blocks of 4 waves = w waves per SIMD
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 1.5);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 1.5);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double cyc = (double) ms * 1e-3 * clk * 1e3 / ((double) REPS * 8 * w);
            printf("  %7.2f", cyc);
        }
        printf("\n");
    }
    return 0;
}
