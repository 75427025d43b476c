// VALU issue-rate micro-bench for gfx950 (developer tool): cycles per wave64 instruction per SIMD
// for the op classes the stacking / warp kernels are made of.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_rate.hip -o build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 64
#define ITER 2048

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (OP == 0) {  // v_min_f32 / v_max_f32 (8 independent chains)
                asm volatile("v_min_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_max_f32 %3, %3, %4\n"
                             "v_min_f32 %4, %4, %5\n v_max_f32 %5, %5, %6\n v_min_f32 %6, %6, %7\n v_max_f32 %7, %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 1) {  // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n"
                             "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 2) {  // v_add_f64
                asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %4\n"
                             "v_add_f64 %4, %4, %5\n v_add_f64 %5, %5, %6\n v_add_f64 %6, %6, %7\n v_add_f64 %7, %7, %0\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
            } else if (OP == 3) {  // v_mul_f64
                asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %4\n"
                             "v_mul_f64 %4, %4, %5\n v_mul_f64 %5, %5, %6\n v_mul_f64 %6, %6, %7\n v_mul_f64 %7, %7, %0\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
            } else if (OP == 4) {  // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %3, %3, %4, %5\n"
                             "v_fma_f64 %4, %4, %5, %6\n v_fma_f64 %5, %5, %6, %7\n v_fma_f64 %6, %6, %7, %0\n v_fma_f64 %7, %7, %0, %1\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
            } else if (OP == 5) {  // v_cvt_f64_f32
                asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n"
                             "v_cvt_f64_f32 %4, %12\n v_cvt_f64_f32 %5, %13\n v_cvt_f64_f32 %6, %14\n v_cvt_f64_f32 %7, %15\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            } else if (OP == 6) {  // v_cndmask_b32 with vcc
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                             "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (OP == 7) {  // v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %6\n v_pk_add_f32 %6, %6, %7\n v_pk_add_f32 %7, %7, %0\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
            } else if (OP == 8) {  // v_min3_f32 / v_med3_f32 / v_max3_f32
                asm volatile("v_min3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_min3_f32 %3, %3, %4, %5\n"
                             "v_med3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_min3_f32 %6, %6, %7, %0\n v_med3_f32 %7, %7, %0, %1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 9) {  // v_sub_f32
                asm volatile("v_sub_f32 %0, %0, %1\n v_sub_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_sub_f32 %3, %3, %4\n"
                             "v_sub_f32 %4, %4, %5\n v_sub_f32 %5, %5, %6\n v_sub_f32 %6, %6, %7\n v_sub_f32 %7, %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 10) {  // v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %4\n"
                             "v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %6\n v_pk_mul_f32 %6, %6, %7\n v_pk_mul_f32 %7, %7, %0\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
            } else if (OP >= 20 && OP < 40) {
#define CHAIN2(OPN) asm volatile(OPN " %0, %0, %1\n " OPN " %1, %1, %2\n " OPN " %2, %2, %3\n " OPN " %3, %3, %4\n " \
                                 OPN " %4, %4, %5\n " OPN " %5, %5, %6\n " OPN " %6, %6, %7\n " OPN " %7, %7, %0\n"     \
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
                if (OP == 20) CHAIN2("v_min_i32");
                if (OP == 21) CHAIN2("v_max_u32");
                if (OP == 22) CHAIN2("v_add_u32");
                if (OP == 23) CHAIN2("v_pk_min_i16");
                if (OP == 24) CHAIN2("v_pk_max_u16");
                if (OP == 25) CHAIN2("v_and_b32");
                if (OP == 26) CHAIN2("v_min_f16");
                if (OP == 27) CHAIN2("v_pk_min_f16");
                if (OP == 28) CHAIN2("v_add_f32");
                if (OP == 29) CHAIN2("v_mul_f32");
                if (OP == 30) CHAIN2("v_xor_b32");
                if (OP == 31) CHAIN2("v_lshlrev_b32");
                if (OP == 32) CHAIN2("v_sub_u32");
                if (OP == 33) CHAIN2("v_max_i16");
            } else if (OP >= 50 && OP < 70) {
#define CHAIN2V(OPN) asm volatile(OPN " %0, vcc, %0, %1\n " OPN " %1, vcc, %1, %2\n " OPN " %2, vcc, %2, %3\n " OPN " %3, vcc, %3, %4\n " \
                                 OPN " %4, vcc, %4, %5\n " OPN " %5, vcc, %5, %6\n " OPN " %6, vcc, %6, %7\n " OPN " %7, vcc, %7, %0\n"     \
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc")
#define CHAINC(OPN) asm volatile(OPN " vcc, %0, %1\n " OPN " vcc, %1, %2\n " OPN " vcc, %2, %3\n " OPN " vcc, %3, %4\n " \
                                 OPN " vcc, %4, %5\n " OPN " vcc, %5, %6\n " OPN " vcc, %6, %7\n " OPN " vcc, %7, %0\n"     \
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc")
#define CHAIN3(OPN) asm volatile(OPN " %0, %0, %1, %2\n " OPN " %1, %1, %2, %3\n " OPN " %2, %2, %3, %4\n " OPN " %3, %3, %4, %5\n " \
                                 OPN " %4, %4, %5, %6\n " OPN " %5, %5, %6, %7\n " OPN " %6, %6, %7, %0\n " OPN " %7, %7, %0, %1\n"     \
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
                if (OP == 50) CHAIN2V("v_sub_co_u32");
                if (OP == 51) CHAINC("v_cmp_lt_u32");
                if (OP == 52) CHAINC("v_cmp_lt_i32");
                if (OP == 53) CHAIN2("v_min_u32");
                if (OP == 54) CHAIN2("v_max_i32");
                if (OP == 55) CHAIN3("v_perm_b32");
                if (OP == 56) CHAIN3("v_add3_u32");
                if (OP == 57) CHAIN3("v_lshl_add_u32");
                if (OP == 58) CHAIN3("v_and_or_b32");
                if (OP == 59) CHAIN3("v_bfe_u32");
                if (OP == 60) CHAIN3("v_minimum3_f32");
                if (OP == 61) CHAIN3("v_mad_u32_u24");
                                if (OP == 63) CHAIN3("v_alignbit_b32");
                if (OP == 64) CHAIN3("v_sad_u32");
                if (OP == 65) CHAIN3("v_max3_u32");
                if (OP == 66) CHAIN2("v_max_u16");
                if (OP == 67) CHAIN3("v_med3_f32");
            } else if (OP >= 80 && OP < 100) {
#define CHAIN3S(OPN, SUF) asm volatile(OPN " %0, %0, %1, %2" SUF "\n " OPN " %1, %1, %2, %3" SUF "\n " OPN " %2, %2, %3, %4" SUF "\n " OPN " %3, %3, %4, %5" SUF "\n " \
                                 OPN " %4, %4, %5, %6" SUF "\n " OPN " %5, %5, %6, %7" SUF "\n " OPN " %6, %6, %7, %0" SUF "\n " OPN " %7, %7, %0, %1" SUF "\n"     \
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define CHAIN2D(OPN) asm volatile(OPN " %0, %0, %1\n " OPN " %1, %1, %2\n " OPN " %2, %2, %3\n " OPN " %3, %3, %4\n " \
                                 OPN " %4, %4, %5\n " OPN " %5, %5, %6\n " OPN " %6, %6, %7\n " OPN " %7, %7, %0\n"     \
                                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7))
#define CHAIN1D(OPN) asm volatile(OPN " %0, %1\n " OPN " %1, %2\n " OPN " %2, %3\n " OPN " %3, %4\n " \
                                 OPN " %4, %5\n " OPN " %5, %6\n " OPN " %6, %7\n " OPN " %7, %0\n"     \
                                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7))
#define CHAIN1(OPN) asm volatile(OPN " %0, %1\n " OPN " %1, %2\n " OPN " %2, %3\n " OPN " %3, %4\n " \
                                 OPN " %4, %5\n " OPN " %5, %6\n " OPN " %6, %7\n " OPN " %7, %0\n"     \
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define CHAIN3D(OPN) asm volatile(OPN " %0, %0, %1, %2\n " OPN " %1, %1, %2, %3\n " OPN " %2, %2, %3, %4\n " OPN " %3, %3, %4, %5\n " \
                                 OPN " %4, %4, %5, %6\n " OPN " %5, %5, %6, %7\n " OPN " %6, %6, %7, %0\n " OPN " %7, %7, %0, %1\n"     \
                                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7))
                if (OP == 80) CHAIN3S("v_bitop3_b32", " bitop3:0x96");
                if (OP == 81) CHAIN2D("v_min_f64");
                if (OP == 82) CHAIN1D("v_rcp_f64");
                if (OP == 83) CHAIN1D("v_rsq_f64");
                if (OP == 84) CHAIN1D("v_sqrt_f64");
                if (OP == 85) CHAIN3D("v_div_fixup_f64");
                if (OP == 86) CHAIN1("v_mov_b32");
                if (OP == 87) CHAIN1("v_rcp_f32");
                if (OP == 88) CHAIN1("v_sqrt_f32");
                if (OP == 89) {  // the compare-exchange as the sorter would issue it: v_min_f32 + v_bitop3_b32 (a ^ b ^ min)  (4 CEs = 8 instr)
                    float t0, t1, t2, t3;
                    asm volatile("v_min_f32 %8, %0, %1\n v_min_f32 %9, %2, %3\n v_min_f32 %10, %4, %5\n v_min_f32 %11, %6, %7\n"
                                 "v_bitop3_b32 %1, %0, %1, %8 bitop3:0x96\n v_bitop3_b32 %3, %2, %3, %9 bitop3:0x96\n"
                                 "v_bitop3_b32 %5, %4, %5, %10 bitop3:0x96\n v_bitop3_b32 %7, %6, %7, %11 bitop3:0x96\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
                }
                if (OP == 90) {  // the same with the two-input xor twice (12 instr per 4 CEs; normalised to 8 by the caller's REP)
                    float t0, t1, t2, t3;
                    asm volatile("v_min_f32 %8, %0, %1\n v_min_f32 %9, %2, %3\n v_xor_b32 %1, %0, %1\n v_xor_b32 %3, %2, %3\n"
                                 "v_xor_b32 %1, %1, %8\n v_xor_b32 %3, %3, %9\n v_mov_b32 %0, %8\n v_mov_b32 %2, %9\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
                }
                if (OP == 91)
                    asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n"
                                 "v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                 : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
                if (OP == 92) CHAIN3D("v_pk_fma_f32");
                if (OP == 93) {  // v_cmp -> sgpr pair, v_addc with that carry-in (per-lane counting: 2 instr per element)
                    unsigned long long m0, m1, m2, m3;
                    asm volatile("v_cmp_lt_f32 %8, %0, %1\n v_cmp_lt_f32 %9, %1, %2\n v_cmp_lt_f32 %10, %2, %3\n v_cmp_lt_f32 %11, %3, %0\n"
                                 "v_addc_co_u32 %4, vcc, 0, %4, %8\n v_addc_co_u32 %5, vcc, 0, %5, %9\n v_addc_co_u32 %6, vcc, 0, %6, %10\n v_addc_co_u32 %7, vcc, 0, %7, %11\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : : "vcc");
                }
            } else if (OP == 70) {  // integer compare-exchange: v_sub_co_u32 -> vcc ; 2 x v_cndmask (3 instr per CE, 4 CEs)
                float t0, t1, t2, t3;
                asm volatile("v_sub_co_u32 %8, vcc, %0, %1\n v_cndmask_b32 %8, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_mov_b32 %0, %8\n"
                             "v_sub_co_u32 %9, vcc, %2, %3\n v_cndmask_b32 %9, %2, %3, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n v_mov_b32 %2, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : : "vcc");
            } else if (OP == 71) {  // CE through an SGPR pair (no vcc serialisation): v_cmp_lt_u32 s[..], 2 cndmask
                float t0, t1; unsigned long long m0, m1;
                asm volatile("v_cmp_lt_u32 %10, %0, %1\n v_cmp_lt_u32 %11, %2, %3\n v_cndmask_b32 %8, %0, %1, %10\n v_cndmask_b32 %1, %1, %0, %10\n"
                             "v_cndmask_b32 %9, %2, %3, %11\n v_cndmask_b32 %3, %3, %2, %11\n v_mov_b32 %0, %8\n v_mov_b32 %2, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1), "=&s"(m0), "=&s"(m1));
            } else if (OP == 72) {  // v_sub_co into sgpr pair then cndmask
                float t0, t1, u0, u1; unsigned long long m0, m1;
                asm volatile("v_sub_co_u32 %12, %10, %0, %1\n v_sub_co_u32 %13, %11, %2, %3\n v_cndmask_b32 %8, %0, %1, %10\n v_cndmask_b32 %1, %1, %0, %10\n"
                             "v_cndmask_b32 %9, %2, %3, %11\n v_cndmask_b32 %3, %3, %2, %11\n v_mov_b32 %0, %8\n v_mov_b32 %2, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1), "=&s"(m0), "=&s"(m1), "=&v"(u0), "=&v"(u1));
            } else if (OP == 40) {  // v_cmp_lt_f32 (vcc) only
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n"
                             "v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (OP == 41) {  // v_cndmask_b32 reading a constant vcc
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n"
                             "v_cndmask_b32 %4, %5, %6, vcc\n v_cndmask_b32 %5, %6, %7, vcc\n v_cndmask_b32 %6, %7, %0, vcc\n v_cndmask_b32 %7, %0, %1, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 42) {  // v_med3_i32 / v_min3_i32 / v_max3_i32
                asm volatile("v_min3_i32 %0, %0, %1, %2\n v_med3_i32 %1, %1, %2, %3\n v_max3_i32 %2, %2, %3, %4\n v_min3_i32 %3, %3, %4, %5\n"
                             "v_med3_i32 %4, %4, %5, %6\n v_max3_i32 %5, %5, %6, %7\n v_min3_i32 %6, %6, %7, %0\n v_med3_i32 %7, %7, %0, %1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 11) {  // v_cmp_lt_f32 -> vcc ; v_cndmask (dependent pair, as the compiler emits it)
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %5\n s_nop 1\n v_cndmask_b32 %6, %6, %7, vcc\n"
                             "v_cmp_lt_f32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %3, %3, %4, vcc\n v_cmp_lt_f32 vcc, %5, %6\n s_nop 1\n v_cndmask_b32 %7, %7, %0, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

template <int OP>
void run(const char *name, float *out, int blocks_per_cu, int cus, double clk_ghz) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * blocks_per_cu;
    k<OP><<<grid, 256>>>(out, 1.0f);
    hipEventRecord(e0);
    k<OP><<<grid, 256>>>(out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: blocks_per_cu waves (one wave of each block per SIMD), each REP*ITER instructions
    const double instr_per_simd = (double)blocks_per_cu * REP * ITER * (OP == 11 ? 0.75 : 1.0);
    const double cycles = ms * 1e-3 * clk_ghz * 1e9;
    printf("%-34s %2d waves/SIMD  %8.3f ms  %.2f cycles/instr (at %.2f GHz)\n", name, blocks_per_cu, ms, cycles / instr_per_simd, clk_ghz);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double clk = p.clockRate * 1e-6;
    float *out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    printf("%s: %d CUs, clock %.2f GHz\n", p.gcnArchName, cus, clk);
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("v_min/max_f32", out, 1, cus, clk); run<1>("v_fma_f32", out, 1, cus, clk); run<2>("v_add_f64", out, 1, cus, clk); }
        if (w == 2) { run<0>("v_min/max_f32", out, 2, cus, clk); run<1>("v_fma_f32", out, 2, cus, clk); run<2>("v_add_f64", out, 2, cus, clk); }
        if (w == 4) {
            run<0>("v_min/max_f32", out, 4, cus, clk); run<1>("v_fma_f32", out, 4, cus, clk); run<9>("v_sub_f32", out, 4, cus, clk);
            run<8>("v_min3/med3/max3_f32", out, 4, cus, clk); run<6>("v_cndmask_b32 (vcc)", out, 4, cus, clk);
            run<11>("v_cmp+s_nop1+v_cndmask (per instr)", out, 4, cus, clk);
            run<7>("v_pk_add_f32", out, 4, cus, clk); run<10>("v_pk_mul_f32", out, 4, cus, clk);
            run<2>("v_add_f64", out, 4, cus, clk); run<3>("v_mul_f64", out, 4, cus, clk); run<4>("v_fma_f64", out, 4, cus, clk);
            run<5>("v_cvt_f64_f32", out, 4, cus, clk);
            run<20>("v_min_i32", out, 4, cus, clk); run<21>("v_max_u32", out, 4, cus, clk); run<22>("v_add_u32", out, 4, cus, clk);
            run<32>("v_sub_u32", out, 4, cus, clk); run<25>("v_and_b32", out, 4, cus, clk); run<30>("v_xor_b32", out, 4, cus, clk);
            run<31>("v_lshlrev_b32", out, 4, cus, clk); run<23>("v_pk_min_i16", out, 4, cus, clk); run<24>("v_pk_max_u16", out, 4, cus, clk);
            run<33>("v_max_i16", out, 4, cus, clk); run<26>("v_min_f16", out, 4, cus, clk); run<27>("v_pk_min_f16", out, 4, cus, clk);
            run<28>("v_add_f32", out, 4, cus, clk); run<29>("v_mul_f32", out, 4, cus, clk);
            run<40>("v_cmp_lt_f32 -> vcc", out, 4, cus, clk); run<41>("v_cndmask_b32 (const vcc)", out, 4, cus, clk);
            run<42>("v_min3/med3/max3_i32", out, 4, cus, clk);
            run<50>("v_sub_co_u32 ->vcc", out, 4, cus, clk); run<51>("v_cmp_lt_u32 ->vcc", out, 4, cus, clk); run<52>("v_cmp_lt_i32 ->vcc", out, 4, cus, clk);
            run<53>("v_min_u32", out, 4, cus, clk); run<54>("v_max_i32", out, 4, cus, clk); run<55>("v_perm_b32", out, 4, cus, clk);
            run<56>("v_add3_u32", out, 4, cus, clk); run<57>("v_lshl_add_u32", out, 4, cus, clk); run<58>("v_and_or_b32", out, 4, cus, clk);
            run<59>("v_bfe_u32", out, 4, cus, clk); run<60>("v_minimum3_f32", out, 4, cus, clk); run<61>("v_mad_u32_u24", out, 4, cus, clk);
            run<63>("v_alignbit_b32", out, 4, cus, clk); run<64>("v_sad_u32", out, 4, cus, clk); run<65>("v_max3_u32", out, 4, cus, clk);
            run<66>("v_max_u16", out, 4, cus, clk); run<67>("v_med3_f32", out, 4, cus, clk);
            run<70>("CE int: sub_co+2cndmask+mov (8 instr/blk)", out, 4, cus, clk); run<71>("CE: cmp_u32->sgpr,2cndmask (8/blk)", out, 4, cus, clk);
            run<72>("CE: sub_co->sgpr,2cndmask (8/blk)", out, 4, cus, clk);
            run<80>("v_bitop3_b32", out, 4, cus, clk); run<81>("v_min_f64", out, 4, cus, clk); run<82>("v_rcp_f64", out, 4, cus, clk);
            run<83>("v_rsq_f64", out, 4, cus, clk); run<84>("v_sqrt_f64", out, 4, cus, clk); run<85>("v_div_fixup_f64", out, 4, cus, clk);
            run<86>("v_mov_b32", out, 4, cus, clk); run<87>("v_rcp_f32", out, 4, cus, clk); run<88>("v_sqrt_f32", out, 4, cus, clk);
            run<89>("CE: v_min_f32 + v_bitop3 (8/blk)", out, 4, cus, clk); run<90>("CE: v_min_f32 + 2 v_xor + mov (8/blk)", out, 4, cus, clk);
            run<91>("v_cvt_f32_f64", out, 4, cus, clk); run<92>("v_pk_fma_f32", out, 4, cus, clk);
            run<93>("count: v_cmp->sgpr + v_addc (8/blk)", out, 4, cus, clk);
        }
    }
    return 0;
}
