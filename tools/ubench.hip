// VALU issue-cost microbenchmark (development tool): cycles per wave-instruction at full occupancy.
// hipcc --offload-arch=gfx950 -O2 -o ubench tools/ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define KERNEL(name, body)                                                         \
    __global__ __launch_bounds__(256) void name(float* out, int iters, float seed) { \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                        \
        float a = seed + threadIdx.x, b = seed * 1.5f, c = seed * 0.25f, d = a + 1.0f; \
        float e = a + 2.0f, f = a + 3.0f, g = a + 4.0f, h = a + 5.0f;                \
        unsigned ua = threadIdx.x * 2654435761u, ub = 12345u;                       \
        for (int i = 0; i < iters; i++) {                                          \
            REP64(body)                                                            \
        }                                                                          \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (float)(ua + ub);   \
        if (threadIdx.x == 0 && blockIdx.x == 7) ((unsigned long long*)out)[1 << 20] = t1 - t0;      \
    }
// 8 independent chains to avoid dependency stalls within a wave
KERNEL(k_mul, asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f));)
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b));)
KERNEL(k_sqrt, asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b));)
KERNEL(k_mullo, asm volatile("v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2" : "+v"(ua), "+v"(ub) : "v"(2654435761u));)
KERNEL(k_mul24, asm volatile("v_mul_u32_u24 %0, %0, %2\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %0, %0, %2\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %0, %0, %2\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %0, %0, %2\n v_mul_u32_u24 %1, %1, %2" : "+v"(ua), "+v"(ub) : "v"(0x9E3779u));)
KERNEL(k_divscale, asm volatile("v_div_scale_f32 %0, vcc, %0, %8, %0\n v_div_scale_f32 %1, vcc, %1, %8, %1\n v_div_scale_f32 %2, vcc, %2, %8, %2\n v_div_scale_f32 %3, vcc, %3, %8, %3\n v_div_scale_f32 %4, vcc, %4, %8, %4\n v_div_scale_f32 %5, vcc, %5, %8, %5\n v_div_scale_f32 %6, vcc, %6, %8, %6\n v_div_scale_f32 %7, vcc, %7, %8, %7" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.5f) : "vcc");)
KERNEL(k_divfmas, asm volatile("v_div_fmas_f32 %0, %0, %8, %8\n v_div_fmas_f32 %1, %1, %8, %8\n v_div_fmas_f32 %2, %2, %8, %8\n v_div_fmas_f32 %3, %3, %8, %8\n v_div_fmas_f32 %4, %4, %8, %8\n v_div_fmas_f32 %5, %5, %8, %8\n v_div_fmas_f32 %6, %6, %8, %8\n v_div_fmas_f32 %7, %7, %8, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f) : "vcc");)
KERNEL(k_divfixup, asm volatile("v_div_fixup_f32 %0, %0, %8, %8\n v_div_fixup_f32 %1, %1, %8, %8\n v_div_fixup_f32 %2, %2, %8, %8\n v_div_fixup_f32 %3, %3, %8, %8\n v_div_fixup_f32 %4, %4, %8, %8\n v_div_fixup_f32 %5, %5, %8, %8\n v_div_fixup_f32 %6, %6, %8, %8\n v_div_fixup_f32 %7, %7, %8, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f));)
KERNEL(k_cmp_cnd, asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f) : "vcc");)
KERNEL(k_cmp64_cnd, asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cmp_lt_f32 s[22:23], %2, %8\n v_cndmask_b32 %3, %3, %8, s[22:23]\n v_cmp_lt_f32 s[24:25], %4, %8\n v_cndmask_b32 %5, %5, %8, s[24:25]\n v_cmp_lt_f32 s[26:27], %6, %8\n v_cndmask_b32 %7, %7, %8, s[26:27]" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f) : "s20","s21","s22","s23","s24","s25","s26","s27");)
KERNEL(k_minmax, asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f));)
KERNEL(k_pkmul, asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(*(double*)&a), "+v"(*(double*)&c), "+v"(*(double*)&e), "+v"(*(double*)&g) : "v"(1.0));)
KERNEL(k_mul_sgpr, asm volatile("v_mul_f32 %0, s20, %0\n v_mul_f32 %1, s21, %1\n v_mul_f32 %2, s22, %2\n v_mul_f32 %3, s23, %3\n v_mul_f32 %4, s20, %4\n v_mul_f32 %5, s21, %5\n v_mul_f32 %6, s22, %6\n v_mul_f32 %7, s23, %7" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) :: "s20","s21","s22","s23");)
KERNEL(k_fma64, asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(*(double*)&a), "+v"(*(double*)&c), "+v"(*(double*)&e), "+v"(*(double*)&g) : "v"(1.0));)
KERNEL(k_cvt, asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b));)
KERNEL(k_xor_shift, asm volatile("v_lshrrev_b32 %1, 15, %0\n v_xor_b32 %0, %1, %0\n v_lshrrev_b32 %1, 13, %0\n v_xor_b32 %0, %1, %0\n v_lshrrev_b32 %1, 15, %0\n v_xor_b32 %0, %1, %0\n v_lshrrev_b32 %1, 13, %0\n v_xor_b32 %0, %1, %0" : "+v"(ua), "+v"(ub));)


KERNEL(k_add, asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f));)
KERNEL(k_muladd, asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f));)
KERNEL(k_min, asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f));)
KERNEL(k_fmac, asm volatile("v_fmac_f32 %0, %8, %8\n v_fmac_f32 %1, %8, %8\n v_fmac_f32 %2, %8, %8\n v_fmac_f32 %3, %8, %8\n v_fmac_f32 %4, %8, %8\n v_fmac_f32 %5, %8, %8\n v_fmac_f32 %6, %8, %8\n v_fmac_f32 %7, %8, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f));)
KERNEL(k_mul_lit, asm volatile("v_mul_f32 %0, 0x3f800001, %0\n v_mul_f32 %1, 0x3f800001, %1\n v_mul_f32 %2, 0x3f800001, %2\n v_mul_f32 %3, 0x3f800001, %3\n v_mul_f32 %4, 0x3f800001, %4\n v_mul_f32 %5, 0x3f800001, %5\n v_mul_f32 %6, 0x3f800001, %6\n v_mul_f32 %7, 0x3f800001, %7" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b));)
KERNEL(k_mul_inl, asm volatile("v_mul_f32 %0, 1.0, %0\n v_mul_f32 %1, 1.0, %1\n v_mul_f32 %2, 1.0, %2\n v_mul_f32 %3, 1.0, %3\n v_mul_f32 %4, 1.0, %4\n v_mul_f32 %5, 1.0, %5\n v_mul_f32 %6, 1.0, %6\n v_mul_f32 %7, 1.0, %7" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b));)
KERNEL(k_mul2src, asm volatile("v_mul_f32 %0, %1, %2\n v_mul_f32 %1, %2, %3\n v_mul_f32 %2, %3, %4\n v_mul_f32 %3, %4, %5\n v_mul_f32 %4, %5, %6\n v_mul_f32 %5, %6, %7\n v_mul_f32 %6, %7, %0\n v_mul_f32 %7, %0, %1" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b));)
KERNEL(k_sub, asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1e-9f));)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f) : "vcc");)
KERNEL(k_cmp, asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f) : "vcc");)

// round 3: the classes the C3 sample loop's ISA holds besides the ones above (tools/isa_mix.py prices a loop with this table)
KERNEL(k_add_e64abs, asm volatile("v_add_f32_e64 %0, |%0|, |%8|\n v_add_f32_e64 %1, |%1|, |%8|\n v_add_f32_e64 %2, |%2|, |%8|\n v_add_f32_e64 %3, |%3|, |%8|\n v_add_f32_e64 %4, |%4|, |%8|\n v_add_f32_e64 %5, |%5|, |%8|\n v_add_f32_e64 %6, |%6|, |%8|\n v_add_f32_e64 %7, |%7|, |%8|" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f));)
KERNEL(k_sub_e64neg, asm volatile("v_sub_f32_e64 %0, -%0, %8\n v_sub_f32_e64 %1, -%1, %8\n v_sub_f32_e64 %2, -%2, %8\n v_sub_f32_e64 %3, -%3, %8\n v_sub_f32_e64 %4, -%4, %8\n v_sub_f32_e64 %5, -%5, %8\n v_sub_f32_e64 %6, -%6, %8\n v_sub_f32_e64 %7, -%7, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1e-9f));)
KERNEL(k_xor_sdwa, asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %1, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %1, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %1, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %1, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(ua), "+v"(ub));)
KERNEL(k_alignbit, asm volatile("v_alignbit_b32 %0, %2, %0, 9\n v_alignbit_b32 %1, %2, %1, 9\n v_alignbit_b32 %0, %2, %0, 9\n v_alignbit_b32 %1, %2, %1, 9\n v_alignbit_b32 %0, %2, %0, 9\n v_alignbit_b32 %1, %2, %1, 9\n v_alignbit_b32 %0, %2, %0, 9\n v_alignbit_b32 %1, %2, %1, 9" : "+v"(ua), "+v"(ub) : "v"(0x7fu));)
KERNEL(k_cmp_e64, asm volatile("v_cmp_ge_f32_e64 s[20:21], |%0|, %8\n v_cmp_ge_f32_e64 s[22:23], |%1|, %8\n v_cmp_ge_f32_e64 s[24:25], |%2|, %8\n v_cmp_ge_f32_e64 s[26:27], |%3|, %8\n v_cmp_ge_f32_e64 s[20:21], |%4|, %8\n v_cmp_ge_f32_e64 s[22:23], |%5|, %8\n v_cmp_ge_f32_e64 s[24:25], |%6|, %8\n v_cmp_ge_f32_e64 s[26:27], |%7|, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f) : "s20","s21","s22","s23","s24","s25","s26","s27");)
KERNEL(k_add_u32, asm volatile("v_add_u32 %0, 0x85ebca6b, %0\n v_add_u32 %1, 0x85ebca6b, %1\n v_add_u32 %0, 0x85ebca6b, %0\n v_add_u32 %1, 0x85ebca6b, %1\n v_add_u32 %0, 0x85ebca6b, %0\n v_add_u32 %1, 0x85ebca6b, %1\n v_add_u32 %0, 0x85ebca6b, %0\n v_add_u32 %1, 0x85ebca6b, %1" : "+v"(ua), "+v"(ub));)
KERNEL(k_and_b32, asm volatile("v_and_b32 %0, 0x7fffffff, %0\n v_and_b32 %1, 0x7fffffff, %1\n v_and_b32 %0, 0x7fffffff, %0\n v_and_b32 %1, 0x7fffffff, %1\n v_and_b32 %0, 0x7fffffff, %0\n v_and_b32 %1, 0x7fffffff, %1\n v_and_b32 %0, 0x7fffffff, %0\n v_and_b32 %1, 0x7fffffff, %1" : "+v"(ua), "+v"(ub));)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b));)
KERNEL(k_add_sgpr, asm volatile("v_add_f32 %0, s20, %0\n v_add_f32 %1, s21, %1\n v_add_f32 %2, s22, %2\n v_add_f32 %3, s23, %3\n v_add_f32 %4, s20, %4\n v_add_f32 %5, s21, %5\n v_add_f32 %6, s22, %6\n v_add_f32 %7, s23, %7" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) :: "s20","s21","s22","s23");)
KERNEL(k_mullo_sgpr, asm volatile("v_mul_lo_u32 %0, %0, s20\n v_mul_lo_u32 %1, %1, s21\n v_mul_lo_u32 %0, %0, s20\n v_mul_lo_u32 %1, %1, s21\n v_mul_lo_u32 %0, %0, s20\n v_mul_lo_u32 %1, %1, s21\n v_mul_lo_u32 %0, %0, s20\n v_mul_lo_u32 %1, %1, s21" : "+v"(ua), "+v"(ub) :: "s20","s21");)
KERNEL(k_fmamk, asm volatile("v_mul_f32 %0, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_mul_f32 %2, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_mul_f32 %4, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_mul_f32 %6, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f) : "vcc");)

template <class K> void run(const char* name, K kern, int waves_per_simd) {
    int dev; hipGetDevice(&dev); hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    int cus = p.multiProcessorCount;
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
    float* out; hipMalloc(&out, (size_t)(1 << 23) + 64);
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts_per_simd = (double)iters * 64 * 8 * waves_per_simd;  // wave-instructions issued on one SIMD
    double clk = p.clockRate * 1e3;                                     // Hz (nominal)
    unsigned long long dt = 0; hipMemcpy(&dt, (char*)out + (size_t)(1 << 20) * 8, 8, hipMemcpyDeviceToHost);
    double wave_insts = (double)iters * 64 * 8;
    printf("%-14s waves/SIMD=%d  %.3f ms  nominal %.2f cyc/inst | s_memtime: %.2f SIMD-cycles per wave-inst, clock %.2f GHz\n", name, waves_per_simd, ms,
           ms * 1e-3 * clk / insts_per_simd, (double)dt / wave_insts / waves_per_simd, (double)dt / (ms * 1e-3) / 1e9);
    hipFree(out);
}
int main() {
    for (int w : {1, 4}) {
        run("v_add_f32", k_add, w); run("mul/add alt", k_muladd, w); run("v_min only", k_min, w); run("v_fmac_f32", k_fmac, w); run("v_mul literal", k_mul_lit, w);
        run("v_mul inline", k_mul_inl, w); run("v_mul 2src", k_mul2src, w); run("v_sub_f32", k_sub, w); run("v_cndmask", k_cndmask, w); run("v_cmp", k_cmp, w);
        run("v_mul_f32", k_mul, w); run("v_fma_f32", k_fma, w); run("v_mul s,v", k_mul_sgpr, w); run("v_rcp_f32", k_rcp, w); run("v_sqrt_f32", k_sqrt, w);
        run("v_mul_lo_u32", k_mullo, w); run("v_mul_u32_u24", k_mul24, w); run("v_div_scale", k_divscale, w); run("v_div_fmas", k_divfmas, w);
        run("v_div_fixup", k_divfixup, w); run("cmp+cndmask", k_cmp_cnd, w); run("cmp64+cnd", k_cmp64_cnd, w); run("min/max", k_minmax, w);
        run("v_pk_mul_f32", k_pkmul, w); run("v_fma_f64", k_fma64, w); run("v_cvt_f32_u32", k_cvt, w); run("shift+xor", k_xor_shift, w);
        run("add_e64 |abs|", k_add_e64abs, w); run("sub_e64 -neg", k_sub_e64neg, w); run("xor_sdwa", k_xor_sdwa, w); run("v_alignbit", k_alignbit, w);
        run("v_cmp_e64 sgpr", k_cmp_e64, w); run("v_add_u32 lit", k_add_u32, w); run("v_and_b32 lit", k_and_b32, w); run("v_mov_b32", k_mov, w);
        run("v_add s,v", k_add_sgpr, w); run("v_mul_lo s", k_mullo_sgpr, w); run("mul/cmp alt", k_fmamk, w);
        printf("\n");
    }
    return 0;
}
