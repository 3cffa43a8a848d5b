// Round-3 VALU issue-rate micro-benchmark for gfx950 (companion of valu_ubench.hip): packed-f16 arithmetic, the gfx950 three-operand
// packed min / max, SDWA byte selects, carry-in mask assembly and the conversions a 16-bit slab test would be made of.
// Same method: 256-thread blocks, W waves per SIMD, 4 independent chains per lane, 64 x 4 instructions per loop iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
typedef float f2 __attribute__((ext_vector_type(2)));

#define KINDS(X) \
  X(0,  "v_fma_f32 (reference)",        "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9") \
  X(1,  "v_pk_fma_f16",                 "v_pk_fma_f16 %4, %4, %10, %11\n v_pk_fma_f16 %5, %5, %10, %11\n v_pk_fma_f16 %6, %6, %10, %11\n v_pk_fma_f16 %7, %7, %10, %11") \
  X(2,  "v_pk_max/min_f16",             "v_pk_max_f16 %4, %4, %10\n v_pk_min_f16 %5, %5, %11\n v_pk_max_f16 %6, %6, %10\n v_pk_min_f16 %7, %7, %11") \
  X(3,  "v_pk_maximum3/minimum3_f16",   "v_pk_maximum3_f16 %4, %4, %10, %11\n v_pk_minimum3_f16 %5, %5, %10, %11\n v_pk_maximum3_f16 %6, %6, %10, %11\n v_pk_minimum3_f16 %7, %7, %10, %11") \
  X(4,  "v_pk_add_f16 (neg)",           "v_pk_add_f16 %4, %4, %10 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f16 %5, %5, %11\n v_pk_add_f16 %6, %6, %10 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f16 %7, %7, %11") \
  X(5,  "v_pk_mul_f16",                 "v_pk_mul_f16 %4, %4, %10\n v_pk_mul_f16 %5, %5, %11\n v_pk_mul_f16 %6, %6, %10\n v_pk_mul_f16 %7, %7, %11") \
  X(6,  "v_perm_b32",                   "v_perm_b32 %4, %10, %11, %5\n v_perm_b32 %5, %10, %11, %6\n v_perm_b32 %6, %10, %11, %7\n v_perm_b32 %7, %10, %11, %4") \
  X(7,  "v_or_b32_sdwa (byte sel)",     "v_or_b32_sdwa %4, %10, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_or_b32_sdwa %5, %10, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_or_b32_sdwa %6, %10, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_or_b32_sdwa %7, %10, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0") \
  X(8,  "v_add_f32_sdwa (word sel)",    "v_add_f32_sdwa %0, %0, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_f32_sdwa %1, %1, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_add_f32_sdwa %2, %2, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_f32_sdwa %3, %3, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0") \
  X(9,  "v_or_b32 (plain)",             "v_or_b32 %4, %10, %4\n v_or_b32 %5, %10, %5\n v_or_b32 %6, %11, %6\n v_or_b32 %7, %11, %7") \
  X(10, "v_cvt_pk_f32_fp8",             "v_cvt_pk_f32_fp8 %12, %4\n v_cvt_pk_f32_fp8 %13, %5\n v_cvt_pk_f32_fp8 %14, %6\n v_cvt_pk_f32_fp8 %15, %7") \
  X(11, "v_cvt_pkrtz_f16_f32",          "v_cvt_pkrtz_f16_f32 %4, %0, %1\n v_cvt_pkrtz_f16_f32 %5, %1, %2\n v_cvt_pkrtz_f16_f32 %6, %2, %3\n v_cvt_pkrtz_f16_f32 %7, %3, %0") \
  X(12, "v_cvt_f16_f32",                "v_cvt_f16_f32 %4, %0\n v_cvt_f16_f32 %5, %1\n v_cvt_f16_f32 %6, %2\n v_cvt_f16_f32 %7, %3") \
  X(13, "v_addc_co_u32 (vcc in/out)",   "v_addc_co_u32 %4, vcc, %4, %4, vcc\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n v_addc_co_u32 %6, vcc, %6, %6, vcc\n v_addc_co_u32 %7, vcc, %7, %7, vcc") \
  X(14, "v_cmp_le_f32 (vcc only)",      "v_cmp_le_f32 vcc, %0, %8\n v_cmp_le_f32 vcc, %1, %8\n v_cmp_le_f32 vcc, %2, %9\n v_cmp_le_f32 vcc, %3, %9") \
  X(15, "v_cmp_le_f32 + v_addc_co_u32", "v_cmp_le_f32 vcc, %0, %8\n v_addc_co_u32 %4, vcc, %4, %4, vcc\n v_cmp_le_f32 vcc, %1, %9\n v_addc_co_u32 %5, vcc, %5, %5, vcc") \
  X(16, "v_med3_f32",                   "v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9") \
  X(17, "v_maximum3/minimum3_f32",      "v_maximum3_f32 %0, %0, %8, %9\n v_minimum3_f32 %1, %1, %8, %9\n v_maximum3_f32 %2, %2, %8, %9\n v_minimum3_f32 %3, %3, %8, %9") \
  X(18, "v_cndmask_b32 (vcc)",          "v_cndmask_b32 %4, %4, %10, vcc\n v_cndmask_b32 %5, %5, %10, vcc\n v_cndmask_b32 %6, %6, %11, vcc\n v_cndmask_b32 %7, %7, %11, vcc") \
  X(19, "v_mov_b32_dpp quad_perm",      "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
  X(20, "v_fma_mix_f32 (f16 src0)",     "v_fma_mix_f32 %0, %10, %8, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %10, %8, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %11, %8, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %11, %8, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]") \
  X(21, "v_lshlrev_b32",                "v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7") \
  X(22, "v_and_b32",                    "v_and_b32 %4, %10, %4\n v_and_b32 %5, %10, %5\n v_and_b32 %6, %11, %6\n v_and_b32 %7, %11, %7") \
  X(23, "v_bfi_b32",                    "v_bfi_b32 %4, %10, %4, %11\n v_bfi_b32 %5, %10, %5, %11\n v_bfi_b32 %6, %10, %6, %11\n v_bfi_b32 %7, %10, %7, %11") \
  X(24, "v_cvt_f32_ubyteN (reference)", "v_cvt_f32_ubyte1 %0, %4\n v_cvt_f32_ubyte2 %1, %5\n v_cvt_f32_ubyte3 %2, %6\n v_cvt_f32_ubyte0 %3, %7") \
  X(25, "v_max_f32 e32 (VOP2)",         "v_max_f32_e32 %0, %8, %0\n v_min_f32_e32 %1, %9, %1\n v_max_f32_e32 %2, %8, %2\n v_min_f32_e32 %3, %9, %3") \
  X(26, "v_sub_f32 + v_ashrrev (sign)", "v_sub_f32 %0, %0, %8\n v_ashrrev_i32 %4, 31, %0\n v_sub_f32 %1, %1, %9\n v_ashrrev_i32 %5, 31, %1") \
  X(27, "v_pk_fma_f16 op_sel bcast",    "v_pk_fma_f16 %4, %4, %10, %11 op_sel_hi:[1,0,0]\n v_pk_fma_f16 %5, %5, %10, %11 op_sel_hi:[1,0,0]\n v_pk_fma_f16 %6, %6, %10, %11 op_sel:[0,1,1] op_sel_hi:[1,1,1]\n v_pk_fma_f16 %7, %7, %10, %11 op_sel:[0,1,1] op_sel_hi:[1,1,1]") \
  X(28, "v_bfe_u32",                    "v_bfe_u32 %4, %10, 8, 8\n v_bfe_u32 %5, %11, 16, 8\n v_bfe_u32 %6, %10, 24, 8\n v_bfe_u32 %7, %11, 0, 8") \
  X(29, "v_mul_f32 (reference)",        "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %9")

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a0 = threadIdx.x * seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = seed * 1.0001f, c = seed * 0.5f;
    uint32_t u0 = 0x3c003800u + threadIdx.x, u1 = u0 * 3u, u2 = u0 * 5u, u3 = u0 * 7u, ub = 0x3c003c00u, uc = 0x38003a00u;
    f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a1, a2 }, p3 = { a3, a0 };
    for (int i = 0; i < iters; ++i) {
#define X(ID, NAME, ASM) if (KIND == ID) { REP64(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(b), "v"(c), "v"(ub), "v"(uc), "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "vcc");) }
        KINDS(X)
#undef X
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p3.y + __uint_as_float(u0 ^ u1 ^ u2 ^ u3);
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
    float *out; CHECK(hipMalloc(&out, (size_t) cus * 8 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int waves_per_simd : { 2, 6 }) {
        int grid = cus * waves_per_simd;
        const int iters = 200;
#define X(ID, NAME, ASM) { \
            hipLaunchKernelGGL(k<ID>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); CHECK(hipDeviceSynchronize()); \
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k<ID>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); \
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); \
            double instr_per_simd = (double) waves_per_simd * iters * 64 * 4; \
            printf("%d waves/SIMD  %-30s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.0f MHz)\n", waves_per_simd, NAME, ms, ms * 1e-3 * clk / instr_per_simd, clk / 1e6); }
        KINDS(X)
#undef X
    }
    return 0;
}
