// VALU issue-rate micro-benchmark for gfx950: cycles per wave64 instruction per SIMD for the
// instruction kinds the traversal kernel is made of.  8 waves per SIMD (2 blocks of 1024 threads per CU
// would also do; here 256-thread blocks, 8 per CU), 4 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a0 = threadIdx.x * seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = seed * 1.0001f, c = seed * 0.5f;
    uint32_t u0 = __float_as_uint(a0), u1 = u0 * 3u, u2 = u0 * 5u, u3 = u0 * 7u;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a1, a2 }, p3 = { a3, a0 }, pb = { b, b }, pc = { c, c };
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (KIND == 1) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));) }
        if (KIND == 2) { REP64(asm volatile("v_cvt_f32_ubyte1 %0, %4\n v_cvt_f32_ubyte2 %1, %5\n v_cvt_f32_ubyte3 %2, %6\n v_cvt_f32_ubyte0 %3, %7" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));) }
        if (KIND == 3) { REP64(asm volatile("v_max3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (KIND == 4) { REP64(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));) }
        if (KIND == 5) { REP64(asm volatile("v_and_b32 %0, %0, %4\n v_lshl_or_b32 %1, %1, %5, %4\n v_xor_b32 %2, %2, %4\n v_add_u32 %3, %3, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x7fffffffu), "v"(1u));) }
        if (KIND == 6) { REP64(asm volatile("v_cmp_le_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_gt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");) }
        if (KIND == 7) { REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_mul_f32 %2, %2, %4\n v_sub_f32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (KIND == 9) { REP64(asm volatile("v_perm_b32 %0, %4, %5, %6\n v_perm_b32 %1, %4, %5, %7\n v_perm_b32 %2, %4, %5, %6\n v_perm_b32 %3, %4, %5, %7" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(__float_as_uint(b)), "v"(0x3f800000u), "v"(0x07060005u), "v"(0x07060105u));) }
        if (KIND == 10) { REP64(asm volatile("v_bfe_u32 %0, %4, 8, 8\n v_bfe_u32 %1, %5, 16, 8\n v_bfe_u32 %2, %4, 24, 8\n v_bfe_u32 %3, %5, 0, 8" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(__float_as_uint(b)), "v"(__float_as_uint(c)));) }
        if (KIND == 11) { REP64(asm volatile("v_and_or_b32 %0, %4, %5, %6\n v_and_or_b32 %1, %4, %5, %6\n v_and_or_b32 %2, %4, %5, %6\n v_and_or_b32 %3, %4, %5, %6" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(__float_as_uint(b)), "v"(0xff00u), "v"(0x3f800000u));) }
        if (KIND == 12) { REP64(asm volatile("v_max_f32 %0, %0, %4\n v_min_f32 %1, %1, %5\n v_max_f32 %2, %2, %4\n v_min_f32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (KIND == 13) { REP64(asm volatile("v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %5\n v_cvt_f32_u32 %2, %6\n v_cvt_f32_u32 %3, %7" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));) }
        if (KIND == 14) { REP64(asm volatile("v_lshl_or_b32 %0, %4, %5, %0\n v_lshl_or_b32 %1, %4, %5, %1\n v_lshl_or_b32 %2, %4, %5, %2\n v_lshl_or_b32 %3, %4, %5, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(3u), "v"(5u));) }
        if (KIND == 15) { REP64(asm volatile("v_fma_mix_f32 %0, %4, %5, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %5, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %6, %5, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %6, %5, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0x3c003800u), "v"(b), "v"(0x40003a00u));) }
        if (KIND == 16) { REP64(asm volatile("v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16 %1, %5\n v_cvt_f32_f16 %2, %4\n v_cvt_f32_f16 %3, %5" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(0x3c003800u), "v"(0x40003a00u));) }
        if (KIND == 17) { REP64(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %6, %5, %2\n v_fma_f32 %3, %6, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(b), "v"(seed));) }
        if (KIND == 8) { REP64(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p3.y + __uint_as_float(u0 ^ u1 ^ u2 ^ u3);
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
    float *out; CHECK(hipMalloc(&out, (size_t) cus * 8 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[] = { "v_fma_f32", "v_pk_fma_f32", "v_cvt_f32_ubyteN", "v_max3/min3_f32", "v_mov_b32", "int and/lshl_or/xor/add", "v_cmp+v_cndmask", "v_mul/add/sub_f32", "v_rcp_f32", "v_perm_b32", "v_bfe_u32", "v_and_or_b32", "v_max/min_f32", "v_cvt_f32_u32", "v_lshl_or_b32", "v_fma_mix_f32 (f16 src0)", "v_cvt_f32_f16", "v_fma_f32 acc (c += a*b)" };
    for (int waves_per_simd : { 2, 8 }) {
        int grid = cus * waves_per_simd;      // 256-thread block = 4 waves = 1 per SIMD
        for (int kind = 0; kind < 18; ++kind) {
            int iters = 200;
            auto launch = [&]() {
                switch (kind) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 6: hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 7: hipLaunchKernelGGL(k<7>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 8: hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 9: hipLaunchKernelGGL(k<9>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 10: hipLaunchKernelGGL(k<10>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 11: hipLaunchKernelGGL(k<11>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 12: hipLaunchKernelGGL(k<12>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 13: hipLaunchKernelGGL(k<13>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 14: hipLaunchKernelGGL(k<14>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 15: hipLaunchKernelGGL(k<15>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 16: hipLaunchKernelGGL(k<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                    case 17: hipLaunchKernelGGL(k<17>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                }
            };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            double instr_per_simd = (double) waves_per_simd * iters * 64 * 4;
            printf("%d waves/SIMD  %-26s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.0f MHz)\n", waves_per_simd, names[kind], ms,
                   ms * 1e-3 * clk / instr_per_simd, clk / 1e6);
        }
    }
    return 0;
}
