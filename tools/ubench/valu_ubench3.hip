// Round-6 VALU issue-rate micro-benchmark for gfx950 (companion of valu_ubench.hip / valu_ubench2.hip): integer min / max (could the slab test's four float
// min3 / max3 per child be integer ones?), the byte decode by v_and_b32 into a DENORMAL float (as_float(w & 0xff << 8k) = q * 2^(8k - 149) for k = 0..2, exact through
// the denormal / normal seam) fed to v_fma_f32, and the plain integer ops around them.  Same method as the other two files.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
typedef float f2 __attribute__((ext_vector_type(2)));

#define KINDS(X) \
  X(0,  "v_fma_f32 (reference)",        "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9") \
  X(1,  "v_max3/min3_i32",              "v_max3_i32 %4, %4, %10, %11\n v_min3_i32 %5, %5, %10, %11\n v_max3_i32 %6, %6, %10, %11\n v_min3_i32 %7, %7, %10, %11") \
  X(2,  "v_max/min_i32 (VOP2)",         "v_max_i32 %4, %10, %4\n v_min_i32 %5, %11, %5\n v_max_i32 %6, %10, %6\n v_min_i32 %7, %11, %7") \
  X(3,  "v_max3/min3_u32",              "v_max3_u32 %4, %4, %10, %11\n v_min3_u32 %5, %5, %10, %11\n v_max3_u32 %6, %6, %10, %11\n v_min3_u32 %7, %7, %10, %11") \
  X(4,  "v_max/min_u32 (VOP2)",         "v_max_u32 %4, %10, %4\n v_min_u32 %5, %11, %5\n v_max_u32 %6, %10, %6\n v_min_u32 %7, %11, %7") \
  X(5,  "v_max3/min3_f32 (reference)",  "v_max3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9") \
  X(6,  "v_add3_u32",                   "v_add3_u32 %4, %4, %10, %11\n v_add3_u32 %5, %5, %10, %11\n v_add3_u32 %6, %6, %10, %11\n v_add3_u32 %7, %7, %10, %11") \
  X(7,  "v_sub_u32",                    "v_sub_u32 %4, %4, %10\n v_sub_u32 %5, %5, %11\n v_sub_u32 %6, %6, %10\n v_sub_u32 %7, %7, %11") \
  X(8,  "v_alignbit_b32",               "v_alignbit_b32 %4, %4, %10, 31\n v_alignbit_b32 %5, %5, %11, 31\n v_alignbit_b32 %6, %6, %10, 31\n v_alignbit_b32 %7, %7, %11, 31") \
  X(9,  "v_and_b32 literal mask",       "v_and_b32 %4, 0xff00, %4\n v_and_b32 %5, 0xff0000, %5\n v_and_b32 %6, 0xff, %6\n v_and_b32 %7, 0xff00, %7") \
  X(10, "v_fma_f32 denormal src",       "v_fma_f32 %0, %12, %8, %0\n v_fma_f32 %1, %13, %8, %1\n v_fma_f32 %2, %14, %8, %2\n v_fma_f32 %3, %15, %8, %3") \
  X(11, "v_or3_b32",                    "v_or3_b32 %4, %4, %10, %11\n v_or3_b32 %5, %5, %10, %11\n v_or3_b32 %6, %6, %10, %11\n v_or3_b32 %7, %7, %10, %11") \
  X(12, "v_lshrrev_b32",                "v_lshrrev_b32 %4, 1, %4\n v_lshrrev_b32 %5, 1, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7") \
  X(13, "v_cvt_f32_ubyteN (reference)", "v_cvt_f32_ubyte1 %0, %4\n v_cvt_f32_ubyte2 %1, %5\n v_cvt_f32_ubyte3 %2, %6\n v_cvt_f32_ubyte0 %3, %7") \
  X(14, "v_and_b32 + v_fma (denorm decode)", "v_and_b32 %4, 0xff00, %10\n v_fma_f32 %0, %4, %8, %0\n v_and_b32 %5, 0xff0000, %11\n v_fma_f32 %1, %5, %8, %1") \
  X(15, "v_cvt_ubyte + v_fma (today)",  "v_cvt_f32_ubyte1 %2, %10\n v_fma_f32 %0, %2, %8, %0\n v_cvt_f32_ubyte2 %3, %11\n v_fma_f32 %1, %3, %8, %1") \
  X(16, "v_sub_f32 (reference)",        "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %9\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %9") \
  X(17, "v_xor_b32",                    "v_xor_b32 %4, %10, %4\n v_xor_b32 %5, %10, %5\n v_xor_b32 %6, %11, %6\n v_xor_b32 %7, %11, %7") \
  X(18, "v_add_u32",                    "v_add_u32 %4, %10, %4\n v_add_u32 %5, %10, %5\n v_add_u32 %6, %11, %6\n v_add_u32 %7, %11, %7") \
  X(19, "v_ashrrev_i32",                "v_ashrrev_i32 %4, 1, %4\n v_ashrrev_i32 %5, 1, %5\n v_ashrrev_i32 %6, 3, %6\n v_ashrrev_i32 %7, 3, %7")

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a0 = threadIdx.x * seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = seed * 1.0001f, c = seed * 0.5f;
    uint32_t u0 = 0x3c003800u + threadIdx.x, u1 = u0 * 3u, u2 = u0 * 5u, u3 = u0 * 7u, ub = 0x3c003c00u, uc = 0x38003a00u;
    float p0 = __uint_as_float(0x00003400u + threadIdx.x), p1 = __uint_as_float(0x00120000u + threadIdx.x), p2 = __uint_as_float(0x000000a0u + threadIdx.x), p3 = __uint_as_float(0x00800000u + threadIdx.x);
    for (int i = 0; i < iters; ++i) {
#define X(ID, NAME, ASM) if (KIND == ID) { REP64(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(b), "v"(c), "v"(ub), "v"(uc), "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "vcc");) }
        KINDS(X)
#undef X
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0 + p1 + p2 + p3 + __uint_as_float(u0 ^ u1 ^ u2 ^ u3);
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
