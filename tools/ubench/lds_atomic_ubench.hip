// LDS atomic rate micro-benchmark for gfx950 (round 3): what one wave instruction of ds_add_f32 / ds_add_u32 / ds_add_rtn_u32 costs the CU's LDS pipeline as a function of the
// address pattern (distinct banks, all lanes one address, runs of 4 / 16 equal addresses) and of the number of active lanes.  Method: every block of 256 threads (4 waves)
// loops over `iters` x 16 atomics on a 24 KB LDS array; B blocks per CU; cycles per wave instruction = time x clock x CUs / (blocks x 4 x iters x 16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>   // 0: ds_add_f32, 1: ds_add_u32, 2: ds_add_rtn_u32, 3: ds_add_f32 via 1 active lane per wave, 4: ds_add_f32 with 8 active lanes, 5: ds_add_u64
__global__ __launch_bounds__(256) void k(float *out, int iters, int pattern) {
    __shared__ float lds[6144];
    for (int i = threadIdx.x; i < 6144; i += 256) lds[i] = 0.f;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // pattern 0: lane-distinct consecutive floats (conflict-free); 1: all lanes one address; 2: runs of 4 equal; 3: runs of 16 equal; 4: random-ish (hash) addresses; 5: stride 3 floats (the texel layout)
    uint32_t idx;
    if (pattern == 0) idx = lane; else if (pattern == 1) idx = 0; else if (pattern == 2) idx = lane >> 2; else if (pattern == 3) idx = lane >> 4;
    else if (pattern == 4) idx = (lane * 2654435761u >> 20) % 1500u; else idx = lane * 3u;
    idx += wave * 1536u;
    uint32_t acc = 0;
    const bool on = OP == 3 ? lane == 0 : OP == 4 ? lane < 8 : true;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t a = idx + (uint32_t) ((r * 7 + i) & 15);
            if (OP == 0 || OP == 3 || OP == 4) { if (on) atomicAdd(&lds[a], 1.f); }
            else if (OP == 1) atomicAdd(reinterpret_cast<uint32_t *>(lds) + a, 1u);
            else if (OP == 2) acc += atomicAdd(reinterpret_cast<uint32_t *>(lds) + a, 1u);
            else atomicAdd(reinterpret_cast<unsigned long long *>(lds) + (a >> 1), 1ull);
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x] + (float) acc;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;
    float *out; CHECK(hipMalloc(&out, (size_t) cus * 8 * 256 * 4));
    const char *ops[] = { "ds_add_f32", "ds_add_u32", "ds_add_rtn_u32", "ds_add_f32 (1 lane on)", "ds_add_f32 (8 lanes on)", "ds_add_u64" };
    const char *pats[] = { "64 distinct consecutive", "one address", "runs of 4", "runs of 16", "scattered", "stride 3 floats" };
    printf("%d CUs, %.0f MHz; LDS cycles per wave instruction (per CU)\n", cus, clk / 1e6);
    for (int bpc = 1; bpc <= 4; bpc *= 4) for (int op = 0; op < 6; ++op) for (int pat = 0; pat < 6; ++pat) {
        if ((op == 3 || op == 4) && pat != 0) continue;
        const int iters = 2000, blocks = cus * bpc;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            switch (op) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, pat); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, pat); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, pat); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, pat); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, pat); break;
            default: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, out, iters, pat); break;
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        }
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_cu = (double) bpc * 4 * iters * 16;
        printf("blocks/CU %d  %-24s %-26s %8.1f cycles\n", bpc, ops[op], pats[pat], ms * 1e-3 * clk / instr_per_cu);
    }
    return 0;
}
