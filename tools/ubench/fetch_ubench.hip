// Micro-benchmark: how fast can a CU fetch scattered 80-byte records (BVH nodes) with different
// access shapes?  Dependent chains (next index derives from the loaded data), 64-lane waves.
//   A: lane-per-record, 5 x dwordx4 per lane (AoS, what k_trace_closest does today)
//   B: lane-per-record, fetched TRANSPOSED: 5 lanes fetch one record (one 16 B piece each), 12 records
//      per load instruction, staged through LDS, then each lane reads its own record from LDS
//   C: lane-per-record, only 1 x dwordx4 per lane (cost of one scattered load instruction)
//   D: 8 lanes per record (8 records per wave): lanes 0..4 of the group load one piece each
// Build: hipcc -O3 --offload-arch=gfx950 fetch_ubench.hip -o fetch_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static constexpr int kBlock = 256;

__device__ __forceinline__ uint32_t mix(uint4 v) { return v.x ^ (v.y * 0x9e3779b9u) ^ (v.z >> 3) ^ (v.w * 0x85ebca6bu); }

template <int STRIDE16>   // record stride in 16-byte units (5 = packed 80 B, 8 = 128 B aligned)
__global__ __launch_bounds__(kBlock) void k_A(const uint4 *rec, uint32_t nrec, int iters, uint32_t *out) {
    uint32_t idx = (blockIdx.x * kBlock + threadIdx.x) * 2654435761u % nrec, acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint4 *p = rec + (size_t) idx * STRIDE16;
        uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
        uint32_t h = mix(a) + mix(b) + mix(c) + mix(d) + mix(e);
        acc += h; idx = (h ^ (idx * 31u)) % nrec;
    }
    out[blockIdx.x * kBlock + threadIdx.x] = acc;
}

template <int STRIDE16>
__global__ __launch_bounds__(kBlock) void k_C(const uint4 *rec, uint32_t nrec, int iters, uint32_t *out) {
    uint32_t idx = (blockIdx.x * kBlock + threadIdx.x) * 2654435761u % nrec, acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint4 *p = rec + (size_t) idx * STRIDE16;
        uint4 a = p[0];
        uint32_t h = mix(a) * 5u;
        acc += h; idx = (h ^ (idx * 31u)) % nrec;
    }
    out[blockIdx.x * kBlock + threadIdx.x] = acc;
}

template <int STRIDE16>
__global__ __launch_bounds__(kBlock) void k_B(const uint4 *rec, uint32_t nrec, int iters, uint32_t *out) {
    __shared__ uint4 stage[kBlock * 5];          // 80 B per lane, 20 KB per block
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint4 *wst = stage + wave * 64 * 5;
    uint32_t idx = (blockIdx.x * kBlock + threadIdx.x) * 2654435761u % nrec, acc = 0;
    const uint32_t slot = lane / 5u, piece = lane - slot * 5u;   // lanes 60..63 idle in the fetch
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const uint32_t src = r * 12u + slot;                 // lane whose record this lane helps to fetch
            const uint32_t ridx = (uint32_t) __shfl((int) idx, (int) (src & 63u), 64);
            if (lane < 60u && src < 64u) wst[src * 5u + piece] = rec[(size_t) ridx * STRIDE16 + piece];
        }
        __builtin_amdgcn_wave_barrier();
        uint4 a = wst[lane * 5 + 0], b = wst[lane * 5 + 1], c = wst[lane * 5 + 2], d = wst[lane * 5 + 3], e = wst[lane * 5 + 4];
        __builtin_amdgcn_wave_barrier();
        uint32_t h = mix(a) + mix(b) + mix(c) + mix(d) + mix(e);
        acc += h; idx = (h ^ (idx * 31u)) % nrec;
    }
    out[blockIdx.x * kBlock + threadIdx.x] = acc;
}

template <int STRIDE16>
__global__ __launch_bounds__(kBlock) void k_D(const uint4 *rec, uint32_t nrec, int iters, uint32_t *out) {
    // 8 lanes per record: every group of 8 lanes follows ONE chain
    const uint32_t lane = threadIdx.x & 63u, piece = lane & 7u;
    uint32_t idx = ((blockIdx.x * kBlock + threadIdx.x) >> 3) * 2654435761u % nrec, acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint4 a = make_uint4(0, 0, 0, 0);
        if (piece < 5u) a = rec[(size_t) idx * STRIDE16 + piece];
        uint32_t h = mix(a);
        h += __shfl_xor((int) h, 1, 64); h += __shfl_xor((int) h, 2, 64); h += __shfl_xor((int) h, 4, 64);
        acc += h; idx = (h ^ (idx * 31u)) % nrec;
    }
    out[blockIdx.x * kBlock + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    int iters = 256;
    uint32_t *out; CHECK(hipMalloc(&out, 4096 * kBlock * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    double clk = prop.clockRate * 1e3; int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
    for (uint32_t nrec : { 16u, 4096u, 8192u, 400000u }) {
        for (int stride : { 5, 8 }) {
            size_t bytes = (size_t) nrec * stride * 16;
            std::vector<uint32_t> h(bytes / 4);
            for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t) (i * 2654435761u) ^ (uint32_t) (i >> 7);
            uint4 *rec; CHECK(hipMalloc(&rec, bytes)); CHECK(hipMemcpy(rec, h.data(), bytes, hipMemcpyHostToDevice));
            for (int blocks_per_cu : { 2, 6 }) {
                int grid = cus * blocks_per_cu;
                auto run = [&](const char *name, auto launch, double recs_per_thread) {
                    launch(grid); CHECK(hipDeviceSynchronize());
                    CHECK(hipEventRecord(e0)); launch(grid); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    double fetches = (double) grid * kBlock * recs_per_thread * iters;
                    printf("nrec %7u (%6.1f KB) stride %3d B  %d blk/CU  %-28s %8.3f ms  %7.2f Gfetch/s  %6.2f clk/fetch/CU\n", nrec, bytes / 1024.0, stride * 16,
                           blocks_per_cu, name, ms, fetches / ms / 1e6, ms * 1e-3 * clk / (fetches / cus));
                };
                if (stride == 5) {
                    run("A lane/rec 5xdwordx4", [&](int g) { hipLaunchKernelGGL(k_A<5>, dim3(g), dim3(kBlock), 0, 0, rec, nrec, iters, out); }, 1);
                    run("B transposed via LDS", [&](int g) { hipLaunchKernelGGL(k_B<5>, dim3(g), dim3(kBlock), 0, 0, rec, nrec, iters, out); }, 1);
                    run("C lane/rec 1xdwordx4", [&](int g) { hipLaunchKernelGGL(k_C<5>, dim3(g), dim3(kBlock), 0, 0, rec, nrec, iters, out); }, 1);
                    run("D 8 lanes/rec", [&](int g) { hipLaunchKernelGGL(k_D<5>, dim3(g), dim3(kBlock), 0, 0, rec, nrec, iters, out); }, 1.0 / 8);
                } else {
                    run("A lane/rec 5xdwordx4", [&](int g) { hipLaunchKernelGGL(k_A<8>, dim3(g), dim3(kBlock), 0, 0, rec, nrec, iters, out); }, 1);
                    run("B transposed via LDS", [&](int g) { hipLaunchKernelGGL(k_B<8>, dim3(g), dim3(kBlock), 0, 0, rec, nrec, iters, out); }, 1);
                    run("C lane/rec 1xdwordx4", [&](int g) { hipLaunchKernelGGL(k_C<8>, dim3(g), dim3(kBlock), 0, 0, rec, nrec, iters, out); }, 1);
                    run("D 8 lanes/rec", [&](int g) { hipLaunchKernelGGL(k_D<8>, dim3(g), dim3(kBlock), 0, 0, rec, nrec, iters, out); }, 1.0 / 8);
                }
            }
            CHECK(hipFree(rec));
        }
    }
    return 0;
}
