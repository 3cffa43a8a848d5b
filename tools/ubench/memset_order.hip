// Is hipMemsetAsync ordered with the kernels around it on the SAME stream?  (round-2 hunt for the BENCH_r01 GPU fault)
// Pattern of run_chunk: [kernel A writes buf] ... hipMemsetAsync(buf, 0) ; kernel B reads buf and counts non-zero words.
// Tested for hipMalloc memory and for hipMemCreate/hipMemMap (VMM) memory, on the null stream and on a created stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_dirty(uint32_t *buf, uint32_t n, uint32_t v, int spin) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = v;
    for (int k = 0; k < spin; ++k) acc = acc * 1664525u + 1013904223u;     // keep the GPU busy so that later commands queue up behind it
    if (i < n) buf[i] = acc | 1u;
}
__global__ void k_check(const uint32_t *buf, uint32_t n, uint32_t *bad) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && buf[i] != 0u) atomicAdd(bad, 1u);
}
static void *vmm_alloc(size_t bytes) {
    int dev = 0; CK(hipGetDevice(&dev));
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    size_t sz = (bytes + gran - 1) / gran * gran;
    void *p = nullptr; CK(hipMemAddressReserve(&p, sz, gran, nullptr, 0));
    hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, sz, &prop, 0)); CK(hipMemMap(p, sz, 0, h, 0));
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite; CK(hipMemSetAccess(p, sz, &acc, 1));
    return p;
}
int main() {
    for (uint32_t n : { 1280u, 8u, 1u })         // 5120 bytes (the counters run_chunk clears), 32 bytes (totals), 4 bytes (status)
    for (int vmm = 0; vmm < 2; ++vmm) for (int own_stream = 0; own_stream < 3; ++own_stream) for (int offset = 0; offset < 2; ++offset) {
        uint32_t *base = nullptr, *bad = nullptr;
        if (vmm) base = (uint32_t *) vmm_alloc(1 << 22); else CK(hipMalloc(&base, 1 << 22));
        CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
        uint32_t *buf = base + (offset ? 131328 : 0);     // an interior pointer, as cnt_items / cur_trace are
        hipStream_t s = nullptr; if (own_stream == 1) CK(hipStreamCreate(&s)); if (own_stream == 2) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int it = 0; it < 2000; ++it) {
            hipLaunchKernelGGL(k_dirty, dim3(2048), dim3(256), 0, s, buf, n, (uint32_t) it, 2000);
            CK(hipMemsetAsync(buf, 0, n * 4, s));
            hipLaunchKernelGGL(k_check, dim3((n + 255) / 256), dim3(256), 0, s, buf, n, bad);
        }
        CK(hipStreamSynchronize(s));
        uint32_t h = 0; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
        printf("bytes=%u memory=%s stream=%s pointer=%s : %u non-zero words seen after hipMemsetAsync in 2000 rounds\n", n * 4, vmm ? "vmm" : "hipMalloc",
               own_stream == 0 ? "null" : own_stream == 1 ? "created" : "non-blocking", offset ? "interior" : "base", h);
    }
    return 0;
}
