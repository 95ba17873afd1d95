// FETCH_SIZE calibration for the index lookup's access pattern (MI355X_MICROARCH.md, HBM: "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Three kernels
// with a KNOWN number of accesses over a table far larger than the 256 MiB Infinity Cache:
//   stream16 : every lane reads consecutive 16 B (the wide coalesced case the guide's x2 rule is for)
//   rand16   : every lane reads 16 B at an independent random 16-byte-aligned address (a directory probe)
//   rand2x12 : every lane reads 12 consecutive uint16 at a random address (a short posting list)
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 profiles/calib_fetch.hip -o /tmp/calib_fetch
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum -d <out> -o c -- /tmp/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

__global__ void stream16(const uint4 *t, size_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = t[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) *sink = acc;
}

__global__ void rand16(const uint4 *t, size_t n16, size_t reads_per_thread, uint32_t *sink) {
    uint32_t acc = 0;
    uint64_t s = mix((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (size_t r = 0; r < reads_per_thread; ++r) {
        s = mix(s + r);
        const uint4 v = t[s % n16];
        acc ^= v.x ^ v.w;
    }
    if (acc == 0x12345u) *sink = acc;
}

__global__ void rand2x12(const uint16_t *t, size_t n2, size_t reads_per_thread, uint32_t *sink) {
    uint32_t acc = 0;
    uint64_t s = mix((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 7);
    for (size_t r = 0; r < reads_per_thread; ++r) {
        s = mix(s + r);
        const uint16_t *p = t + (s % (n2 - 16));
#pragma unroll
        for (int j = 0; j < 12; ++j) acc += p[j];
    }
    if (acc == 0x12345u) *sink = acc;
}

int main() {
    const size_t bytes = (size_t)4 << 30;                        // 4 GiB: 16x the Infinity Cache
    void *t = nullptr;
    uint32_t *sink = nullptr;
    if (hipMalloc(&t, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(t, 1, bytes);
    const size_t n16 = bytes / 16;
    const int blocks = 256 * 8, threads = 256;
    const size_t rpt = 256;                                      // random reads per thread
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream16, dim3(blocks), dim3(threads), 0, 0, (const uint4 *)t, n16, sink);
        hipLaunchKernelGGL(rand16, dim3(blocks), dim3(threads), 0, 0, (const uint4 *)t, n16, rpt, sink);
        hipLaunchKernelGGL(rand2x12, dim3(blocks), dim3(threads), 0, 0, (const uint16_t *)t, bytes / 2, rpt, sink);
    }
    hipDeviceSynchronize();
    printf("{\"table_bytes\": %zu, \"stream16_bytes\": %zu, \"rand16_reads\": %zu, \"rand2x12_reads\": %zu}\n", bytes, bytes,
           (size_t)blocks * threads * rpt, (size_t)blocks * threads * rpt);
    return 0;
}
