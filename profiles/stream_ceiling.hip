// What a pure streaming read reaches on this GPU: the practical ceiling the scene kernel's 6.7 TB/s is
// to be read against (8 TB/s is the interface peak).  16 GiB table, every lane reads 16 B per step with
// `nt` loads, several loads in flight, grid-stride; HIP events around 10 launches of each shape.
//   hipcc --offload-arch=gfx950 -O3 profiles/stream_ceiling.hip -o /tmp/stream_ceiling && /tmp/stream_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int U>
__global__ __launch_bounds__(256) void stream_read(const uint4 *__restrict__ t, size_t n16, uint32_t *sink) {
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256 * U;
    uint32_t acc = 0;
    for (; i + (size_t)(U - 1) * 256 < n16; i += step) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint4 *p = t + i + (size_t)u * 256;
            v[u].x = __builtin_nontemporal_load(&p->x);
            v[u].y = __builtin_nontemporal_load(&p->y);
            v[u].z = __builtin_nontemporal_load(&p->z);
            v[u].w = __builtin_nontemporal_load(&p->w);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int U>
static void run(const uint4 *t, size_t n16, uint32_t *sink, int blocks) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stream_read<U>, dim3(blocks), dim3(256), 0, 0, t, n16, sink);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(stream_read<U>, dim3(blocks), dim3(256), 0, 0, t, n16, sink);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("{\"loads_in_flight\": %d, \"blocks\": %d, \"TBps\": %.3f}\n", U, blocks, (double)n16 * 16 * 10 / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = (size_t)16 << 30, n16 = bytes / 16;
    uint4 *t;
    uint32_t *sink;
    if (hipMalloc(&t, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(t, 1, bytes);
    for (int blocks : {2048, 8192, 32768}) {
        run<2>(t, n16, sink, blocks);
        run<4>(t, n16, sink, blocks);
        run<8>(t, n16, sink, blocks);
    }
    return 0;
}
