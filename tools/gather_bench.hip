// Microbenchmark: throughput of divergent 8-byte gathers on MI355X as a function of the working set and of how many
// lanes of a wave share an address.  Build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o gpurun_out/gather_bench
// Each wave issues batches of 32 independent global_load_dwordx2 (as field_kernel does) from a table of `entries` 8-byte
// entries; lane addresses come from a cheap hash of (wave, iteration, lane / share).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(512, 2) gather_kernel(const f2 *__restrict__ table, uint32_t mask, int share, int iters, float *out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    f2 acc = {0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f2 v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            uint32_t h = (wave * 2654435761u) ^ ((uint32_t)(it * 32 + k) * 805459861u) ^ ((lane / share) * 2246822519u);
            h ^= h >> 15; h *= 2654435761u; h ^= h >> 13;
            v[k] = table[h & mask];
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) acc += v[k];
    }
    if (acc.x == 123.f) out[0] = acc.y;
}

int main() {
    const size_t max_entries = 1u << 26;   // 512 MB
    f2 *table; float *out;
    hipMalloc(&table, max_entries * sizeof(f2)); hipMalloc(&out, 4);
    hipMemset(table, 0, max_entries * sizeof(f2));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 16, grid = 256 * 8;
    printf("working_set share  lane_gathers/clk/CU   GB/s(8B/lane)\n");
    for (int lg = 10; lg <= 26; lg += 2) {
        for (int share : {1, 4, 64}) {
            const uint32_t mask = (1u << lg) - 1;
            hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(512), 0, 0, table, mask, share, 2, out);
            hipEventRecord(e0);
            hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(512), 0, 0, table, mask, share, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double lanes = (double)grid * 8 * 64 * 32 * iters;
            printf("%8.1f KB  %3d   %8.3f   %8.1f   (%.3f ms)\n", (double)(8u << lg) / 1024.0, share, lanes / (ms * 1e-3 * 2.4e9 * 256), lanes * 8 / (ms * 1e-3) / 1e9, ms);
        }
    }
    return 0;
}
