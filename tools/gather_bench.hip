// Divergent-gather ceiling of gfx950's vector memory path (tools/, not product): every lane of every wave loads W bytes
// from a pseudo-random entry of a table of S bytes.  What the hash-grid gather of csrc/field.hip does per corner, without
// anything else.  Reports lane-loads/s, chip-wide and per CU per clock (2.4 GHz), for S in {16 KiB .. 256 MiB} and
// W in {4, 8, 16}; `coherent` variants let groups of G consecutive lanes fall in one 128-B line (neighbouring samples of
// a ray meeting the same coarse cell).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_bench tools/gather_bench.hip && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int W> struct Vec;
template <> struct Vec<4> { using T = uint32_t; };
template <> struct Vec<8> { using T = uint2; };
template <> struct Vec<16> { using T = uint4; };

__device__ __forceinline__ uint32_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t fold(uint2 v) { return v.x ^ v.y; }
__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <int W, int G>
__global__ __launch_bounds__(256) void gather(const uint8_t* __restrict__ table, uint32_t mask_entries, int iters, uint32_t* out) {
    using T = typename Vec<W>::T;
    const T* t = (const T*)table;
    uint32_t lane_id = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t grp = lane_id / G, sub = lane_id % G;
    uint32_t s = grp * 2654435761u + 12345u;
    uint32_t acc = 0;
    constexpr int per_line = 128 / W;
    for (int i = 0; i < iters; ++i) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            uint32_t h = s ^ (s >> 15);
            uint32_t e = G == 1 ? h : (h / per_line * per_line + (sub * (per_line / G) + (h >> 20)) % per_line);
            v[u] = t[e & mask_entries];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= fold(v[u]);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int W, int G>
double run(const uint8_t* table, size_t bytes, int blocks, int iters, uint32_t* out) {
    uint32_t mask = (uint32_t)(bytes / W - 1);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    gather<W, G><<<blocks, 256>>>(table, mask, iters, out);
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) gather<W, G><<<blocks, 256>>>(table, mask, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double loads = 3.0 * blocks * 256.0 * iters * 8;
    return loads / (ms * 1e-3);
}

int main() {
    size_t cap = 256u << 20;
    uint8_t* table; hipMalloc(&table, cap);
    std::vector<uint32_t> h(cap / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i * 2246822519u;
    hipMemcpy(table, h.data(), cap, hipMemcpyHostToDevice);
    uint32_t* out; hipMalloc(&out, 4);
    const int blocks = 256 * 8;      // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    const int iters = 256;
    size_t sizes[] = {16u << 10, 256u << 10, 2u << 20, 4u << 20, 16u << 20, 64u << 20, 256u << 20};
    printf("%-10s %-4s %-6s %12s %14s %12s\n", "table", "W", "lanes/line", "Gloads/s", "loads/clk/CU", "GB/s useful");
    for (size_t s : sizes) {
        double r;
#define ROW(W, G) r = run<W, G>(table, s, blocks, iters, out); \
        printf("%-10zu %-4d %-6d %12.1f %14.3f %12.1f\n", s >> 10, W, G, r * 1e-9, r / 256 / 2.4e9, r * W * 1e-9);
        ROW(4, 1) ROW(8, 1) ROW(16, 1) ROW(8, 4) ROW(8, 16)
    }
    return 0;
}
