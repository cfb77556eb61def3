// Scattered float-atomic ceiling of gfx950 (tools/, not product): what the hash-table gradient scatter of csrc/train.hip issues,
// without anything else.  Every group of G consecutive lanes adds G consecutive dwords (G = 4: the four features of one table entry,
// a 16-byte "quad"; G = 1: one dword per lane; G = 64: a whole 256-byte row) at a pseudo-random entry of a table of S bytes.
// Variants: fp32 adds (global_atomic_add_f32), packed fp16 adds (global_atomic_pk_add_f16: two features per lane, a quad is 8 bytes
// from 2 lanes), 64-bit integer adds (deterministic fixed-point accumulation).  Reports adds/s by group and added bytes/s.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_bench tools/atomic_bench.hip && /tmp/atomic_bench
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>

template <int MODE, int G>   // MODE 0: f32, 1: pk f16 (G counts 4-byte lanes), 2: u64 (G counts 8-byte lanes)
__global__ __launch_bounds__(256) void scatter(void* __restrict__ table, uint32_t mask_groups, int iters) {
    const uint32_t lane_id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = lane_id / G, sub = lane_id % G;
    uint32_t s = grp * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            const uint32_t h = (s ^ (s >> 15)) & mask_groups;
            const size_t e = (size_t)h * G + sub;
            if (MODE == 0) atomicAdd((float*)table + e, 1.0f);
            else if (MODE == 1) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                h2 v = {(_Float16)1.0f, (_Float16)1.0f};
                __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)((h2*)table + e), v);
            } else atomicAdd((unsigned long long*)table + e, 1ull);
        }
    }
}

template <int MODE, int G>
void run(void* table, size_t bytes, const char* name) {
    const int lane_bytes = MODE == 2 ? 8 : 4;
    const uint32_t mask = (uint32_t)(bytes / ((size_t)lane_bytes * G) - 1);
    const int blocks = 256 * 16, iters = 16;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    scatter<MODE, G><<<blocks, 256>>>(table, mask, iters);
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) scatter<MODE, G><<<blocks, 256>>>(table, mask, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double lanes = 3.0 * blocks * 256.0 * iters * 8;
    printf("%-28s table %4zu MiB  group %2d lanes x %d B: %7.1f G groups/s  %7.1f G lane-adds/s  %6.3f TB/s added\n", name, bytes >> 20, G, lane_bytes,
           lanes / G / (ms * 1e-3) / 1e9, lanes / (ms * 1e-3) / 1e9, lanes * lane_bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t cap = 512u << 20;
    void* table; hipMalloc(&table, cap); hipMemset(table, 0, cap);
    for (size_t bytes : {(size_t)4 << 20, (size_t)128 << 20, (size_t)512 << 20}) {
        run<0, 1>(table, bytes, "f32 dword per lane");
        run<0, 4>(table, bytes, "f32 quad (16 B entry)");
        run<0, 16>(table, bytes, "f32 64-B segment");
        run<0, 64>(table, bytes, "f32 256-B row");
        run<1, 1>(table, bytes, "pk f16 dword per lane");
        run<1, 2>(table, bytes, "pk f16 quad (8 B entry)");
        run<1, 64>(table, bytes, "pk f16 256-B row");
        run<2, 1>(table, bytes, "u64 per lane");
        run<2, 4>(table, bytes, "u64 quad (32 B entry)");
        run<2, 32>(table, bytes, "u64 256-B row");
    }
    return 0;
}
