// Do two 16-byte quad adds of ONE wave instruction that fall into the same 64-byte segment leave as one memory-side request?  (tools/, not product)
// Every 8 lanes hold two quads: A (lanes 0-3) at a random 16-B entry e of the table, B (lanes 4-7) at
//   same32: e ^ 1 (the x-neighbour of a hashed level when x is even)      same64: e ^ 2      other: an independent random entry
//   half:   only quad A issues (B's lanes idle)                            cross16: A and its partner e ^ 1 sit in lanes l and l + 16
// Reported: quads/s.  If same32 / same64 run at twice the quad rate of `other`, the pair is one request.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_pairs tools/atomic_pairs.hip && /tmp/atomic_pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int PAT>
__global__ __launch_bounds__(256) void scatter(float* __restrict__ table, uint32_t mask_entries, int iters) {
    const uint32_t lane_id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t feat = lane & 3;
    uint32_t grp, second;
    if (PAT == 4) { grp = (lane_id >> 6) * 8 + ((lane & 15) >> 2) ; second = (lane >> 4) & 1; grp = grp * 2 + (lane >> 5); }
    else { grp = lane_id >> 3; second = (lane >> 2) & 1; }
    uint32_t s = grp * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            uint32_t h = (s ^ (s >> 15)) & mask_entries;
            if (second) {
                if (PAT == 0 || PAT == 4) h ^= 1u;
                else if (PAT == 1) h ^= 2u;
                else if (PAT == 2) h = ((s * 747796405u) ^ (s >> 13)) & mask_entries;
            }
            if (PAT == 3 && second) continue;
            atomicAdd(table + (size_t)h * 4 + feat, 1.0f);
        }
    }
}

template <int PAT>
void run(float* table, size_t bytes, const char* name) {
    const uint32_t mask = (uint32_t)(bytes / 16 - 1);
    const int blocks = 256 * 16, iters = 16;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    scatter<PAT><<<blocks, 256>>>(table, mask, iters);
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) scatter<PAT><<<blocks, 256>>>(table, mask, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double quads = 3.0 * blocks * 256.0 * iters * 8 / 4 * (PAT == 3 ? 0.5 : 1.0);
    printf("%-10s table %4zu MiB: %7.1f G quads/s\n", name, bytes >> 20, quads / (ms * 1e-3) / 1e9);
}

int main() {
    const size_t cap = 128u << 20;
    float* table; hipMalloc(&table, cap); hipMemset(table, 0, cap);
    for (size_t bytes : {(size_t)4 << 20, (size_t)128 << 20}) {
        run<2>(table, bytes, "other");
        run<0>(table, bytes, "same32");
        run<1>(table, bytes, "same64");
        run<3>(table, bytes, "half");
        run<4>(table, bytes, "cross16");
    }
    return 0;
}
