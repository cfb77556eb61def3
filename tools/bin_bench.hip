// Where does the binned scatter spend its time?  (tools/, not product)  Stand-alone model of csrc/train.hip's bin_items_kernel /
// bin_accumulate_kernel on pseudo-random entries: N samples x 8 corners x LV levels, tables of 2^19 entries cut into bins of 4096.
//   pass A variants: 0 ranks + reservation only (no item stores)   1 + 16-byte item stores (as built)   2 stores staged through LDS, whole runs written by consecutive lanes (as built)
//   pass B variants: 0 loads only   1 loads + ds_add_f32 (as built)   2 loads + ds_add_u32   3 ds_add_f32, feature-major LDS   4 loads + one ds_add_f32 per item
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bin_bench tools/bin_bench.hip && /tmp/bin_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int kLog2 = 12; constexpr uint32_t kEntries = 1u << kLog2; constexpr int kMaxBins = 1024;
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t rnd(uint32_t s) { s ^= s >> 16; s *= 0x7feb352du; s ^= s >> 15; s *= 0x846ca68bu; s ^= s >> 16; return s; }

template <int VAR>
__global__ __launch_bounds__(256) void pass_a(int64_t n, uint32_t size, v4f *items, uint32_t *cursors, uint32_t cap) {
    __shared__ uint32_t s_cnt[kMaxBins], s_base[kMaxBins];
    __shared__ v4f s_stage[VAR == 2 ? 2048 : 1];
    __shared__ uint16_t s_binof[VAR == 2 ? 2048 : 1];
    const int lk = blockIdx.y;
    const uint32_t nb = size >> kLog2;
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) s_cnt[b] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx[8], rank[8];
    const bool live = i < n;
    // like the hash: the eight corners of a sample are unrelated entries
    for (int c = 0; c < 8; ++c) idx[c] = rnd((uint32_t)i * 8u + c + lk * 0x9e3779b9u) & (size - 1u);
    if (live) for (int c = 0; c < 8; ++c) {
        if (VAR == 3) { rank[c] = threadIdx.x; if (threadIdx.x < 8) s_cnt[idx[c] >> kLog2] = 16; }     // no rank atomics (timing floor of everything else)
        else rank[c] = atomicAdd(&s_cnt[idx[c] >> kLog2], 1u);
    }
    __syncthreads();
    if (VAR == 2) {   // exclusive scan of the counts (small: one wave)
        if (threadIdx.x < 64) {
            uint32_t run = 0;
            for (uint32_t b0 = 0; b0 < nb; b0 += 64) {
                const uint32_t b = b0 + threadIdx.x;
                uint32_t c = b < nb ? s_cnt[b] : 0, incl = c;
                for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if ((int)threadIdx.x >= d) incl += t; }
                if (b < nb) { s_base[b] = run + incl - c; }
                run += __shfl(incl, 63, 64);
            }
        }
        __syncthreads();
        if (live) for (int c = 0; c < 8; ++c) {
            const uint32_t b = idx[c] >> kLog2, pos = s_base[b] + rank[c];
            const float v = (float)idx[c];
            s_stage[pos] = v4f{v, v + 1, v + 2, v + 3};
            s_binof[pos] = (uint16_t)b;
        }
        __syncthreads();
        // reserve the global runs (reuse s_cnt as the global base minus the local base)
        for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
            const uint32_t c = s_cnt[b];
            const uint32_t g = c ? atomicAdd(&cursors[(size_t)lk * kMaxBins + b], c) : 0u;
            s_cnt[b] = g - s_base[b];
        }
        __syncthreads();
        const uint32_t total = min((int64_t)2048, (n - (int64_t)blockIdx.x * blockDim.x) * 8);
        for (uint32_t p = threadIdx.x; p < total; p += blockDim.x) {
            const uint32_t b = s_binof[p], slot = s_cnt[b] + p;
            if (slot < cap) items[((size_t)lk * nb + b) * cap + slot] = s_stage[p];
        }
        return;
    }
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
        const uint32_t c = s_cnt[b];
        s_base[b] = c ? atomicAdd(&cursors[(size_t)lk * kMaxBins + b], c) : 0u;
    }
    __syncthreads();
    if (!live) return;
    uint32_t acc = 0;
    for (int c = 0; c < 8; ++c) {
        const uint32_t b = idx[c] >> kLog2, slot = s_base[b] + rank[c];
        const float v = (float)idx[c];
        if (VAR == 1 || VAR == 3) { if (VAR == 3) acc += slot; else if (slot < cap) items[((size_t)lk * nb + b) * cap + slot] = v4f{v, v + 1, v + 2, v + 3}; }
        else acc += slot;
    }
    if (VAR == 0 && acc == 0x12345678u) items[0] = v4f{0, 0, 0, 0};
}

// pass A with S sub-chunks of 256 samples per workgroup: count, reserve once, then recompute the entries and take the ranks
template <int S, bool RESERVE>
__global__ __launch_bounds__(256) void pass_a_loop(int64_t n, uint32_t size, v4f *items, uint32_t *cursors, uint32_t cap, int n_levels) {
    __shared__ uint32_t s_cnt[kMaxBins], s_base[kMaxBins];
    const int lk = blockIdx.x % n_levels;
    const int64_t chunk = blockIdx.x / n_levels;
    const uint32_t nb = size >> kLog2;
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) s_cnt[b] = 0;
    __syncthreads();
    for (int sc = 0; sc < S; ++sc) {
        const int64_t i = (chunk * S + sc) * 256 + threadIdx.x;
        if (i < n) for (int c = 0; c < 8; ++c) atomicAdd(&s_cnt[(rnd((uint32_t)i * 8u + c + lk * 0x9e3779b9u) & (size - 1u)) >> kLog2], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
        const uint32_t c = s_cnt[b];
        s_base[b] = RESERVE ? (c ? atomicAdd(&cursors[(size_t)lk * kMaxBins + b], c) : 0u) : cursors[(size_t)lk * kMaxBins + b] + (uint32_t)chunk * 24;
        s_cnt[b] = 0;
    }
    __syncthreads();
    for (int sc = 0; sc < S; ++sc) {
        const int64_t i = (chunk * S + sc) * 256 + threadIdx.x;
        if (i < n) for (int c = 0; c < 8; ++c) {
            const uint32_t idx = rnd((uint32_t)i * 8u + c + lk * 0x9e3779b9u) & (size - 1u);
            const uint32_t b = idx >> kLog2, slot = s_base[b] + atomicAdd(&s_cnt[b], 1u);
            const float v = (float)idx;
            if (slot < cap) items[((size_t)lk * nb + b) * cap + slot] = v4f{v, v + 1, v + 2, v + 3};
        }
    }
}

template <int VAR, int THREADS>
__global__ __launch_bounds__(THREADS) void pass_b(uint32_t size, const v4f *items, const uint32_t *cursors, uint32_t cap, float *table) {
    __shared__ float s_sum[kEntries * 4];
    const int lk = blockIdx.y;
    const uint32_t nb = size >> kLog2, b = blockIdx.x;
    uint32_t count = cursors[(size_t)lk * kMaxBins + b];
    if (count > cap) count = cap;
    for (uint32_t e = threadIdx.x; e < kEntries; e += THREADS) reinterpret_cast<v4f *>(s_sum)[e] = v4f{0, 0, 0, 0};
    __syncthreads();
    const v4f *list = items + ((size_t)lk * nb + b) * cap;
    constexpr int U = 4;
    float keep = 0.f;
    for (uint32_t i0 = threadIdx.x; i0 < count; i0 += THREADS * U) {
        v4f it[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const uint32_t i = i0 + u * THREADS; it[u] = list[i < count ? i : count - 1u]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * THREADS >= count) break;
            const uint32_t e = (uint32_t)it[u][0] & (kEntries - 1u);
            if (VAR == 0) keep += it[u][0] + it[u][1] + it[u][2] + it[u][3];
            else if (VAR == 1) { float *d = s_sum + e * 4; atomicAdd(d, it[u][0]); atomicAdd(d + 1, it[u][1]); atomicAdd(d + 2, it[u][2]); atomicAdd(d + 3, it[u][3]); }
            else if (VAR == 2) { uint32_t *d = reinterpret_cast<uint32_t *>(s_sum) + e * 4; atomicAdd(d, (uint32_t)it[u][0]); atomicAdd(d + 1, (uint32_t)it[u][1]); atomicAdd(d + 2, (uint32_t)it[u][2]); atomicAdd(d + 3, (uint32_t)it[u][3]); }
            else if (VAR == 3) { atomicAdd(s_sum + e, it[u][0]); atomicAdd(s_sum + kEntries + e, it[u][1]); atomicAdd(s_sum + 2 * kEntries + e, it[u][2]); atomicAdd(s_sum + 3 * kEntries + e, it[u][3]); }
            else if (VAR == 4) { atomicAdd(s_sum + e * 4, it[u][0] + it[u][1] + it[u][2] + it[u][3]); }
        }
    }
    __syncthreads();
    if (VAR == 0) { if (keep == 123.456f) table[0] = keep; return; }
    v4f *out = reinterpret_cast<v4f *>(table) + ((size_t)lk * size + (size_t)b * kEntries);
    for (uint32_t e = threadIdx.x; e < kEntries; e += THREADS) {
        const v4f a = reinterpret_cast<const v4f *>(s_sum)[e];
        if (a[0] != 0.f || a[1] != 0.f || a[2] != 0.f || a[3] != 0.f) out[e] += a;
    }
}

template <int VAR>
__global__ __launch_bounds__(1024) void pass_b64(uint32_t size, const v4f *items, const uint32_t *cursors, uint32_t cap, float *table) {
    __shared__ unsigned long long s_sum[kEntries * 4];      // 128 KB
    const int lk = blockIdx.y;
    const uint32_t nb = size >> kLog2, b = blockIdx.x;
    uint32_t count = cursors[(size_t)lk * kMaxBins + b];
    if (count > cap) count = cap;
    for (uint32_t e = threadIdx.x; e < kEntries * 4; e += 1024) s_sum[e] = 0;
    __syncthreads();
    const v4f *list = items + ((size_t)lk * nb + b) * cap;
    constexpr int U = 4;
    for (uint32_t i0 = threadIdx.x; i0 < count; i0 += 1024 * U) {
        v4f it[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const uint32_t i = i0 + u * 1024; it[u] = list[i < count ? i : count - 1u]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * 1024 >= count) break;
            const uint32_t e = (uint32_t)it[u][0] & (kEntries - 1u);
            if (VAR == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(s_sum + e * 4 + k, (unsigned long long)(long long)((double)it[u][k] * 72057594037927936.0));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(reinterpret_cast<double *>(s_sum) + e * 4 + k, (double)it[u][k]);
            }
        }
    }
    __syncthreads();
    v4f *out = reinterpret_cast<v4f *>(table) + ((size_t)lk * size + (size_t)b * kEntries);
    for (uint32_t e = threadIdx.x; e < kEntries; e += 1024) {
        v4f a;
        for (int k = 0; k < 4; ++k) a[k] = VAR == 0 ? (float)((double)(long long)s_sum[e * 4 + k] * (1.0 / 72057594037927936.0)) : (float)reinterpret_cast<double *>(s_sum)[e * 4 + k];
        if (a[0] != 0.f || a[1] != 0.f || a[2] != 0.f || a[3] != 0.f) out[e] += a;
    }
}

int main() {
    const int64_t n = 929000; const int LV = 8; const uint32_t size = 1u << 19, nb = size >> kLog2;
    const uint32_t cap = (uint32_t)(n * 8 / nb * 9 / 8 + 2048);
    v4f *items; uint32_t *cursors; float *table;
    hipMalloc(&items, (size_t)LV * nb * cap * sizeof(v4f)); hipMalloc(&cursors, LV * kMaxBins * 4); hipMalloc(&table, (size_t)LV * size * 16);
    hipMemset(table, 0, (size_t)LV * size * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](const char *name, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) {
            hipMemset(cursors, 0, LV * kMaxBins * 4);
            if (name[5] == 'B') { pass_a<1><<<dim3((unsigned)((n + 255) / 256), LV), 256>>>(n, size, items, cursors, cap); }
            hipDeviceSynchronize();
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("%-60s %.3f ms   (%.2f TB/s of 16-byte items)\n", name, best, (double)n * 8 * LV * 16 / (best * 1e-3) / 1e12);
    };
    const dim3 ga((unsigned)((n + 255) / 256), LV), gb(nb, LV);
    time("pass A: ranks + reservation, no item stores", [&] { pass_a<0><<<ga, 256>>>(n, size, items, cursors, cap); });
    time("pass A: + one 16-byte store per item (first version)", [&] { pass_a<1><<<ga, 256>>>(n, size, items, cursors, cap); });
    time("pass A: items staged in LDS, runs written by consecutive lanes (as built)", [&] { pass_a<2><<<ga, 256>>>(n, size, items, cursors, cap); });
    time("pass B: loads only, 512 threads", [&] { pass_b<0, 512><<<gb, 512>>>(size, items, cursors, cap, table); });
    time("pass B: loads + 4 ds_add_f32 per item (first version)", [&] { pass_b<1, 512><<<gb, 512>>>(size, items, cursors, cap, table); });
    time("pass B: loads + 4 ds_add_u32 per item", [&] { pass_b<2, 512><<<gb, 512>>>(size, items, cursors, cap, table); });
    time("pass B: 4 ds_add_f32, feature-major LDS", [&] { pass_b<3, 512><<<gb, 512>>>(size, items, cursors, cap, table); });
    time("pass B: loads + 1 ds_add_f32 per item", [&] { pass_b<4, 512><<<gb, 512>>>(size, items, cursors, cap, table); });
    time("pass A: 256 samples per workgroup, plain load instead of the reservation", [&] { pass_a_loop<1, false><<<dim3((unsigned)((n + 255) / 256 * LV)), 256>>>(n, size, items, cursors, cap, LV); });
    time("pass A: 256 samples per workgroup, two-phase", [&] { pass_a_loop<1, true><<<dim3((unsigned)((n + 255) / 256 * LV)), 256>>>(n, size, items, cursors, cap, LV); });
    time("pass A: 1024 samples per workgroup, two-phase", [&] { pass_a_loop<4, true><<<dim3((unsigned)((n + 1023) / 1024 * LV)), 256>>>(n, size, items, cursors, cap, LV); });
    time("pass A: 2048 samples per workgroup, two-phase", [&] { pass_a_loop<8, true><<<dim3((unsigned)((n + 2047) / 2048 * LV)), 256>>>(n, size, items, cursors, cap, LV); });
    time("pass A: 4096 samples per workgroup, two-phase", [&] { pass_a_loop<16, true><<<dim3((unsigned)((n + 4095) / 4096 * LV)), 256>>>(n, size, items, cursors, cap, LV); });
    time("pass A: no rank atomics (floor of the rest), no stores", [&] { pass_a<3><<<ga, 256>>>(n, size, items, cursors, cap); });
    time("pass B: 4 ds_add_u64 fixed point per item, 1024 threads, 128 KB", [&] { pass_b64<0><<<gb, 1024>>>(size, items, cursors, cap, table); });
    time("pass B: 4 ds_add_f64 per item, 1024 threads, 128 KB (as built)", [&] { pass_b64<1><<<gb, 1024>>>(size, items, cursors, cap, table); });
    time("pass B: ds_add_f32, 1024 threads", [&] { pass_b<1, 1024><<<gb, 1024>>>(size, items, cursors, cap, table); });
    time("pass B: ds_add_f32, 256 threads", [&] { pass_b<1, 256><<<gb, 256>>>(size, items, cursors, cap, table); });
    return 0;
}
