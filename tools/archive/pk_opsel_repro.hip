// Minimal reproducer for the fault behind -DMNF_PK=1 (csrc/field_dev.h): a packed-fp32 VALU instruction whose first source is an SGPR
// PAIR with op_sel_hi = 0 (broadcast of the pair's LOW dword to both halves), while the pair's HIGH dword holds unrelated data.
// hipcc 7.2 emits exactly that for  fma(f32x2{s, s}, f32x2{x, y}, 0.5)  with a wave-uniform s:
//     v_pk_fma_f32 v[a:b], s[N:N+1], v[c:d], 0.5 op_sel_hi:[0,1,0]
// and on MI355X lanes 48..63 of occasional waves come out as if the high half had been taken from s[N+1].
// The kernel runs that instruction (inline asm, exact encoding) in a loop beside memory traffic and matrix instructions of the SIMD's
// other wave and counts, per lane, results that differ from the scalar fma — and how many of those equal fma(s[N+1], y, 0.5).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_opsel_repro tools/pk_opsel_repro.hip && /tmp/pk_opsel_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>   // 0: op_sel_hi broadcast, stale high dword (the failing form); 1: both dwords = scale, no broadcast; 2: scale pair in VGPRs
__global__ __launch_bounds__(512, 2) void repro(const float* __restrict__ scales, const uint32_t* __restrict__ junk, const float* __restrict__ table,
                                                 int iters, unsigned long long* __restrict__ bad_lane, unsigned long long* __restrict__ bad_as_high,
                                                 float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float x = (float)(tid % 977) * 1.0e-3f, y = (float)(tid % 613) * 1.3e-3f;
    f32x16 acc = {0};
    half8 a = {(_Float16)1, (_Float16)2, (_Float16)3, (_Float16)4, (_Float16)5, (_Float16)6, (_Float16)7, (_Float16)8}, b = a;
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        if ((wave & 1) == 0) {
            // wave-uniform scale from memory (scalar load), junk for the pair's high dword
            const int k = __builtin_amdgcn_readfirstlane((it * 7 + wave * 3 + (int)blockIdx.x) & 1023);
            typedef const float __attribute__((address_space(4))) *CF; typedef const uint32_t __attribute__((address_space(4))) *CU;
            const float s = ((CF)(uintptr_t)scales)[k];
            const uint32_t hi = ((CU)(uintptr_t)junk)[k];
            uint32_t sb; memcpy(&sb, &s, 4);
            f32x2 xy = {x, y}, out;
            if (VARIANT == 0) {
                const uint64_t pair = ((uint64_t)hi << 32) | sb;
                asm volatile("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[0,1,0]" : "=v"(out) : "s"(pair), "v"(xy));
            } else if (VARIANT == 1) {
                const uint64_t pair = ((uint64_t)sb << 32) | sb;
                asm volatile("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0]" : "=v"(out) : "s"(pair), "v"(xy));
            } else {
                f32x2 sv = {s, s};
                asm volatile("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0]" : "=v"(out) : "v"(sv), "v"(xy));
            }
            const float e0 = __builtin_fmaf(s, x, 0.5f), e1 = __builtin_fmaf(s, y, 0.5f);
            if (out.x != e0 || out.y != e1) {
                atomicAdd(&bad_lane[lane], 1ull);
                float hf; memcpy(&hf, &hi, 4);
                if (out.y == __builtin_fmaf(hf, y, 0.5f) || out.x == __builtin_fmaf(hf, x, 0.5f)) atomicAdd(&bad_as_high[lane], 1ull);
            }
            // memory traffic like the gather it sits in
            keep += table[(tid * 2654435761u + it * 40503u) & ((1u << 22) - 1)];
            x = x * 0.999f + 1.0e-4f; y = y * 0.998f + 2.0e-4f;
        } else {
            // the SIMD's other wave: matrix instructions + loads
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            keep += table[(tid * 40503u + it * 2654435761u) & ((1u << 22) - 1)];
        }
    }
    if (keep + acc[0] == 1.2345e30f) sink[0] = keep;
}

template <int V>
void run(const char* name, const float* sc, const uint32_t* jk, const float* tb, unsigned long long* bl, unsigned long long* bh, float* sink) {
    hipMemset(bl, 0, 64 * 8); hipMemset(bh, 0, 64 * 8);
    for (int r = 0; r < 5; ++r) repro<V><<<1024, 512>>>(sc, jk, tb, 2000, bl, bh, sink);
    hipDeviceSynchronize();
    unsigned long long hl[64], hh[64];
    hipMemcpy(hl, bl, sizeof(hl), hipMemcpyDeviceToHost); hipMemcpy(hh, bh, sizeof(hh), hipMemcpyDeviceToHost);
    unsigned long long q[4] = {0, 0, 0, 0}, tot = 0, th = 0;
    for (int l = 0; l < 64; ++l) { q[l / 16] += hl[l]; tot += hl[l]; th += hh[l]; }
    printf("%-64s wrong results %llu of %.3g (lanes 0-15: %llu, 16-31: %llu, 32-47: %llu, 48-63: %llu); equal to the value computed from the pair's HIGH dword: %llu\n", name, tot,
           5.0 * 1024 * 256 * 2000, q[0], q[1], q[2], q[3], th);
}

int main() {
    float* sc; uint32_t* jk; float* tb; unsigned long long *bl, *bh; float* sink;
    hipMalloc(&sc, 4096); hipMalloc(&jk, 4096); hipMalloc(&tb, 16u << 20); hipMalloc(&bl, 512); hipMalloc(&bh, 512); hipMalloc(&sink, 64);
    float hs[1024]; uint32_t hj[1024];
    for (int i = 0; i < 1024; ++i) { hs[i] = 15.0f + 0.37f * i; float g = -3.0f - 0.11f * i; memcpy(&hj[i], &g, 4); }
    hipMemcpy(sc, hs, 4096, hipMemcpyHostToDevice); hipMemcpy(jk, hj, 4096, hipMemcpyHostToDevice); hipMemset(tb, 0, 16u << 20);
    run<0>("SGPR pair, op_sel_hi:[0,1,0], stale high dword (hipcc's form)", sc, jk, tb, bl, bh, sink);
    run<1>("SGPR pair, both dwords = scale, op_sel_hi:[1,1,0]", sc, jk, tb, bl, bh, sink);
    run<2>("scale pair in VGPRs", sc, jk, tb, bl, bh, sink);
    return 0;
}
