#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
__global__ void k(const int* cells, const int* validv, const float* vin, float* vout, int* tailout, int* headout) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = validv[i] != 0;
    const int cell = cells[i];
    bool head = true;
    {
        const int px = __shfl_up(cell, 1, 64);
        const int pv = __shfl_up((int)valid, 1, 64);
        if (lane > 0 && pv && valid && px == cell) head = false;
    }
    float v[4];
    for (int q = 0; q < 4; ++q) v[q] = valid ? vin[i] * (q + 1) : 0.f;
    bool f = head;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int fu = __shfl_up((int)f, d, 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float t = __shfl_up(v[q], d, 64);
            if (lane >= d && !f) v[q] += t;
        }
        if (lane >= d) f = f || (fu != 0);
    }
    const unsigned long long brk = __ballot(head || !valid);
    const bool tail = valid && (lane == 63 || ((brk >> (lane + 1)) & 1ull));
    vout[i] = v[0]; tailout[i] = tail; headout[i] = head;
}
int main() {
    const int N = 256;
    std::vector<int> cells(N), valid(N, 1); std::vector<float> v(N);
    srand(1);
    int c = 0;
    for (int i = 0; i < N; ++i) { if (rand() % 2) c++; cells[i] = c; v[i] = (rand() % 1000) / 1000.f; }
    for (int i = 250; i < N; ++i) valid[i] = 0;
    int *dc, *dv, *dt, *dh; float *din, *dout;
    hipMalloc(&dc, N * 4); hipMalloc(&dv, N * 4); hipMalloc(&dt, N * 4); hipMalloc(&dh, N * 4); hipMalloc(&din, N * 4); hipMalloc(&dout, N * 4);
    hipMemcpy(dc, cells.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(dv, valid.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(din, v.data(), N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dc, dv, din, dout, dt, dh);
    std::vector<float> out(N); std::vector<int> tail(N), head(N);
    hipMemcpy(out.data(), dout, N * 4, hipMemcpyDeviceToHost); hipMemcpy(tail.data(), dt, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(head.data(), dh, N * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w) {
        for (int i = w * 64; i < w * 64 + 64; ++i) {
            if (!valid[i]) continue;
            // expected: sum over run within wave ending at i
            float s = 0; int j = i;
            while (j >= w * 64 && cells[j] == cells[i] && valid[j]) { s += v[j]; --j; }
            bool etail = (i == w * 64 + 63) || !valid[i + 1] || cells[i + 1] != cells[i];
            if (fabsf(out[i] - s) > 1e-5 || (int)etail != tail[i]) { if (bad < 10) printf("lane %d cell %d: got %f exp %f tail %d exp %d head %d\n", i, cells[i], out[i], s, tail[i], (int)etail, head[i]); bad++; }
        }
    }
    printf("bad=%d\n", bad);
    return 0;
}
