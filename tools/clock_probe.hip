// tools/clock_probe.hip: effective shader clock while another kernel runs.  One wave per probed compute unit samples s_memtime (shader clock) against s_memrealtime
// (constant 100 MHz) `n` times, `gap` wall ticks apart; out[block][i] = (shader ticks, wall ticks) since the probe's start.  Launched on its own stream beside the
// kernel under test (tools/exp_clock.py).  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_build/libclock_probe.so tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(64) probe_kernel(int n, long long gap, long long *out) {
    if (threadIdx.x != 0) return;
    const long long s0 = clock64(), w0 = wall_clock64();
    long long *o = out + (long long)blockIdx.x * n * 2;
    for (int i = 0; i < n; ++i) {
        long long w;
        do { __builtin_amdgcn_s_sleep(8); w = wall_clock64(); } while (w - w0 < gap * (i + 1));
        o[2 * i] = clock64() - s0; o[2 * i + 1] = w - w0;
    }
}

extern "C" int clock_probe_launch(int blocks, int n, long long gap, long long *out, void *stream) {
    hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, n, gap, out);
    return (int)hipGetLastError();
}
