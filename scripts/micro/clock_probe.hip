// clock_probe.hip - does a short single-wave kernel run at the full shader clock?  A chain of N dependent fp64 FMAs by
// one wave is timed (a) on an otherwise idle GPU, launch after launch, (b) while a second stream keeps all CUs busy.
// s_memtime (constant 100 MHz on gfx9) next to the chain gives the wall time inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void chain(double* out, int n, long long* ticks) {
    double x = out[0];
    const long long t0 = wall_clock64(), c0 = clock64();
#pragma unroll 16
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
    const long long t1 = wall_clock64(), c1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) {
        ticks[0] = t1 - t0;
        ticks[1] = c1 - c0;
    }
}
__global__ void burn(double* out, int n) {
    double x = out[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
    double *d, *d2;
    long long* t;
    hipMalloc(&d, 64 * 8);
    hipMalloc(&d2, 8 * 256 * 1024 * 4);
    hipMalloc(&t, 16);
    hipMemset(d, 0, 64 * 8);
    hipMemset(d2, 0, 8 * 256 * 1024 * 4);
    hipStream_t s1, s2;
    hipStreamCreate(&s1);
    hipStreamCreate(&s2);
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    const int N = 20000;
    auto run = [&](const char* name, bool load) {
        std::vector<double> us, mhz;
        for (int r = 0; r < 40; ++r) {
            if (load) hipLaunchKernelGGL(burn, dim3(1024 * 4), dim3(256), 0, s2, d2, 400000);
            hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, N, t);
            hipStreamSynchronize(s1);
            long long h[2];
            hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            us.push_back(1e3 * (double)h[0] / rate);  // rate in kHz
            mhz.push_back((double)h[1] / (1e3 * (double)h[0] / rate));
            if (load) hipStreamSynchronize(s2);
        }
        std::sort(us.begin(), us.end());
        std::sort(mhz.begin(), mhz.end());
        std::printf("%-28s chain of %d dependent fp64 FMAs: median %.1f us, min %.1f us -> %.2f ns per FMA; clock64 / wall clock = %.0f MHz (median)\n", name, N,
                    us[us.size() / 2], us[0], 1e3 * us[us.size() / 2] / N, mhz[mhz.size() / 2]);
    };
    run("idle GPU, back to back", false);
    run("GPU saturated by a 2nd stream", true);
    run("idle again", false);
    return 0;
}
