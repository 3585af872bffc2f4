// mfma_valu_overlap.hip - do fp64 MFMA and fp64 VALU work of ONE wave overlap on gfx950?  Per loop iteration one
// v_mfma_f64_16x16x4 (64 cycles of the matrix pipe) and K independent v_fma_f64 (K = 0, 6, 12, 24).  If the two pipes run side by
// side the iteration stays at ~64 cycles until the VALU work exceeds it; if they share the fp64 units it is 64 + 5 K.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int K, bool MFMA>
__global__ void probe(double* out, int n, long long* ticks) {
    const double a = out[threadIdx.x] + 1.0, b = out[64 + threadIdx.x] + 1e-9;
    v4f64 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double x[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) x[k] = a + k;
    const long long c0 = clock64();
    for (int i = 0; i < n; i += 4) {  // four independent accumulators (static indices), K FMAs behind every MFMA
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MFMA) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[q], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) x[k] = __builtin_fma(x[k], a, b);
        }
    }
    const long long c1 = clock64();
    double s = 0;
    for (int q = 0; q < 4; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
#pragma unroll
    for (int k = 0; k < 24; ++k) s += x[k];
    out[1024 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[threadIdx.x >> 6] = c1 - c0;
}
template <int K, bool M>
static double run(double* d, long long* t, int lanes) {
    const int N = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<K, M>), dim3(1), dim3(lanes), 0, 0, d, N, t);
        hipDeviceSynchronize();
    }
    long long h[16];
    hipMemcpy(h, t, 8 * 16, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < lanes / 64; ++w) mx = h[w] > mx ? h[w] : mx;
    return (double)mx / N;
}
int main() {
    double* d;
    long long* t;
    hipMalloc(&d, 8 * 4096);
    hipMalloc(&t, 8 * 64);
    hipMemset(d, 0, 8 * 4096);
    for (int lanes : {256, 768}) {
        std::printf("%d wave(s) per SIMD, shader cycles per loop iteration and wave:\n", lanes / 256);
        std::printf("  MFMA only %.1f | 6 FMA only %.1f | 12 FMA only %.1f | 24 FMA only %.1f\n", run<0, true>(d, t, lanes), run<6, false>(d, t, lanes), run<12, false>(d, t, lanes), run<24, false>(d, t, lanes));
        std::printf("  MFMA + 6 FMA %.1f | MFMA + 12 FMA %.1f | MFMA + 24 FMA %.1f\n", run<6, true>(d, t, lanes), run<12, true>(d, t, lanes), run<24, true>(d, t, lanes));
    }
    return 0;
}
