// mfma_f64_rate.hip - issue cost of the two fp64 MFMA shapes of gfx950 on one SIMD: v_mfma_f64_16x16x4_f64 (2048 FLOP) and
// v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4 = 512 FLOP), independent accumulators, one wave (and four waves on one CU's
// four SIMDs).  Decides whether a 4x4-tiled Schur kernel (fewer padded columns) can beat the 16x16 tiling.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void rate(double* out, int n, long long* ticks) {
    const double a = out[threadIdx.x], b = out[64 + threadIdx.x];
    v4f64 c16[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double c4[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (SHAPE == 16) {
#pragma unroll
            for (int k = 0; k < 4; ++k) c16[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c16[k], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) c4[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4[k], 0, 0, 0);
        }
    }
    const long long c1 = clock64();
    double s = 0;
    for (int k = 0; k < 4; ++k) s += c16[k][0] + c16[k][1] + c16[k][2] + c16[k][3];
    for (int k = 0; k < 8; ++k) s += c4[k];
    out[128 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = c1 - c0;
}
int main() {
    double* d;
    long long* t;
    hipMalloc(&d, 8 * 1024);
    hipMalloc(&t, 8 * 64);
    hipMemset(d, 0, 8 * 1024);
    const int N = 20000;
    for (int waves : {64, 256}) {
        for (int shape : {16, 4}) {
            for (int rep = 0; rep < 3; ++rep) {
                if (shape == 16)
                    hipLaunchKernelGGL(rate<16>, dim3(1), dim3(waves), 0, 0, d, N, t);
                else
                    hipLaunchKernelGGL(rate<4>, dim3(1), dim3(waves), 0, 0, d, N, t);
                hipDeviceSynchronize();
            }
            long long h = 0;
            hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
            const int per_iter = shape == 16 ? 4 : 8;
            const double cyc = (double)h / ((double)N * per_iter);
            const double flop = shape == 16 ? 2048.0 : 512.0;
            std::printf("%3d lanes in the workgroup (%d wave(s) per SIMD): mfma_f64_%s  %.1f shader cycles per instruction per wave -> %.1f FLOP / cycle / SIMD\n", waves, waves / 256 + (waves < 256),
                        shape == 16 ? "16x16x4" : "4x4x4 (4 blocks)", cyc, flop / cyc);
        }
    }
    return 0;
}
