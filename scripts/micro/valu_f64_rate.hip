// valu_f64_rate.hip - issue cost of fp64 VALU instructions of gfx950 on one SIMD, independent chains (8 per lane), one wave per
// SIMD and three waves per SIMD: v_fma_f64, v_add_f64, v_mul_f64, the 32-bit DPP move, a division and a square root as the
// compiler expands them.  Decides what a VALU instruction of k_lin_lm costs.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void rate(double* out, int n, long long* ticks) {
    double x[8];
    for (int k = 0; k < 8; ++k) x[k] = out[threadIdx.x + 64 * k] + 1.0 + k;
    const double a = out[threadIdx.x] + 1.000001, b = out[threadIdx.x + 64] + 1e-9;
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (OP == 0) x[k] = __builtin_fma(x[k], a, b);
            if (OP == 1) x[k] = x[k] + b;
            if (OP == 2) x[k] = x[k] * a;
            if (OP == 3) {
                int lo = __double2loint(x[k]);
                lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false);
                x[k] = __hiloint2double(__double2hiint(x[k]), lo);
            }
            if (OP == 4) x[k] = a / (x[k] + 2.0);
            if (OP == 5) x[k] = sqrt(x[k] + 2.0);
        }
    }
    const long long c1 = clock64();
    double s = 0;
    for (int k = 0; k < 8; ++k) s += x[k];
    out[1024 + threadIdx.x + 256 * blockIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 16 + (threadIdx.x >> 6)] = c1 - c0;
}
int main() {
    double* d;
    long long* t;
    hipMalloc(&d, 8 * 65536);
    hipMalloc(&t, 8 * 1024);
    hipMemset(d, 0, 8 * 65536);
    const int N = 4000;
    const char* names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_mov_b32 dpp", "division (a / x)", "sqrt"};
    for (int lanes : {256, 768}) {  // 1 or 3 waves per SIMD on ONE CU
        for (int op = 0; op < 6; ++op) {
            for (int rep = 0; rep < 2; ++rep) {
                switch (op) {
                    case 0: hipLaunchKernelGGL(rate<0>, dim3(1), dim3(lanes), 0, 0, d, N, t); break;
                    case 1: hipLaunchKernelGGL(rate<1>, dim3(1), dim3(lanes), 0, 0, d, N, t); break;
                    case 2: hipLaunchKernelGGL(rate<2>, dim3(1), dim3(lanes), 0, 0, d, N, t); break;
                    case 3: hipLaunchKernelGGL(rate<3>, dim3(1), dim3(lanes), 0, 0, d, N, t); break;
                    case 4: hipLaunchKernelGGL(rate<4>, dim3(1), dim3(lanes), 0, 0, d, N, t); break;
                    default: hipLaunchKernelGGL(rate<5>, dim3(1), dim3(lanes), 0, 0, d, N, t); break;
                }
                hipDeviceSynchronize();
            }
            long long h[16];
            hipMemcpy(h, t, 8 * 16, hipMemcpyDeviceToHost);
            long long mx = 0;
            for (int w = 0; w < lanes / 64; ++w) mx = h[w] > mx ? h[w] : mx;
            // SIMD cycles per wave-instruction = (slowest wave's cycles) / (N * 8) / (waves per SIMD)
            const int wps = lanes / 256;
            std::printf("%d wave(s) per SIMD: %-18s %6.2f shader cycles per operation and wave (%.2f of SIMD time each)\n", wps, names[op], (double)mx / ((double)N * 8), (double)mx / ((double)N * 8) / wps);
        }
    }
    return 0;
}
