// Checks kba::block_sum28 (reduce-scatter with v_permlane32_swap / v_permlane16_swap / DPP) against a host sum.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/reduce28_check.hip -o /tmp/r28 && /tmp/r28
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../../limo_amd/csrc/kba_kernels.hip"

__global__ void k_check(const double* in, double* out) {
    __shared__ double lds[4 * 28];
    double v[28];
    for (int i = 0; i < 28; ++i) v[i] = in[(size_t)(blockIdx.x * 256 + threadIdx.x) * 28 + i];
    kba::block_sum28(v, lds, out + blockIdx.x * 28);
}

int main() {
    const int nb = 7;
    std::vector<double> h((size_t)nb * 256 * 28);
    for (size_t i = 0; i < h.size(); ++i) h[i] = std::sin(0.37 * (double)i) * (1.0 + (double)(i % 13));
    double *d_in, *d_out;
    hipMalloc(&d_in, h.size() * 8);
    hipMalloc(&d_out, nb * 28 * 8);
    hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(nb), dim3(256), 0, 0, d_in, d_out);
    std::vector<double> o(nb * 28);
    if (hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    double worst = 0.0;
    for (int b = 0; b < nb; ++b)
        for (int i = 0; i < 28; ++i) {
            long double s = 0.0L;
            double mag = 0.0;
            for (int t = 0; t < 256; ++t) {
                s += h[(size_t)(b * 256 + t) * 28 + i];
                mag += std::fabs(h[(size_t)(b * 256 + t) * 28 + i]);
            }
            worst = std::fmax(worst, std::fabs((double)s - o[b * 28 + i]) / mag);
        }
    std::printf("reduce28: worst |gpu - host| / sum|x| = %.3e  %s\n", worst, worst < 1e-14 ? "OK" : "MISMATCH");
    return worst < 1e-14 ? 0 : 1;
}
