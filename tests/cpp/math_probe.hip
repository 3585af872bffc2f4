// math_probe.hip - TEST INFRASTRUCTURE (not part of the product library): runs the device forms of kba_math.hpp's reciprocal helpers
// on arrays of operands so that tests/test_gpu_math.py can hold them against IEEE arithmetic.
//
// Since round 5 the solve's inner loops take 1 / x, 1 / sqrt(x) and the pivots of the 3 x 3 landmark factor from the hardware seeds
// (v_rcp_f64 / v_rsq_f64) + Newton steps instead of IEEE division / square root (kba_math.hpp:rcp_nr, rsqrt_nr, chol3_inv).  They
// replace the divisions of ReprojectionErrorWithQuaternions (cost_functors_ceres.hpp:116-152), the Cauchy corrector's square root
// (Ceres 1.13 corrector.cc) and the Cholesky pivots of the SchurEliminator's 3 x 3 blocks.  The CPU-tier emulation keeps IEEE, so
// only a test on the device sees what the kernels really compute.
#include <hip/hip_runtime.h>

#include "../../limo_amd/csrc/kba_math.hpp"

namespace {
__global__ void k_probe_rcp(const double* x, double* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = kba::rcp_nr(x[i]);
}
__global__ void k_probe_rsqrt(const double* x, double* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = kba::rsqrt_nr(x[i]);
}
// A: [n][6] upper triangle (a00 a01 a02 a11 a12 a22), Li: [n][6], ok: [n]
__global__ void k_probe_chol3(const double* A, double* Li, int* ok, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a[6], l[6];
    for (int k = 0; k < 6; ++k) a[k] = A[6 * i + k];
    ok[i] = kba::chol3_inv(a, l) ? 1 : 0;
    for (int k = 0; k < 6; ++k) Li[6 * i + k] = l[k];
}
// xn, yn, 1/z of view_xy (the projection every consumer of the factored planes rebuilds): vl [n][12], p [n][3] -> out [n][4] (xn, yn, iz, ok)
__global__ void k_probe_view_xy(const double* vl, const double* p, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double xn, yn, iz;
    const bool ok = kba::view_xy(vl + 12 * i, p + 3 * i, &xn, &yn, &iz);
    out[4 * i] = xn;
    out[4 * i + 1] = yn;
    out[4 * i + 2] = iz;
    out[4 * i + 3] = ok ? 1.0 : 0.0;
}

template <class F>
int on_device(const void* const* in, const size_t* in_bytes, int n_in, void* const* out, const size_t* out_bytes, int n_out, F launch) {
    void* d[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = 0;
    for (int i = 0; i < n_in + n_out && !rc; ++i) {
        const size_t b = i < n_in ? in_bytes[i] : out_bytes[i - n_in];
        if (hipMalloc(&d[i], b ? b : 8) != hipSuccess) rc = 1;
    }
    for (int i = 0; i < n_in && !rc; ++i)
        if (hipMemcpy(d[i], in[i], in_bytes[i], hipMemcpyHostToDevice) != hipSuccess) rc = 2;
    if (!rc) {
        launch(d);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = 3;
    }
    for (int i = 0; i < n_out && !rc; ++i)
        if (hipMemcpy(out[i], d[n_in + i], out_bytes[i], hipMemcpyDeviceToHost) != hipSuccess) rc = 4;
    for (int i = 0; i < n_in + n_out; ++i)
        if (d[i]) (void)hipFree(d[i]);
    return rc;
}
}  // namespace

extern "C" {
// op 0: rcp_nr, 1: rsqrt_nr
int probe_unary(int op, const double* x, double* y, int n) {
    const void* in[1] = {x};
    const size_t ib[1] = {sizeof(double) * n};
    void* out[1] = {y};
    const size_t ob[1] = {sizeof(double) * n};
    return on_device(in, ib, 1, out, ob, 1, [&](void** d) {
        if (op == 0)
            hipLaunchKernelGGL(k_probe_rcp, dim3((n + 255) / 256), dim3(256), 0, 0, (const double*)d[0], (double*)d[1], n);
        else
            hipLaunchKernelGGL(k_probe_rsqrt, dim3((n + 255) / 256), dim3(256), 0, 0, (const double*)d[0], (double*)d[1], n);
    });
}
int probe_chol3(const double* A, double* Li, int* ok, int n) {
    const void* in[1] = {A};
    const size_t ib[1] = {sizeof(double) * 6 * n};
    void* out[2] = {Li, ok};
    const size_t ob[2] = {sizeof(double) * 6 * n, sizeof(int) * n};
    return on_device(in, ib, 1, out, ob, 2,
                     [&](void** d) { hipLaunchKernelGGL(k_probe_chol3, dim3((n + 255) / 256), dim3(256), 0, 0, (const double*)d[0], (double*)d[1], (int*)d[2], n); });
}
int probe_view_xy(const double* vl, const double* p, double* out4, int n) {
    const void* in[2] = {vl, p};
    const size_t ib[2] = {sizeof(double) * 12 * n, sizeof(double) * 3 * n};
    void* out[1] = {out4};
    const size_t ob[1] = {sizeof(double) * 4 * n};
    return on_device(in, ib, 2, out, ob, 1,
                     [&](void** d) { hipLaunchKernelGGL(k_probe_view_xy, dim3((n + 255) / 256), dim3(256), 0, 0, (const double*)d[0], (const double*)d[1], (double*)d[2], n); });
}
}
