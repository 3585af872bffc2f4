// store_stream.hip - what the memory system takes from a pure streaming-STORE kernel (the shape of k_evaluate: many output planes,
// a lane writes 8 or 16 bytes to each).  Prints GB/s for: one contiguous stream vs NP planes, 8- vs 16-byte stores, plain vs
// non-temporal, and a read : write mix like the materialised Jacobian pass (48 B read per 240 B written).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/store_stream.hip -o /tmp/store_stream && /tmp/store_stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double v2f64 __attribute__((ext_vector_type(2)));

template <int NP, bool NT, bool WIDE>
__global__ __launch_bounds__(256) void k_store(double* out, long long stride, long long n_items, double seed) {
    // item = one "observation pair" (WIDE: 16 B per plane) or one observation (8 B per plane)
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_items) return;
    double v = seed + (double)i;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if (WIDE) {
            v2f64 x = {v + k, v - k};
            v2f64* p = reinterpret_cast<v2f64*>(out + k * stride + 2 * i);
            if (NT) __builtin_nontemporal_store(x, p); else *p = x;
        } else {
            double* p = out + k * stride + i;
            if (NT) __builtin_nontemporal_store(v + k, p); else *p = v + k;
        }
    }
}
// the same with a read stream (6 doubles per item, 16-byte loads) in front
template <int NP>
__global__ __launch_bounds__(256) void k_mix(const double* in, double* out, long long stride, long long n_items) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_items) return;
    const v2f64* q = reinterpret_cast<const v2f64*>(in + 6 * i);
    const v2f64 a = q[0], b = q[1], c = q[2];
    const double v = a.x + a.y * b.x + b.y * c.x + c.y;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        v2f64 x = {v + k, v - k};
        __builtin_nontemporal_store(x, reinterpret_cast<v2f64*>(out + k * stride + 2 * i));
    }
}

template <class F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const long long n_obs = 9 * 1000 * 1000;  // observations (the 1024-window evaluate batch)
    const long long stride = n_obs + 1024;
    double *out, *in;
    hipMalloc(&out, sizeof(double) * stride * 32);
    hipMalloc(&in, sizeof(double) * n_obs * 6);
    hipMemset(in, 0, sizeof(double) * n_obs * 6);
    const int reps = 20;
    auto report = [&](const char* what, double bytes, double ms) { std::printf("%-72s %8.3f ms  %7.0f GB/s\n", what, ms, bytes / (ms * 1e-3) / 1e9); };
    {
        const long long n = n_obs / 2;
        const int g = (int)((n + 255) / 256);
        report("30 planes, 16-B stores, non-temporal", 30.0 * 8 * n_obs, time_ms([&] { hipLaunchKernelGGL((k_store<30, true, true>), dim3(g), dim3(256), 0, 0, out, stride, n, 1.0); }, reps));
        report("30 planes, 16-B stores, plain", 30.0 * 8 * n_obs, time_ms([&] { hipLaunchKernelGGL((k_store<30, false, true>), dim3(g), dim3(256), 0, 0, out, stride, n, 1.0); }, reps));
        report("20 planes, 16-B stores, non-temporal", 20.0 * 8 * n_obs, time_ms([&] { hipLaunchKernelGGL((k_store<20, true, true>), dim3(g), dim3(256), 0, 0, out, stride, n, 1.0); }, reps));
        report("8 planes, 16-B stores, non-temporal", 8.0 * 8 * n_obs, time_ms([&] { hipLaunchKernelGGL((k_store<8, true, true>), dim3(g), dim3(256), 0, 0, out, stride, n, 1.0); }, reps));
        report("1 plane x 30 launches-worth (one stream), 16-B stores, non-temporal", 1.0 * 8 * n_obs * 30, time_ms([&] {
                   const long long nn = n * 30;
                   hipLaunchKernelGGL((k_store<1, true, true>), dim3((int)((nn + 255) / 256)), dim3(256), 0, 0, out, stride * 32, nn, 1.0);
               }, reps));
        report("30 planes, 16-B nt stores + 48 B read per observation", (30.0 * 8 + 48) * n_obs, time_ms([&] { hipLaunchKernelGGL((k_mix<30>), dim3(g), dim3(256), 0, 0, in, out, stride, n); }, reps));
        report("25 planes, 16-B nt stores + 48 B read per observation", (25.0 * 8 + 48) * n_obs, time_ms([&] { hipLaunchKernelGGL((k_mix<25>), dim3(g), dim3(256), 0, 0, in, out, stride, n); }, reps));
    }
    {
        const long long n = n_obs;
        const int g = (int)((n + 255) / 256);
        report("30 planes, 8-B stores, non-temporal", 30.0 * 8 * n_obs, time_ms([&] { hipLaunchKernelGGL((k_store<30, true, false>), dim3(g), dim3(256), 0, 0, out, stride, n, 1.0); }, reps));
        report("30 planes, 8-B stores, plain", 30.0 * 8 * n_obs, time_ms([&] { hipLaunchKernelGGL((k_store<30, false, false>), dim3(g), dim3(256), 0, 0, out, stride, n, 1.0); }, reps));
    }
    {   // memset / copy references
        report("hipMemsetAsync of the same 30 planes", 30.0 * 8 * n_obs, time_ms([&] { hipMemsetAsync(out, 0, sizeof(double) * n_obs * 30, 0); }, reps));
        report("hipMemcpyAsync device->device, 15 planes (read + write counted)", 2 * 15.0 * 8 * n_obs, time_ms([&] { hipMemcpyAsync(out, out + stride * 16, sizeof(double) * n_obs * 15, hipMemcpyDeviceToDevice, 0); }, reps));
    }
    return 0;
}
