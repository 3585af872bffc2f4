// do the landmark-major gather kernels (k_backsub, k_lm_accum) read faster from ONE interleaved
// 64-byte record per observation (4 x global_load_dwordx4) than from the 7 fp64 planes they read today
// (7 x global_load_dwordx2 out of 7 streams per view)?  Same access pattern as the solve: 1024 windows x 2000 landmarks,
// 5 views, 88 % of the (landmark, view) pairs observed, observations stored view-major in landmark order, lane = landmark.
//   producer: lane = observation, writes the 7 values (planes vs. one record)
//   consumer: lane = landmark, loops over its views, reads the 7 values through the slot table, 14 FMAs, 3 doubles out
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/gather_record.hip -o /tmp/gr && /tmp/gr
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kViews = 5;

__global__ __launch_bounds__(256) void produce_planes(double* planes, size_t n, const float* src) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double s = src[i];
#pragma unroll
    for (int k = 0; k < 7; ++k) planes[(size_t)k * n + i] = s * (k + 1);
}
__global__ __launch_bounds__(256) void produce_records(double* rec, size_t n, const float* src) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double s = src[i];
    double4* r = reinterpret_cast<double4*>(rec + 8 * i);
    r[0] = {s, 2 * s, 3 * s, 4 * s};
    r[1] = {5 * s, 6 * s, 7 * s, 0.0};
}
// 56-byte records, no padding: 7 doubles per observation back to back (dwordx2 loads, 3.5 per cache line of 128 B)
__global__ __launch_bounds__(256) void produce_packed(double* rec, size_t n, const float* src) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double s = src[i];
#pragma unroll
    for (int k = 0; k < 7; ++k) rec[7 * i + k] = s * (k + 1);
}

template <int MODE>  // 0 planes, 1 records of 64 B, 2 packed records of 56 B
__global__ __launch_bounds__(256) void consume(const double* data, size_t n, const int* slot, int n_lm, double* out) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= n_lm) return;
    double a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
    for (int v = 0; v < kViews; ++v) {
        const int o = slot[(size_t)v * n_lm + l];
        if (o < 0) continue;
        double r[7];
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 7; ++k) r[k] = data[(size_t)k * n + o];
        } else if (MODE == 1) {
            const double4* p = reinterpret_cast<const double4*>(data + 8 * (size_t)o);
            const double4 x = p[0], y = p[1];
            r[0] = x.x, r[1] = x.y, r[2] = x.z, r[3] = x.w, r[4] = y.x, r[5] = y.y, r[6] = y.z;
        } else {
#pragma unroll
            for (int k = 0; k < 7; ++k) r[k] = data[7 * (size_t)o + k];
        }
        a0 += r[0] * r[3] + r[4] * r[1];
        a1 += r[1] * r[4] + r[5] * r[2];
        a2 += r[2] * r[5] + r[6] * r[0];
    }
    out[l] = a0;
    out[(size_t)n_lm + l] = a1;
    out[2 * (size_t)n_lm + l] = a2;
}

template <typename F>
float time_ms(F&& launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int n_win = 1024, lm_per_win = 2000, n_lm = n_win * lm_per_win;
    // slot table: view v of window w holds its observed landmarks in landmark order
    std::vector<int> slot((size_t)kViews * n_lm);
    size_t n_obs = 0;
    uint64_t rng = 12345;
    auto rnd = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (double)(rng >> 11) / 9007199254740992.0; };
    for (int w = 0; w < n_win; ++w)
        for (int v = 0; v < kViews; ++v)
            for (int l = 0; l < lm_per_win; ++l) slot[(size_t)v * n_lm + (size_t)w * lm_per_win + l] = rnd() < 0.88 ? (int)n_obs++ : -1;
    std::printf("%d landmarks, %zu observations (%.1f per landmark)\n", n_lm, n_obs, (double)n_obs / n_lm);
    int* d_slot;
    float* d_src;
    double *d_planes, *d_rec, *d_packed, *d_out;
    hipMalloc(&d_slot, slot.size() * 4);
    hipMemcpy(d_slot, slot.data(), slot.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_src, n_obs * 4);
    hipMemset(d_src, 0, n_obs * 4);
    hipMalloc(&d_planes, n_obs * 7 * 8);
    hipMalloc(&d_rec, n_obs * 8 * 8);
    hipMalloc(&d_packed, n_obs * 7 * 8);
    hipMalloc(&d_out, (size_t)n_lm * 3 * 8);
    const unsigned gb_o = (unsigned)((n_obs + 255) / 256), gb_l = (unsigned)((n_lm + 255) / 256);
    const int reps = 20;
    const float p0 = time_ms([&] { hipLaunchKernelGGL(produce_planes, dim3(gb_o), dim3(256), 0, 0, d_planes, n_obs, d_src); }, reps);
    const float p1 = time_ms([&] { hipLaunchKernelGGL(produce_records, dim3(gb_o), dim3(256), 0, 0, d_rec, n_obs, d_src); }, reps);
    const float p2 = time_ms([&] { hipLaunchKernelGGL(produce_packed, dim3(gb_o), dim3(256), 0, 0, d_packed, n_obs, d_src); }, reps);
    const float c0 = time_ms([&] { hipLaunchKernelGGL(consume<0>, dim3(gb_l), dim3(256), 0, 0, d_planes, n_obs, d_slot, n_lm, d_out); }, reps);
    const float c1 = time_ms([&] { hipLaunchKernelGGL(consume<1>, dim3(gb_l), dim3(256), 0, 0, d_rec, n_obs, d_slot, n_lm, d_out); }, reps);
    const float c2 = time_ms([&] { hipLaunchKernelGGL(consume<2>, dim3(gb_l), dim3(256), 0, 0, d_packed, n_obs, d_slot, n_lm, d_out); }, reps);
    const double useful = 56.0 * n_obs;  // bytes of the 7 values
    const double idx = 4.0 * kViews * n_lm, outb = 24.0 * n_lm;
    std::printf("producer  7 planes       %7.1f us  %5.2f TB/s on the 56 B/obs written (+4 B/obs read)\n", 1e3 * p0, (useful + 4.0 * n_obs) / (p0 * 1e-3) / 1e12);
    std::printf("producer  64 B records   %7.1f us  %5.2f TB/s on 56 B useful (64 B moved: %5.2f TB/s)\n", 1e3 * p1, (useful + 4.0 * n_obs) / (p1 * 1e-3) / 1e12, (64.0 * n_obs + 4.0 * n_obs) / (p1 * 1e-3) / 1e12);
    std::printf("producer  56 B records   %7.1f us  %5.2f TB/s\n", 1e3 * p2, (useful + 4.0 * n_obs) / (p2 * 1e-3) / 1e12);
    std::printf("consumer  7 planes       %7.1f us  %5.2f TB/s on 56 B/obs + slot table + output\n", 1e3 * c0, (useful + idx + outb) / (c0 * 1e-3) / 1e12);
    std::printf("consumer  64 B records   %7.1f us  %5.2f TB/s on the same useful bytes (64 B moved: %5.2f TB/s)\n", 1e3 * c1, (useful + idx + outb) / (c1 * 1e-3) / 1e12, (64.0 * n_obs + idx + outb) / (c1 * 1e-3) / 1e12);
    std::printf("consumer  56 B records   %7.1f us  %5.2f TB/s\n", 1e3 * c2, (useful + idx + outb) / (c2 * 1e-3) / 1e12);
    return 0;
}
