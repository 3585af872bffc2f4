// grid_barrier.hip - what a device-wide barrier costs on gfx950 (8 XCDs, one L2 each): G workgroups of 256 lanes meet N
// times; between two meetings every workgroup writes a word its right-hand neighbour reads after the barrier (checked), so
// the fences are the ones a real phase boundary needs.  Variants: cooperative launch or plain launch; agent-scope fence
// (L2 write-back + invalidate across XCDs) around the arrival counter.  Decides whether one LM iteration of a single
// window can live in ONE launch with barriers between its phases instead of ten launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool grid_sync(int* bar, int n_wg, int& gen) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        const int arrived = __hip_atomic_fetch_add(&bar[0], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == n_wg - 1) {
            __hip_atomic_store(&bar[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&bar[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(&bar[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 100000000ll) {  // 1 s: give up instead of hanging the GPU
                    ok = false;
                    break;
                }
            }
        }
        __threadfence();
    }
    ++gen;
    return __syncthreads_and(ok);
}

__global__ __launch_bounds__(256) void k_meet(int* bar, int* words, int n, int* errors, long long* ticks) {
    int gen = 0;
    const int g = blockIdx.x, G = gridDim.x;
    const long long t0 = wall_clock64();
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        if (threadIdx.x == 0) words[g * 64] = i * 1000 + g;
        if (!grid_sync(bar, G, gen)) {
            if (threadIdx.x == 0) atomicAdd(errors + 1, 1);
            return;
        }
        if (threadIdx.x == 0) {
            const int nb = (g + 1) % G;
            if (words[nb * 64] != i * 1000 + nb) ++bad;
        }
        if (!grid_sync(bar, G, gen)) return;  // (nobody overwrites a word before its reader has seen it)
    }
    if (threadIdx.x == 0) {
        if (bad) atomicAdd(errors, bad);
        ticks[g] = wall_clock64() - t0;
    }
}

int main() {
    int *bar, *words, *errors;
    long long* ticks;
    hipMalloc(&bar, 64);
    hipMalloc(&words, 4 * 64 * 512);
    hipMalloc(&errors, 8);
    hipMalloc(&ticks, 8 * 512);
    const int N = 2000;
    for (int coop = 0; coop < 2; ++coop)
        for (int G : {1, 2, 4, 8, 9, 16, 32, 64, 128}) {
            double best = 1e30;
            int herr[2] = {0, 0};
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(bar, 0, 64);
                hipMemset(errors, 0, 8);
                hipMemset(words, 0xFF, 4 * 64 * 512);
                int n = N;
                void* args[] = {&bar, &words, &n, &errors, &ticks};
                hipError_t e = coop ? hipLaunchCooperativeKernel((const void*)k_meet, dim3(G), dim3(256), args, 0, 0)
                                    : hipLaunchKernel((const void*)k_meet, dim3(G), dim3(256), args, 0, 0);
                if (e != hipSuccess) {
                    std::printf("launch failed: %s\n", hipGetErrorString(e));
                    return 1;
                }
                hipDeviceSynchronize();
                std::vector<long long> h(G);
                hipMemcpy(h.data(), ticks, 8 * G, hipMemcpyDeviceToHost);
                hipMemcpy(herr, errors, 8, hipMemcpyDeviceToHost);
                long long mx = 0;
                for (long long v : h) mx = v > mx ? v : mx;
                best = std::min(best, (double)mx / (2.0 * N) * 0.01);  // 100 MHz ticks -> us per barrier
            }
            std::printf("%s launch, %3d workgroups: %.2f us per barrier, neighbour words wrong %d, timeouts %d\n", coop ? "cooperative" : "plain      ", G, best, herr[0], herr[1]);
        }
    return 0;
}
