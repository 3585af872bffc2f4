// landmark_init.hip — limo_landmark_init on gfx950: one lane per landmark over the CSR ray list (landmark_init.hpp).
// A keyframe brings a few thousand new landmarks at once (every track the frame starts); the caller hands them over
// in ONE call (limo_amd/kba push()).  HBM-side this is a 64-byte-per-ray scan; nothing to tile.
#include <hip/hip_runtime.h>

#include <string>

#include "landmark_init.hpp"
#include "limo_ctx.hpp"

namespace {

__global__ void k_landmark_init(int n, const int32_t* ray_off, const limo_ray* rays, const uint8_t* use_depth, double* pos,
                                uint8_t* ok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p[3] = {0.0, 0.0, 0.0};
    const bool good = kba::lminit_one(ray_off, rays, use_depth, i, p);
    pos[3 * (int64_t)i] = p[0];
    pos[3 * (int64_t)i + 1] = p[1];
    pos[3 * (int64_t)i + 2] = p[2];
    ok[i] = good ? 1 : 0;
}

#define LI_TRY(expr)                                                             \
    do {                                                                         \
        hipError_t e__ = (expr);                                                 \
        if (e__ != hipSuccess) {                                                 \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e__);       \
            rc = LIMO_ERR_RUNTIME;                                               \
            goto done;                                                           \
        }                                                                        \
    } while (0)

}  // namespace

extern "C" int limo_landmark_init(limo_ctx* ctx, int32_t n, const int32_t* ray_off, const limo_ray* rays,
                                  const uint8_t* use_depth, double* pos_out, uint8_t* ok) {
    if (!ctx) return LIMO_ERR_INVALID;  // device work: needs a context (no host fallback)
    if (n < 0 || (n > 0 && (!ray_off || !rays || !use_depth || !pos_out || !ok))) return LIMO_ERR_INVALID;
    if (n == 0) return LIMO_OK;
    for (int i = 0; i < n; ++i)
        if (ray_off[i + 1] < ray_off[i]) return LIMO_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return LIMO_ERR_NO_DEVICE;
    const size_t n_rays = (size_t)ray_off[n];
    int rc = LIMO_OK;
    int32_t* d_off = nullptr;
    limo_ray* d_rays = nullptr;
    uint8_t *d_use = nullptr, *d_ok = nullptr;
    double* d_pos = nullptr;
    hipStream_t s = ctx->stream;
    LI_TRY(hipMallocAsync((void**)&d_off, sizeof(int32_t) * (n + 1), s));
    LI_TRY(hipMallocAsync((void**)&d_rays, sizeof(limo_ray) * (n_rays ? n_rays : 1), s));
    LI_TRY(hipMallocAsync((void**)&d_use, (size_t)n, s));
    LI_TRY(hipMallocAsync((void**)&d_ok, (size_t)n, s));
    LI_TRY(hipMallocAsync((void**)&d_pos, sizeof(double) * 3 * n, s));
    LI_TRY(hipMemcpyAsync(d_off, ray_off, sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice, s));
    if (n_rays) LI_TRY(hipMemcpyAsync(d_rays, rays, sizeof(limo_ray) * n_rays, hipMemcpyHostToDevice, s));
    LI_TRY(hipMemcpyAsync(d_use, use_depth, (size_t)n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_landmark_init, dim3((n + 127) / 128), dim3(128), 0, s, n, d_off, d_rays, d_use, d_pos, d_ok);
    LI_TRY(hipGetLastError());
    LI_TRY(hipMemcpyAsync(pos_out, d_pos, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, s));
    LI_TRY(hipMemcpyAsync(ok, d_ok, (size_t)n, hipMemcpyDeviceToHost, s));
    LI_TRY(hipStreamSynchronize(s));
done:
    if (d_off) (void)hipFreeAsync(d_off, s);
    if (d_rays) (void)hipFreeAsync(d_rays, s);
    if (d_use) (void)hipFreeAsync(d_use, s);
    if (d_ok) (void)hipFreeAsync(d_ok, s);
    if (d_pos) (void)hipFreeAsync(d_pos, s);
    return rc;
}
