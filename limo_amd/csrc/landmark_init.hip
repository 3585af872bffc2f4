// landmark_init.hip — limo_landmark_init on gfx950: one lane per landmark over the CSR ray list (landmark_init.hpp).
// A keyframe brings a few thousand new landmarks at once (every track the frame starts); the caller hands them over
// in ONE call (limo_amd/kba push()).  HBM-side this is a 64-byte-per-ray scan; nothing to tile.
// Compiled with floating-point contraction off (pragma + -ffp-contract=off in the build): see landmark_init.hpp.
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include <cstring>
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

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

// One pinned staging block carries [ray_off | rays | use_depth] up and [positions | ok] down, one device block holds
// both: two copies and one kernel per call, all on the context's stream, blocks from the context's pools (a keyframe
// calls this once; per-buffer hipMallocAsync + copies from pageable memory cost more than the kernel and, with
// ROCm 7.0, were the one place where a 4541-frame drive was not reproducible from run to run).
extern "C" int limo_landmark_init(limo_ctx* ctx, int32_t n, const int32_t* ray_off, const limo_ray* rays,
                                  const uint8_t* use_depth, double* pos_out, uint8_t* ok) {
    if (!ctx) return LIMO_ERR_INVALID;  // device work: needs a context (no host fallback)
    if (n < 0 || (n > 0 && (!ray_off || !rays || !use_depth || !pos_out || !ok))) return LIMO_ERR_INVALID;
    if (n == 0) return LIMO_OK;
    if (ray_off[0] != 0) {  // CSR base: the kernel indexes rays[ray_off[i] ..) on the device
        ctx->err = "limo_landmark_init: ray_off[0] must be 0";
        return LIMO_ERR_INVALID;
    }
    for (int i = 0; i < n; ++i)
        if (ray_off[i + 1] < ray_off[i]) {
            ctx->err = "limo_landmark_init: ray_off must be non-decreasing";
            return LIMO_ERR_INVALID;
        }
    if (hipSetDevice(ctx->device) != hipSuccess) return LIMO_ERR_NO_DEVICE;
    const size_t n_rays = (size_t)ray_off[n];
    const size_t b_off = sizeof(int32_t) * (n + 1), b_rays = sizeof(limo_ray) * n_rays, b_use = (size_t)n;
    const size_t o_rays = align256(b_off), o_use = o_rays + align256(b_rays), in_bytes = o_use + align256(b_use);
    const size_t b_pos = sizeof(double) * 3 * n, o_ok = align256(b_pos), out_bytes = o_ok + align256((size_t)n);
    const size_t total = in_bytes + out_bytes;
    int rc = LIMO_OK;
    char *h = nullptr, *d = nullptr;
    hipStream_t s = ctx->stream;
    LI_TRY(ctx->host_alloc((void**)&h, total));
    LI_TRY(ctx->pool_alloc((void**)&d, total));
    std::memcpy(h, ray_off, b_off);
    if (b_rays) std::memcpy(h + o_rays, rays, b_rays);
    std::memcpy(h + o_use, use_depth, b_use);
    LI_TRY(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_landmark_init, dim3((n + 127) / 128), dim3(128), 0, s, n, (const int32_t*)d, (const limo_ray*)(d + o_rays),
                       (const uint8_t*)(d + o_use), (double*)(d + in_bytes), (uint8_t*)(d + in_bytes + o_ok));
    LI_TRY(hipGetLastError());
    LI_TRY(hipMemcpyAsync(h + in_bytes, d + in_bytes, out_bytes, hipMemcpyDeviceToHost, s));
    LI_TRY(hipStreamSynchronize(s));
    std::memcpy(pos_out, h + in_bytes, b_pos);
    std::memcpy(ok, h + in_bytes + o_ok, (size_t)n);
done:
    // an error after the upload / launch was queued: the blocks may still be in flight - drain the stream before they
    // go back to the pools (the next call would otherwise reuse memory the device is still reading or writing)
    if (rc != LIMO_OK) (void)hipStreamSynchronize(s);
    if (h) ctx->host_free(h, total);
    if (d) ctx->pool_free(d, total);
    return rc;
}
