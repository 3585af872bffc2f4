// limo_ctx.hpp — the context object behind `limo_ctx*` (one per thread / GPU / stream), shared by the translation
// units that implement the C-ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "kba_pack.hpp"

struct limo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own = nullptr;
    std::string err;
    void* depth_ws = nullptr;               // workspace of limo_depth_estimate (depth.hip), grown on demand
    void (*depth_ws_free)(void*) = nullptr;
    void* comm = nullptr;                   // ncclComm_t of a landmark-sharded solve (limo_ctx_comm_init), else null
    int comm_rank = 0, comm_world = 1;
    // ... or a transport of the caller's (limo_ctx_comm_init_host): the exchange steps are staged through pinned host memory and
    // handed to this callback - how two processes that share ONE GPU (RCCL refuses a device twice in a communicator) or a host-side
    // fabric run the landmark-sharded solve with world > 1
    void (*xfn)(const double* send, double* recv, long long count, int kind, void* user) = nullptr;
    void* xuser = nullptr;
    double* xhost = nullptr;  // pinned staging: [1 + world][count]
    size_t xhost_cap = 0;
    bool has_transport() const { return comm != nullptr || xfn != nullptr; }
    long long exchange_stats[3] = {0, 0, 0};  // last landmark-sharded solve: exchange steps, bytes per rank, LM iterations
    long long coop_fallbacks = 0;           // one-launch solves whose barrier timed out and that were redone as a launch sequence
    static constexpr int kCoopRetryAfter = 64;  // solves through the launch sequence before a benched one-launch path is tried again
    int coop_benched = 0;                   // ... counted here
    int coop_strikes = 0;                   // ... in a row: after three the context stops taking the one-launch path (something shares the GPU)
    // Device blocks released by finished batches, kept for the next one (size class = power of two): a single-window
    // call (limo_ba_solve, limo_ba_adjust_pose_only) would otherwise spend more time in hipMalloc / hipFree than in
    // its kernels.  At most kPoolPerClass blocks per class are kept; everything is freed with the context.
    static constexpr int kPoolPerClass = 4;
    static constexpr size_t kPoolMaxBlock = 32u << 20;  // larger blocks (big batches) go straight to hipMalloc / hipFree
    std::map<size_t, std::vector<void*>> pool, host_pool;  // device blocks / pinned host blocks
    void* staging = nullptr;                // pinned host staging buffer of small uploads
    size_t staging_cap = 0;
    // Pinned host arena the big arrays of ONE batch's packing are carved from (kba_pack.hpp:PackArena): grow-only up to kPackArenaMax,
    // lent to a batch at its creation when no other live batch holds it, recycled when that batch is destroyed.  A batch packed into
    // it touches no fresh pages and uploads by DMA straight from where the pack wrote (1024 C2 windows: create 21 -> ~14 ms).
    void* pack_arena = nullptr;
    size_t pack_arena_cap = 0;
    size_t pack_arena_wanted = 0;           // what the last batch that did not fit would have needed
    bool pack_arena_busy = false;
    static constexpr size_t kPackArenaMax = size_t(1) << 30;
    void* staging_big = nullptr;            // pinned host buffer the results of a LARGE batch come back through (grow-only, <= kBigStageMax)
    size_t staging_big_cap = 0;
    static constexpr size_t kBigStageMax = 512u << 20;

    // Larger blocks (the arena of a 1024-window batch is ~1.5 GB) are kept too, in 64 MB classes, one per class and at most
    // kPoolLargeBytes altogether: hipMalloc / hipFree of blocks that size synchronise the DEVICE, so a caller that streams
    // batches through two contexts (pack + upload of batch k + 1 under the solve of batch k) would otherwise serialise on them.
    static constexpr size_t kLargeStep = 64u << 20;
    static constexpr size_t kPoolLargeBytes = size_t(16) << 30;
    size_t pooled_large = 0;
    static size_t size_class(size_t bytes) {
        if (bytes > kPoolMaxBlock) return (bytes + kLargeStep - 1) / kLargeStep * kLargeStep;
        size_t c = 256;
        while (c < bytes) c <<= 1;
        return c;
    }
    hipError_t pool_alloc(void** p, size_t bytes) {
        const size_t c = size_class(bytes);
        static const bool no_reuse = std::getenv("KBA_NO_POOL") != nullptr;  // debugging aid
        if (no_reuse) return hipMalloc(p, c);
        auto it = pool.find(c);
        if (it != pool.end() && !it->second.empty()) {
            *p = it->second.back();
            it->second.pop_back();
            if (bytes > kPoolMaxBlock) pooled_large -= c;
            return hipSuccess;
        }
        hipError_t e = hipMalloc(p, c);
        // Out of memory with idle blocks in the pool (a caller whose batches vary in size leaves one multi-GB block per 64 MB
        // class behind; several contexts may share the GPU): give the idle blocks back, largest first, and try again - the
        // pool may cost time, never an allocation that would have succeeded without it.
        while (e == hipErrorOutOfMemory && trim_largest()) {
            (void)hipGetLastError();
            e = hipMalloc(p, c);
        }
        return e;
    }
    // frees the largest idle device block of the pool; false when the pool holds none
    bool trim_largest() {
        for (auto it = pool.rbegin(); it != pool.rend(); ++it) {
            if (it->second.empty()) continue;
            (void)hipFree(it->second.back());
            it->second.pop_back();
            if (it->first > kPoolMaxBlock) pooled_large -= it->first;
            return true;
        }
        return false;
    }
    size_t pooled_large_cap() const {  // KBA_POOL_LARGE_MB: cap of the idle large blocks of a context (default 16 GB; 0 keeps none)
        static const size_t cap = std::getenv("KBA_POOL_LARGE_MB") ? (size_t)std::strtoull(std::getenv("KBA_POOL_LARGE_MB"), nullptr, 10) << 20 : kPoolLargeBytes;
        return cap;
    }
    void pool_free(void* p, size_t bytes) {
        const size_t c = size_class(bytes);
        auto& v = pool[c];
        static const bool no_reuse = std::getenv("KBA_NO_POOL") != nullptr;
        const bool large = bytes > kPoolMaxBlock;
        if (!no_reuse && (large ? (v.empty() && pooled_large + c <= pooled_large_cap()) : (int)v.size() < kPoolPerClass)) {
            v.push_back(p);
            if (large) pooled_large += c;
        } else {
            (void)hipFree(p);
        }
    }
    hipError_t host_alloc(void** p, size_t bytes) {
        const size_t c = size_class(bytes);
        auto it = host_pool.find(c);
        if (it != host_pool.end() && !it->second.empty()) {
            *p = it->second.back();
            it->second.pop_back();
            return hipSuccess;
        }
        return hipHostMalloc(p, c);
    }
    void host_free(void* p, size_t bytes) {
        auto& v = host_pool[size_class(bytes)];
        if ((int)v.size() < kPoolPerClass && bytes <= kPoolMaxBlock)
            v.push_back(p);
        else
            (void)hipHostFree(p);
    }
    void pool_release() {
        for (auto& kv : host_pool)
            for (void* p : kv.second) (void)hipHostFree(p);
        host_pool.clear();
        for (auto& kv : pool)
            for (void* p : kv.second) (void)hipFree(p);
        pool.clear();
        pooled_large = 0;
        if (xhost) (void)hipHostFree(xhost);
        xhost = nullptr;
        xhost_cap = 0;
        if (staging) (void)hipHostFree(staging);
        staging = nullptr;
        staging_cap = 0;
        if (staging_big) (void)hipHostFree(staging_big);
        staging_big = nullptr;
        staging_big_cap = 0;
        if (pack_arena) {
            kba::pack_arena_register(pack_arena, pack_arena_cap, false);
            (void)hipHostFree(pack_arena);
        }
        pack_arena = nullptr;
        pack_arena_cap = 0;
    }
};
