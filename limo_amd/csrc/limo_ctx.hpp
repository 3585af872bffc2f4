// limo_ctx.hpp — the context object behind `limo_ctx*` (one per thread / GPU / stream), shared by the translation
// units that implement the C-ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

struct limo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own = nullptr;
    std::string err;
    void* depth_ws = nullptr;               // workspace of limo_depth_estimate (depth.hip), grown on demand
    void (*depth_ws_free)(void*) = nullptr;
    void* comm = nullptr;                   // ncclComm_t of a landmark-sharded solve (limo_ctx_comm_init), else null
    int comm_rank = 0, comm_world = 1;
};
