// TEST INFRASTRUCTURE - the product's C-ABI names (include/limo_hip.h) served by the CPU ORACLE (oracle/kba_oracle.cpp,
// oracle/exact_oracle.cpp, oracle/depth_oracle.cpp in liboracle.so), so that C++ code written against the C-ABI - the kba
// shim, limo_amd/kba/stream_driver.hpp, apps/limo_stream - can run a whole drive with the restated Ceres loop behind it.
//
// Why it exists: the emulated backend (tests/cpp/emu_pipeline.cpp) compiles the SAME lane functions as the gfx950 kernels for
// the host, so "GPU drive = emulated drive" only says the GPU runs its own source correctly.  This backend shares NOTHING
// with the kernels: dual-number functors (oracle/functors.hpp <- cost_functors_ceres.hpp:53-555), row-wise evaluation, Ceres
// 1.13's TrustRegionMinimizer / SchurEliminator / dense Cholesky as published (oracle/ceres_like.cpp), solveTrimmed
// (robust_solving.cpp:140-248).  A drive through it is the config-5 oracle (call order mono_lidar.cpp:186-260).
//
// Built only into tests/cpp/_build/libkba_oracle_abi.so by tests/emu_ffi.py.  NEVER linked into the product.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/limo_hip.h"

extern "C" {
void oracle_ba_default_options(limo_ba_options* o);
int oracle_ba_solve(limo_ba_window* w, const limo_ba_options* o, limo_ba_report* rep, int num_threads, int num_linear_solver_threads,
                    double* phase_times);
int oracle_ba_adjust_pose_only(limo_ba_window* w, const limo_speed_prior* prior, const limo_ba_options* o, limo_ba_report* rep,
                               int num_threads);
int oracle_landmark_init(int32_t n, const int32_t* ray_off, const limo_ray* rays, const uint8_t* use_depth, double* pos_out, uint8_t* ok);
void oracle_depth_default_params(limo_depth_params* out);
int oracle_depth_estimate(const float* cloud_xyzi, size_t n_pts, const double* T_cam_lidar, double f, double cx, double cy, int32_t img_w,
                          int32_t img_h, const float* feat_uv, size_t n_feat, const uint8_t* feat_is_ground, const limo_depth_params* params,
                          float* depth_out);
}

struct limo_ctx {
    std::string err;
    int threads = 3;  // opt.num_threads = 3, bundle_adjuster_keyframes.cpp:764 (the oracle's sums do not depend on it)
};

extern "C" {
int limo_abi_version(void) { return LIMO_ABI_VERSION; }
int limo_ctx_create(int, limo_ctx** out) {
    *out = new limo_ctx();
    if (const char* e = std::getenv("ORACLE_ABI_THREADS")) (*out)->threads = std::atoi(e) > 0 ? std::atoi(e) : 3;
    return LIMO_OK;
}
void limo_ctx_destroy(limo_ctx* c) { delete c; }
void* limo_host_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
void limo_host_free(void* p) { std::free(p); }
int limo_ctx_set_stream(limo_ctx*, void*) { return LIMO_OK; }
const char* limo_last_error(const limo_ctx* c) { return c ? c->err.c_str() : ""; }
void limo_ba_default_options(limo_ba_options* o) { oracle_ba_default_options(o); }

int limo_ba_solve(limo_ctx* c, limo_ba_window* w, const limo_ba_options* o, limo_ba_report* r) {
    return oracle_ba_solve(w, o, r, c ? c->threads : 3, 1, nullptr);
}
int limo_ba_adjust_pose_only(limo_ctx* c, limo_ba_window* w, const limo_speed_prior* p, const limo_ba_options* o, limo_ba_report* r) {
    return oracle_ba_adjust_pose_only(w, p, o, r, c ? c->threads : 3);
}
int limo_landmark_init(limo_ctx*, int32_t n, const int32_t* ray_off, const limo_ray* rays, const uint8_t* use_depth, double* pos_out,
                       uint8_t* ok) {
    if (n < 0 || (n > 0 && (!ray_off || !rays || !use_depth || !pos_out || !ok))) return LIMO_ERR_INVALID;
    return oracle_landmark_init(n, ray_off, rays, use_depth, pos_out, ok);
}
void limo_depth_default_params(limo_depth_params* out) { oracle_depth_default_params(out); }
int limo_depth_estimate(limo_ctx*, const float* cloud_xyzi, size_t n_pts, const double* T_cam_lidar, double f, double cx, double cy,
                        int32_t img_w, int32_t img_h, const float* feat_uv, size_t n_feat, const uint8_t* feat_is_ground,
                        const limo_depth_params* params, float* depth_out) {
    return oracle_depth_estimate(cloud_xyzi, n_pts, T_cam_lidar, f, cx, cy, img_w, img_h, feat_uv, n_feat, feat_is_ground, params, depth_out);
}
// the two-halves form of the call (limo_hip.h): these CPU backends do the work in _begin and hand it over in _end
static std::vector<float> g_depth_pending;
static bool g_depth_open = false;
int limo_depth_estimate_begin(limo_ctx*, const float* cloud_xyzi, size_t n_pts, const double* T_cam_lidar, double f, double cx, double cy,
                              int32_t img_w, int32_t img_h, const float* feat_uv, size_t n_feat, const uint8_t* feat_is_ground,
                              const limo_depth_params* params) {
    if (g_depth_open) return LIMO_ERR_INVALID;
    g_depth_pending.assign(n_feat, -1.f);
    const int rc = oracle_depth_estimate(cloud_xyzi, n_pts, T_cam_lidar, f, cx, cy, img_w, img_h, feat_uv, n_feat, feat_is_ground, params, g_depth_pending.data());
    g_depth_open = rc == LIMO_OK;
    return rc;
}
int limo_depth_estimate_end(limo_ctx*, float* depth_out, size_t n_feat) {
    if (!g_depth_open) return LIMO_ERR_INVALID;
    g_depth_open = false;
    if (n_feat != g_depth_pending.size() || (n_feat && !depth_out)) return LIMO_ERR_INVALID;
    std::copy(g_depth_pending.begin(), g_depth_pending.end(), depth_out);
    return LIMO_OK;
}
}
