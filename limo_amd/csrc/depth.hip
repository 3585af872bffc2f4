// depth.hip — LiDAR -> feature depth assignment on gfx950 (SURVEY §8a rows D1–D6, C-ABI limo_depth_estimate).
//
// Replaces the (un-vendored) mono_lidar_depth DepthEstimator as pinned by
// demo_keyframe_bundle_adjustment_meta/res/mono_lidar_fusion_parameters.yaml (cited as yaml:LINE); output contract
// FeaturePoint::d (matches_msg_types/include/matches_msg_types/feature_point.hpp:24-26).  Algorithmic choices the
// parameter file leaves open are the ones documented in oracle/depth_oracle.cpp (the test oracle of this path).
//
// Kernels
//   k_project        1 lane / lidar return   HBM   D1: lidar->camera, cut z<=0, pinhole projection, in-image test; the
//                                                  visible returns are binned into 8x8-pixel image cells (index lists)
//   k_features       1 wave / feature        -     D2: gather the returns of the cells under the 6x9 px rectangle (ballot
//                                                  compaction into LDS, ordered by return index), D3: depth histogram in
//                                                  LDS + nearest local maximum, D4: largest-triangle search as a wave
//                                                  reduction over point pairs, plane, ray intersection, D5: gates;
//                                                  D6b: ground features use the inverse-distance weighted patch
//   k_band_* / k_ransac_* / k_refine_*   D6a: RANSAC ground plane: order-preserving compaction of the z band, one lane
//                                        per hypothesis plane, inlier counts over (hypothesis group, return chunk)
//                                        workgroups with integer atomics, two-level deterministic refinement sums
// algorithmic bytes: 16 B read per return + 48 B written per visible return (u,v,x,y,z as fp64 + index);
// per feature 8 B + ~10 neighbours x 48 B + 4 B out.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/limo_hip.h"
#include "kba_math.hpp"
#include "limo_ctx.hpp"

namespace {

constexpr int kCell = 8;        // pixels per image cell
constexpr int kCellCap = 48;    // returns kept per cell (KITTI density: ~5 per cell)
constexpr int kMaxNb = 64;      // neighbours kept per feature
constexpr int kMaxBins = 512;   // histogram bins per feature (0.3 m bins => 150 m of depth range)
constexpr int kMaxHyp = 4096;   // RANSAC hypotheses

struct DepthView {
    const float* cloud;  // [n*4]
    int n_pts;
    double R[9], t[3];   // camera <- lidar
    double f, cx, cy;
    int img_w, img_h, cells_x, cells_y;
    double *pu, *pv, *px, *py, *pz;  // per return (valid where vis != 0)
    uint8_t* vis;
    int* cell_count;     // [cells]
    int* cell_pts;       // [cells*kCellCap]
    // ground plane
    int* band_idx;       // compacted indices of returns inside the z band, in index order
    double *bx, *by, *bz;  // camera-frame coordinates of the band returns (same order)
    int* band_blk;       // [blocks of 256 returns] in-band count per block, then its exclusive prefix
    double* ref_part;    // [chunks of 1024 band returns][6] partial sums of the refinement passes
    int* band_n;         // [1]
    int* hyp_count;      // [n_hyp]
    double* hyp_plane;   // [n_hyp*4]
    double* plane;       // [8]: n(3), d, ok, inliers, -, -
    double* red;         // reduction scratch
    // features
    const float* feat_uv;
    const uint8_t* feat_ground;
    int n_feat;
    float* out;
    limo_depth_params p;
};

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// ------------------------------------------------------------------------------------------ D1
__global__ __launch_bounds__(256) void k_project(DepthView d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n_pts) return;
    const float4 q = reinterpret_cast<const float4*>(d.cloud)[i];  // 16-byte coalesced read
    const double x = q.x, y = q.y, z = q.z;
    const double cxp = d.R[0] * x + d.R[1] * y + d.R[2] * z + d.t[0];
    const double cyp = d.R[3] * x + d.R[4] * y + d.R[5] * z + d.t[1];
    const double czp = d.R[6] * x + d.R[7] * y + d.R[8] * z + d.t[2];
    uint8_t vis = 0;
    if (!(d.p.do_use_cut_behind_camera && !(czp > 0.0)) && czp != 0.0) {
        const double u = d.f * cxp / czp + d.cx;
        const double v = d.f * cyp / czp + d.cy;
        if (u >= 0.0 && u < (double)d.img_w && v >= 0.0 && v < (double)d.img_h) {
            vis = 1;
            d.pu[i] = u;
            d.pv[i] = v;
            d.px[i] = cxp;
            d.py[i] = cyp;
            d.pz[i] = czp;
            const int cell = ((int)v / kCell) * d.cells_x + (int)u / kCell;
            const int pos = atomicAdd(&d.cell_count[cell], 1);
            if (pos < kCellCap) d.cell_pts[cell * kCellCap + pos] = i;
        }
    }
    d.vis[i] = vis;
}

// ------------------------------------------------------------------------------------------ D6a ground plane
// Order-preserving compaction of the returns with lidar z inside [min_z, max_z], three coalesced passes:
// per-block counts -> exclusive prefix over the blocks -> write (index + camera-frame coordinates).
__device__ __forceinline__ bool in_band(const DepthView& d, int i) {
    const double z = d.cloud[4 * (size_t)i + 2];
    return z >= d.p.ransac_plane_min_z && z <= d.p.ransac_plane_max_z;
}
__global__ __launch_bounds__(256) void k_band_count(DepthView d) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int flag = (i < d.n_pts) && in_band(d, i);
    const int n = __syncthreads_count(flag);
    if (threadIdx.x == 0) d.band_blk[blockIdx.x] = n;
}
__global__ __launch_bounds__(1024) void k_band_scan(DepthView d, int n_blk) {
    __shared__ int part[1024];
    const int chunk = (n_blk + 1023) / 1024;
    const int lo = threadIdx.x * chunk, hi = min(n_blk, lo + chunk);
    int c = 0;
    for (int b = lo; b < hi; ++b) c += d.band_blk[b];
    part[threadIdx.x] = c;
    __syncthreads();
    for (int st = 1; st < 1024; st <<= 1) {  // inclusive scan of the per-lane sums
        const int v = (int)threadIdx.x >= st ? part[threadIdx.x - st] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - c;
    for (int b = lo; b < hi; ++b) {
        const int n = d.band_blk[b];
        d.band_blk[b] = run;
        run += n;
    }
    if (threadIdx.x == 1023) *d.band_n = part[1023];
}
__global__ __launch_bounds__(256) void k_band_write(DepthView d) {
    __shared__ int wave_off[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool flag = (i < d.n_pts) && in_band(d, i);
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_off[wave] = __popcll(m);
    __syncthreads();
    int off = d.band_blk[blockIdx.x];
    for (int k = 0; k < wave; ++k) off += wave_off[k];
    if (flag) {
        const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
        const double x = d.cloud[4 * (size_t)i], y = d.cloud[4 * (size_t)i + 1], z = d.cloud[4 * (size_t)i + 2];
        d.band_idx[pos] = i;
        d.bx[pos] = d.R[0] * x + d.R[1] * y + d.R[2] * z + d.t[0];
        d.by[pos] = d.R[3] * x + d.R[4] * y + d.R[5] * z + d.t[1];
        d.bz[pos] = d.R[6] * x + d.R[7] * y + d.R[8] * z + d.t[2];
    }
}

__device__ __forceinline__ void cam_point(const DepthView& d, int i, double* p) {
    const double x = d.cloud[4 * (size_t)i], y = d.cloud[4 * (size_t)i + 1], z = d.cloud[4 * (size_t)i + 2];
    p[0] = d.R[0] * x + d.R[1] * y + d.R[2] * z + d.t[0];
    p[1] = d.R[3] * x + d.R[4] * y + d.R[5] * z + d.t[1];
    p[2] = d.R[6] * x + d.R[7] * y + d.R[8] * z + d.t[2];
}

// one lane per hypothesis: plane through three seeded band returns (hyp_count = -1 marks a degenerate draw)
__global__ __launch_bounds__(256) void k_ransac_planes(DepthView d, int n_hyp) {
    const int it = blockIdx.x * 256 + threadIdx.x;
    if (it >= n_hyp) return;
    const int nb = *d.band_n;
    double pl[4] = {0.0, 0.0, 0.0, 0.0};
    int ok = 0;
    if (nb >= 3) {
        const uint64_t h = splitmix64(d.p.ransac_seed * 0x100000001B3ull + (uint64_t)it);
        const size_t i0 = splitmix64(h) % nb, i1 = splitmix64(h + 1) % nb, i2 = splitmix64(h + 2) % nb;
        if (i0 != i1 && i0 != i2 && i1 != i2) {
            const double a[3] = {d.bx[i0], d.by[i0], d.bz[i0]}, b[3] = {d.bx[i1], d.by[i1], d.bz[i1]}, c[3] = {d.bx[i2], d.by[i2], d.bz[i2]};
            const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
            const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (nn > 1e-9) {
                pl[0] = n[0] / nn;
                pl[1] = n[1] / nn;
                pl[2] = n[2] / nn;
                pl[3] = -(pl[0] * a[0] + pl[1] * a[1] + pl[2] * a[2]);
                ok = 1;
            }
        }
    }
    d.hyp_count[it] = ok ? 0 : -1;
    for (int k = 0; k < 4; ++k) d.hyp_plane[4 * it + k] = pl[k];
}

// Inlier counts: workgroup (hypothesis group of kHypPerBlock, chunk of kRansacChunk band returns); every return is
// loaded once and tested against the group's planes (LDS); integer atomics make the totals order-independent.
constexpr int kHypPerBlock = 16;
constexpr int kRansacChunk = 2048;
__global__ __launch_bounds__(256) void k_ransac_count(DepthView d, int n_hyp) {
    __shared__ double pl[kHypPerBlock][4];
    __shared__ int valid[kHypPerBlock];
    const int h0 = blockIdx.x * kHypPerBlock;
    const int nb = *d.band_n;
    const int q0 = blockIdx.y * kRansacChunk;
    if (q0 >= nb) return;
    if (threadIdx.x < kHypPerBlock) {
        const int h = h0 + threadIdx.x;
        valid[threadIdx.x] = (h < n_hyp) && d.hyp_count[h] >= 0;  // counts only grow from 0, -1 stays -1
        for (int k = 0; k < 4; ++k) pl[threadIdx.x][k] = h < n_hyp ? d.hyp_plane[4 * h + k] : 0.0;
    }
    __syncthreads();
    int cnt[kHypPerBlock];  // wave-uniform inlier counts (ballot + popcount: no per-lane counters, no shuffles)
#pragma unroll
    for (int k = 0; k < kHypPerBlock; ++k) cnt[k] = 0;
    const double thr = d.p.ransac_plane_distance_treshold;
    const int q_end = min(nb, q0 + kRansacChunk);
    for (int qb = q0; qb < q_end; qb += 256) {
        const int q = qb + threadIdx.x;
        const bool live = q < q_end;
        const double x = live ? d.bx[q] : 0.0, y = live ? d.by[q] : 0.0, z = live ? d.bz[q] : 0.0;
#pragma unroll
        for (int k = 0; k < kHypPerBlock; ++k)
            cnt[k] += __popcll(__ballot(live && fabs(pl[k][0] * x + pl[k][1] * y + pl[k][2] * z + pl[k][3]) < thr));
    }
    __shared__ int bc[kHypPerBlock];
    if (threadIdx.x < kHypPerBlock) bc[threadIdx.x] = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < kHypPerBlock; ++k)
            if (cnt[k]) atomicAdd(&bc[k], cnt[k]);
    }
    __syncthreads();
    if (threadIdx.x < kHypPerBlock && bc[threadIdx.x] && valid[threadIdx.x]) atomicAdd(&d.hyp_count[h0 + threadIdx.x], bc[threadIdx.x]);
}

// sequential RANSAC semantics over the pre-computed hypotheses: keep the best so far, stop once the adaptive iteration
// bound k = log(1-p)/log(1-w^3) is reached
__global__ void k_ransac_pick(DepthView d, int n_hyp) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int nb = *d.band_n;
    int best = 0, bi = -1;
    double k_needed = n_hyp;
    for (int it = 0; it < n_hyp; ++it) {
        if (it >= k_needed) break;
        const int cnt = d.hyp_count[it];
        if (cnt > best) {
            best = cnt;
            bi = it;
            const double w = (double)cnt / (double)nb;
            const double denom = log(fmax(1e-300, 1.0 - w * w * w));
            k_needed = denom < 0 ? log(1.0 - d.p.ransac_plane_probability) / denom : 0.0;
        }
    }
    d.plane[4] = (best >= 3) ? 1.0 : 0.0;
    d.plane[5] = best;
    if (bi >= 0)
        for (int k = 0; k < 4; ++k) d.plane[k] = d.hyp_plane[4 * bi + k];
}

// least-squares refinement: centroid, then scatter matrix, over the band returns within refinement_treshold of the
// RANSAC plane.  pass 0: sums (1, x, y, z); pass 1: scatter (6 unique) around the centroid in d.red[0..3].
// Two levels in a FIXED order (deterministic): a workgroup reduces a chunk of 1024 returns (4 consecutive ones per lane,
// then a tree), one lane adds the chunk sums in chunk order.
__global__ __launch_bounds__(256) void k_refine_part(DepthView d, int pass) {
    if (d.plane[4] == 0.0) return;
    __shared__ double sh[256];
    const int nb = *d.band_n;
    const int q0 = blockIdx.x * 1024;
    if (q0 >= nb) return;
    const double pl[4] = {d.plane[0], d.plane[1], d.plane[2], d.plane[3]};
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const double c[3] = {pass ? d.red[1] / d.red[0] : 0.0, pass ? d.red[2] / d.red[0] : 0.0, pass ? d.red[3] / d.red[0] : 0.0};
    for (int k = 0; k < 4; ++k) {
        const int q = q0 + 4 * threadIdx.x + k;
        if (q >= nb) break;
        const double p[3] = {d.bx[q], d.by[q], d.bz[q]};
        if (!(fabs(pl[0] * p[0] + pl[1] * p[1] + pl[2] * p[2] + pl[3]) < d.p.ransac_plane_refinement_treshold)) continue;
        if (pass == 0) {
            acc[0] += 1.0;
            acc[1] += p[0];
            acc[2] += p[1];
            acc[3] += p[2];
        } else {
            const double e[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
            acc[0] += e[0] * e[0];
            acc[1] += e[0] * e[1];
            acc[2] += e[0] * e[2];
            acc[3] += e[1] * e[1];
            acc[4] += e[1] * e[2];
            acc[5] += e[2] * e[2];
        }
    }
    const int nval = pass ? 6 : 4;
    for (int k = 0; k < nval; ++k) {
        sh[threadIdx.x] = acc[k];
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) d.ref_part[(size_t)blockIdx.x * 6 + k] = sh[0];
        __syncthreads();
    }
}
__global__ __launch_bounds__(384) void k_refine_sum(DepthView d, int pass) {
    if (d.plane[4] == 0.0) return;
    const int nval = pass ? 6 : 4;
    const int v = threadIdx.x >> 6, lane = threadIdx.x & 63;  // wave v sums value v
    if (v >= nval) return;
    const int nb = *d.band_n;
    const int n_chunk = (nb + 1023) / 1024;
    double a = 0.0;
    for (int b = lane; b < n_chunk; b += 64) a += d.ref_part[(size_t)b * 6 + v];  // fixed partition ...
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);          // ... fixed tree
    if (lane == 0) d.red[(pass ? 4 : 0) + v] = a;
}

// One Jacobi rotation of the symmetric 3x3 matrix (a00 a01 a02 a11 a12 a22) in the (I,J) plane, eigenvectors in V;
// indices are compile-time constants so that everything stays in registers.
template <int I, int J>
__device__ __forceinline__ void jacobi_rot(double (&a)[3][3], double (&V)[3][3]) {
    if (a[I][J] == 0.0) return;
    const double tau = (a[J][J] - a[I][I]) / (2.0 * a[I][J]);
    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
    const double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double x = a[k][I], y = a[k][J];
        a[k][I] = cs * x - sn * y;
        a[k][J] = sn * x + cs * y;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double x = a[I][k], y = a[J][k];
        a[I][k] = cs * x - sn * y;
        a[J][k] = sn * x + cs * y;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double x = V[k][I], y = V[k][J];
        V[k][I] = cs * x - sn * y;
        V[k][J] = sn * x + cs * y;
    }
}

__device__ void smallest_eigvec(const double* C6, double* n) {  // C6 = xx xy xz yy yz zz
    double a[3][3] = {{C6[0], C6[1], C6[2]}, {C6[1], C6[3], C6[4]}, {C6[2], C6[4], C6[5]}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 50; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-40 * diag || off < 1e-300) break;  // converged to far below the rounding of the entries
        jacobi_rot<0, 1>(a, V);
        jacobi_rot<0, 2>(a, V);
        jacobi_rot<1, 2>(a, V);
    }
    const double e0 = a[0][0], e1 = a[1][1], e2 = a[2][2];
    const int m = (e1 < e0) ? ((e2 < e1) ? 2 : 1) : ((e2 < e0) ? 2 : 0);
    const double v0 = m == 0 ? V[0][0] : m == 1 ? V[0][1] : V[0][2];
    const double v1 = m == 0 ? V[1][0] : m == 1 ? V[1][1] : V[1][2];
    const double v2 = m == 0 ? V[2][0] : m == 1 ? V[2][1] : V[2][2];
    const double nn = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
    n[0] = v0 / nn;
    n[1] = v1 / nn;
    n[2] = v2 / nn;
}

__global__ void k_refine_finish(DepthView d) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (d.plane[4] == 0.0) return;
    double n[3] = {d.plane[0], d.plane[1], d.plane[2]}, dd = d.plane[3];
    if (d.p.ransac_plane_use_refinement && d.red[0] >= 3.0) {
        const double c[3] = {d.red[1] / d.red[0], d.red[2] / d.red[0], d.red[3] / d.red[0]};
        smallest_eigvec(d.red + 4, n);
        dd = -(n[0] * c[0] + n[1] * c[1] + n[2] * c[2]);
    }
    if (dd < 0) {
        n[0] = -n[0];
        n[1] = -n[1];
        n[2] = -n[2];
        dd = -dd;
    }
    d.plane[0] = n[0];
    d.plane[1] = n[1];
    d.plane[2] = n[2];
    d.plane[3] = dd;
}

// ------------------------------------------------------------------------------------------ D2–D5, D6b
__device__ __forceinline__ bool ray_plane_depth(const double* n, double dd, double u, double v, const DepthView& d,
                                                double* depth) {
    const double r[3] = {(u - d.cx) / d.f, (v - d.cy) / d.f, 1.0};
    const double rn = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    const double nr = n[0] * r[0] + n[1] * r[1] + n[2] * r[2];
    if (fabs(nr / rn) < d.p.viewray_plane_orthoganality_treshold) return false;
    *depth = -dd / nr;
    return true;
}

__device__ __forceinline__ double sin_at(const double* o, const double* a, const double* b) {
    const double e1[3] = {a[0] - o[0], a[1] - o[1], a[2] - o[2]}, e2[3] = {b[0] - o[0], b[1] - o[1], b[2] - o[2]};
    const double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double n1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), n2 = sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
    if (!(n1 > 0.0) || !(n2 > 0.0)) return 0.0;
    return sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) / (n1 * n2);
}

struct WaveLds {
    int nb_idx[kMaxNb];     // neighbour return indices (sorted)
    int tmp_idx[kMaxNb];
    double seg[kMaxNb][3];  // points of the selected histogram bin / ground patch
    int bins[kMaxBins];
    int n_nb, n_seg;
};

__global__ __launch_bounds__(256) void k_features(DepthView d) {
    __shared__ WaveLds lds[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + wave;
    if (k >= d.n_feat) return;  // whole wave exits together
    WaveLds& L = lds[wave];
    const double fu = d.feat_uv[2 * (size_t)k], fv = d.feat_uv[2 * (size_t)k + 1];
    const double hw = 0.5 * d.p.pixelarea_search_width, hh = 0.5 * d.p.pixelarea_search_height;
    const double cu = fu + d.p.pixelarea_search_offset_x, cv = fv + d.p.pixelarea_search_offset_y;
    // ---- D2: candidates from the cells under the rectangle, ballot compaction
    const int cx0 = max(0, (int)floor((cu - hw) / kCell)), cx1 = min(d.cells_x - 1, (int)floor((cu + hw) / kCell));
    const int cy0 = max(0, (int)floor((cv - hh) / kCell)), cy1 = min(d.cells_y - 1, (int)floor((cv + hh) / kCell));
    int n = 0;
    for (int cy = cy0; cy <= cy1; ++cy)
        for (int cx = cx0; cx <= cx1; ++cx) {
            const int cell = cy * d.cells_x + cx;
            const int cnt = min(kCellCap, d.cell_count[cell]);
            bool in = false;
            int idx = -1;
            if (lane < cnt) {
                idx = d.cell_pts[cell * kCellCap + lane];
                in = fabs(d.pu[idx] - cu) <= hw && fabs(d.pv[idx] - cv) <= hh;
            }
            const unsigned long long m = __ballot(in);
            if (in) {
                const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
                if (pos < kMaxNb) L.tmp_idx[pos] = idx;
            }
            n += __popcll(m);
        }
    n = min(n, kMaxNb);
    // order by return index (rank sort inside the wave) so every later step sees the lidar order the oracle sees
    __builtin_amdgcn_wave_barrier();
    if (lane < n) {
        const int mine = L.tmp_idx[lane];
        int rank = 0;
        for (int q = 0; q < n; ++q) rank += L.tmp_idx[q] < mine;
        L.nb_idx[rank] = mine;
    }
    __builtin_amdgcn_wave_barrier();
    float result = -1.0f;
    if (n >= d.p.neighbors_count_min) {
        const bool ground_feat = d.feat_ground && d.feat_ground[k] && d.plane[4] != 0.0;
        double depth = -1.0, zlo = 0.0, zhi = 0.0;
        bool have = false;
        if (ground_feat) {
            // ---- D6b: inverse-distance weighted patch over the neighbours close to the sweep's ground plane (lane 0,
            //      sequential in return order: identical summation order to the oracle)
            if (lane == 0) {
                const double gn[3] = {d.plane[0], d.plane[1], d.plane[2]}, gd = d.plane[3];
                double sw = 0, c[3] = {0, 0, 0};
                int m = 0;
                zlo = 1.79769313486231570e308;
                zhi = -zlo;
                for (int q = 0; q < n; ++q) {
                    const int i = L.nb_idx[q];
                    const double p[3] = {d.px[i], d.py[i], d.pz[i]};
                    const double dist = gn[0] * p[0] + gn[1] * p[1] + gn[2] * p[2] + gd;
                    if (fabs(dist) < d.p.ransac_plane_point_distance_treshold) {
                        const double w = d.p.plane_estimator_use_mestimator ? 1.0 / (fabs(dist) + 0.01) : 1.0;
                        L.seg[m][0] = p[0];
                        L.seg[m][1] = p[1];
                        L.seg[m][2] = p[2];
                        reinterpret_cast<double*>(L.bins)[m] = w;  // weights parked in the (unused) histogram storage
                        sw += w;
                        for (int a = 0; a < 3; ++a) c[a] += w * p[a];
                        zlo = fmin(zlo, p[2]);
                        zhi = fmax(zhi, p[2]);
                        ++m;
                    }
                }
                double pn[3] = {gn[0], gn[1], gn[2]}, pd = gd;
                bool local = false;
                if (m >= 3) {
                    for (int a = 0; a < 3; ++a) c[a] /= sw;
                    double C6[6] = {0, 0, 0, 0, 0, 0};
                    for (int q = 0; q < m; ++q) {
                        const double w = reinterpret_cast<double*>(L.bins)[q];
                        const double e[3] = {L.seg[q][0] - c[0], L.seg[q][1] - c[1], L.seg[q][2] - c[2]};
                        C6[0] += w * e[0] * e[0];
                        C6[1] += w * e[0] * e[1];
                        C6[2] += w * e[0] * e[2];
                        C6[3] += w * e[1] * e[1];
                        C6[4] += w * e[1] * e[2];
                        C6[5] += w * e[2] * e[2];
                    }
                    double ln[3];
                    smallest_eigvec(C6, ln);
                    if (fabs(ln[0] * gn[0] + ln[1] * gn[1] + ln[2] * gn[2]) >= 0.9) {
                        local = true;
                        pn[0] = ln[0];
                        pn[1] = ln[1];
                        pn[2] = ln[2];
                        pd = -(ln[0] * c[0] + ln[1] * c[1] + ln[2] * c[2]);
                    }
                }
                if (!local) {
                    zlo = 0.0;
                    zhi = 1.79769313486231570e308;
                }
                have = ray_plane_depth(pn, pd, fu, fv, d, &depth);
            }
        } else {
            // ---- D3: depth histogram, nearest local maximum
            double zmin = 1.79769313486231570e308, zmax = -1.79769313486231570e308;
            double myz = 0.0;
            if (lane < n) {
                myz = d.pz[L.nb_idx[lane]];
                zmin = zmax = myz;
            }
            for (int off = 32; off > 0; off >>= 1) {
                zmin = fmin(zmin, __shfl_xor(zmin, off, 64));
                zmax = fmax(zmax, __shfl_xor(zmax, off, 64));
            }
            int nseg = 0;
            bool seg_ok = true;
            if (d.p.do_use_histogram_segmentation) {
                const double bw = d.p.histogram_segmentation_bin_width;
                const int nbins = (int)floor((zmax - zmin) / bw) + 1;
                if (nbins > kMaxBins) {
                    seg_ok = false;  // depth span beyond 150 m inside one 6x9 px window: treat as unsegmentable
                } else {
                    for (int b = lane; b < nbins; b += 64) L.bins[b] = 0;
                    __builtin_amdgcn_wave_barrier();
                    int mybin = -1;
                    if (lane < n) {
                        mybin = min(nbins - 1, (int)floor((myz - zmin) / bw));
                        atomicAdd(&L.bins[mybin], 1);
                    }
                    __builtin_amdgcn_wave_barrier();
                    int pick = 0x7fffffff;
                    for (int b = lane; b < nbins; b += 64) {
                        const int cnt = L.bins[b], prev = b > 0 ? L.bins[b - 1] : 0, next = b + 1 < nbins ? L.bins[b + 1] : 0;
                        if (cnt >= d.p.histogram_segmentation_min_pointcount && cnt > prev && cnt >= next) pick = min(pick, b);
                    }
                    for (int off = 32; off > 0; off >>= 1) pick = min(pick, __shfl_xor(pick, off, 64));
                    if (pick == 0x7fffffff) {
                        seg_ok = false;
                    } else {
                        const bool mine = lane < n && mybin == pick;
                        const unsigned long long m = __ballot(mine);
                        if (mine) {
                            const int pos = __popcll(m & ((1ull << lane) - 1ull));
                            const int i = L.nb_idx[lane];
                            L.seg[pos][0] = d.px[i];
                            L.seg[pos][1] = d.py[i];
                            L.seg[pos][2] = d.pz[i];
                        }
                        nseg = __popcll(m);
                    }
                }
            } else {
                if (lane < n) {
                    const int i = L.nb_idx[lane];
                    L.seg[lane][0] = d.px[i];
                    L.seg[lane][1] = d.py[i];
                    L.seg[lane][2] = d.pz[i];
                }
                nseg = n;
            }
            __builtin_amdgcn_wave_barrier();
            if (seg_ok && nseg >= 3) {
                // ---- D4: largest triangle: every lane takes point pairs (i<j) and scans k>j; wave arg-max with the
                //      lexicographically first (i,j,k) among equal areas
                double best = -1.0;
                int bi = 0x7fff, bj = 0x7fff, bk = 0x7fff;
                for (int pq = lane; pq < nseg * nseg; pq += 64) {
                    const int i = pq / nseg, j = pq % nseg;
                    if (j <= i) continue;
                    const double e1[3] = {L.seg[j][0] - L.seg[i][0], L.seg[j][1] - L.seg[i][1], L.seg[j][2] - L.seg[i][2]};
                    for (int l = j + 1; l < nseg; ++l) {
                        const double e2[3] = {L.seg[l][0] - L.seg[i][0], L.seg[l][1] - L.seg[i][1], L.seg[l][2] - L.seg[i][2]};
                        const double c0 = e1[1] * e2[2] - e1[2] * e2[1], c1 = e1[2] * e2[0] - e1[0] * e2[2], c2 = e1[0] * e2[1] - e1[1] * e2[0];
                        const double a2 = c0 * c0 + c1 * c1 + c2 * c2;
                        if (a2 > best) {  // pairs are visited in increasing (i,j), l increasing: first maximum wins
                            best = a2;
                            bi = i;
                            bj = j;
                            bk = l;
                        }
                    }
                }
                for (int off = 32; off > 0; off >>= 1) {
                    const double ob = __shfl_xor(best, off, 64);
                    const int oi = __shfl_xor(bi, off, 64), oj = __shfl_xor(bj, off, 64), ok2 = __shfl_xor(bk, off, 64);
                    const bool better = ob > best || (ob == best && (oi < bi || (oi == bi && (oj < bj || (oj == bj && ok2 < bk)))));
                    if (better) {
                        best = ob;
                        bi = oi;
                        bj = oj;
                        bk = ok2;
                    }
                }
                if (lane == 0 && best >= 0.0) {
                    const double* A = L.seg[bi];
                    const double* B = L.seg[bj];
                    const double* Cc = L.seg[bk];
                    bool ok = true;
                    if (d.p.do_check_triangleplanar_condition) {
                        const double s = fmin(sin_at(A, B, Cc), fmin(sin_at(B, A, Cc), sin_at(Cc, A, B)));
                        if (s < d.p.triangleplanar_crossnorm_treshold) ok = false;
                    }
                    if (ok) {
                        const double e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {Cc[0] - A[0], Cc[1] - A[1], Cc[2] - A[2]};
                        double pn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                        const double nn = sqrt(pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2]);
                        if (nn > 0.0) {
                            pn[0] /= nn;
                            pn[1] /= nn;
                            pn[2] /= nn;
                            const double pd = -(pn[0] * A[0] + pn[1] * A[1] + pn[2] * A[2]);
                            have = ray_plane_depth(pn, pd, fu, fv, d, &depth);
                            zlo = 1.79769313486231570e308;
                            zhi = -zlo;
                            for (int q = 0; q < nseg; ++q) {
                                zlo = fmin(zlo, L.seg[q][2]);
                                zhi = fmax(zhi, L.seg[q][2]);
                            }
                        }
                    }
                }
            }
        }
        // ---- D5: gates
        if (lane == 0 && have) {
            bool ok = true;
            if (d.p.treshold_depth_enabled && !(depth > d.p.treshold_depth_min && depth < d.p.treshold_depth_max)) ok = false;
            if (ok && d.p.treshold_depth_local_enabled) {
                const double v = d.p.treshold_depth_local_value;
                const double lo = d.p.treshold_depth_local_valuetype ? zlo * (1.0 - v) : zlo - v;
                const double hi = d.p.treshold_depth_local_valuetype ? zhi * (1.0 + v) : zhi + v;
                if (!(depth >= lo && depth <= hi)) ok = false;
            }
            if (ok) result = (float)depth;
        }
    }
    if (lane == 0) d.out[k] = result;
}

// ------------------------------------------------------------------------------------------ workspace
struct DepthWs {
    size_t cap_pts = 0, cap_feat = 0, cap_cells = 0;
    float* cloud = nullptr;
    double *pu = nullptr, *pv = nullptr, *px = nullptr, *py = nullptr, *pz = nullptr;
    uint8_t *vis = nullptr, *feat_ground = nullptr;
    int *cell_count = nullptr, *cell_pts = nullptr, *band_idx = nullptr, *band_n = nullptr, *hyp_count = nullptr, *band_blk = nullptr;
    double *hyp_plane = nullptr, *plane = nullptr, *red = nullptr, *bx = nullptr, *by = nullptr, *bz = nullptr, *ref_part = nullptr;
    float *feat_uv = nullptr, *out = nullptr;
    void release() {
        void* ptrs[] = {cloud, pu, pv, px, py, pz, vis, feat_ground, cell_count, cell_pts, band_idx, band_n, hyp_count,
                        hyp_plane, plane, red, feat_uv, out, band_blk, bx, by, bz, ref_part};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        *this = DepthWs();
    }
};

void depth_ws_free(void* p) {
    DepthWs* w = static_cast<DepthWs*>(p);
    w->release();
    delete w;
}

#define HIP_TRY(ctx, expr)                                                   \
    do {                                                                     \
        hipError_t e__ = (expr);                                             \
        if (e__ != hipSuccess) {                                             \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__); \
            return LIMO_ERR_RUNTIME;                                         \
        }                                                                    \
    } while (0)

template <typename T>
int grow(limo_ctx* ctx, T** p, size_t n) {
    if (*p) HIP_TRY(ctx, hipFree(*p));
    *p = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)p, sizeof(T) * std::max<size_t>(n, 1)));
    return LIMO_OK;
}

}  // namespace

extern "C" {

void limo_depth_default_params(limo_depth_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->pixelarea_search_width = 6;
    p->pixelarea_search_height = 9;
    p->pixelarea_search_offset_x = 0;
    p->pixelarea_search_offset_y = 0;
    p->neighbors_count_min = 3;
    p->do_use_histogram_segmentation = 1;
    p->histogram_segmentation_bin_width = 0.3;
    p->histogram_segmentation_min_pointcount = 1;
    p->treshold_depth_enabled = 1;
    p->treshold_depth_max = 100.0;
    p->treshold_depth_min = 0.0;
    p->treshold_depth_local_enabled = 1;
    p->treshold_depth_local_valuetype = 1;
    p->treshold_depth_local_value = 0.5;
    p->do_use_cut_behind_camera = 1;
    p->do_use_triangle_size_maximation = 1;
    p->do_check_triangleplanar_condition = 1;
    p->triangleplanar_crossnorm_treshold = 0.1;
    p->viewray_plane_orthoganality_treshold = 0.1;
    p->do_use_ransac_plane = 1;
    p->ransac_plane_distance_treshold = 0.2;
    p->ransac_plane_min_z = -3.5;
    p->ransac_plane_max_z = -1.0;
    p->ransac_plane_max_iterations = 600;
    p->ransac_plane_probability = 0.99;
    p->ransac_plane_use_refinement = 1;
    p->ransac_plane_refinement_treshold = 10.2;
    p->ransac_plane_point_distance_treshold = 0.2;
    p->plane_estimator_use_mestimator = 1;
    p->ransac_seed = 1;
}

int limo_depth_estimate(limo_ctx* ctx, const float* cloud_xyzi, size_t n_pts, const double* T_cam_lidar, double f,
                        double cx, double cy, int32_t img_w, int32_t img_h, const float* feat_uv, size_t n_feat,
                        const uint8_t* feat_is_ground, const limo_depth_params* params, float* depth_out) {
    if (!ctx || (n_pts && !cloud_xyzi) || !T_cam_lidar || (n_feat && (!feat_uv || !depth_out)) || img_w <= 0 || img_h <= 0)
        return LIMO_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return LIMO_ERR_NO_DEVICE;
    limo_depth_params p;
    if (params)
        p = *params;
    else
        limo_depth_default_params(&p);
    if (p.ransac_plane_max_iterations > kMaxHyp) p.ransac_plane_max_iterations = kMaxHyp;
    if (!ctx->depth_ws) {
        ctx->depth_ws = new DepthWs();
        ctx->depth_ws_free = depth_ws_free;
    }
    DepthWs& W = *static_cast<DepthWs*>(ctx->depth_ws);
    hipStream_t s = ctx->stream;
    const int cells_x = (img_w + kCell - 1) / kCell, cells_y = (img_h + kCell - 1) / kCell;
    const size_t cells = (size_t)cells_x * cells_y;
    if (n_pts > W.cap_pts) {
        int rc = LIMO_OK;
        rc |= grow(ctx, &W.cloud, n_pts * 4);
        rc |= grow(ctx, &W.pu, n_pts);
        rc |= grow(ctx, &W.pv, n_pts);
        rc |= grow(ctx, &W.px, n_pts);
        rc |= grow(ctx, &W.py, n_pts);
        rc |= grow(ctx, &W.pz, n_pts);
        rc |= grow(ctx, &W.vis, n_pts);
        rc |= grow(ctx, &W.band_idx, n_pts);
        rc |= grow(ctx, &W.bx, n_pts);
        rc |= grow(ctx, &W.by, n_pts);
        rc |= grow(ctx, &W.bz, n_pts);
        rc |= grow(ctx, &W.band_blk, (n_pts + 255) / 256 + 1);
        rc |= grow(ctx, &W.ref_part, ((n_pts + 1023) / 1024 + 1) * 6);
        if (rc != LIMO_OK) return LIMO_ERR_RUNTIME;
        W.cap_pts = n_pts;
    }
    if (n_feat > W.cap_feat) {
        int rc = LIMO_OK;
        rc |= grow(ctx, &W.feat_uv, n_feat * 2);
        rc |= grow(ctx, &W.feat_ground, n_feat);
        rc |= grow(ctx, &W.out, n_feat);
        if (rc != LIMO_OK) return LIMO_ERR_RUNTIME;
        W.cap_feat = n_feat;
    }
    if (cells > W.cap_cells) {
        int rc = LIMO_OK;
        rc |= grow(ctx, &W.cell_count, cells);
        rc |= grow(ctx, &W.cell_pts, cells * kCellCap);
        if (rc != LIMO_OK) return LIMO_ERR_RUNTIME;
        W.cap_cells = cells;
    }
    if (!W.band_n) {
        int rc = LIMO_OK;
        rc |= grow(ctx, &W.band_n, 4);
        rc |= grow(ctx, &W.hyp_count, kMaxHyp);
        rc |= grow(ctx, &W.hyp_plane, (size_t)kMaxHyp * 4);
        rc |= grow(ctx, &W.plane, 8);
        rc |= grow(ctx, &W.red, 16);
        if (rc != LIMO_OK) return LIMO_ERR_RUNTIME;
    }
    DepthView d;
    std::memset(&d, 0, sizeof(d));
    d.cloud = W.cloud;
    d.n_pts = (int)n_pts;
    kba::quat_R(T_cam_lidar, d.R);
    for (int i = 0; i < 3; ++i) d.t[i] = T_cam_lidar[4 + i];
    d.f = f;
    d.cx = cx;
    d.cy = cy;
    d.img_w = img_w;
    d.img_h = img_h;
    d.cells_x = cells_x;
    d.cells_y = cells_y;
    d.pu = W.pu;
    d.pv = W.pv;
    d.px = W.px;
    d.py = W.py;
    d.pz = W.pz;
    d.vis = W.vis;
    d.cell_count = W.cell_count;
    d.cell_pts = W.cell_pts;
    d.band_idx = W.band_idx;
    d.band_n = W.band_n;
    d.bx = W.bx;
    d.by = W.by;
    d.bz = W.bz;
    d.band_blk = W.band_blk;
    d.ref_part = W.ref_part;
    d.hyp_count = W.hyp_count;
    d.hyp_plane = W.hyp_plane;
    d.plane = W.plane;
    d.red = W.red;
    d.feat_uv = W.feat_uv;
    d.n_feat = (int)n_feat;
    d.out = W.out;
    d.p = p;
    bool any_ground = false;
    if (feat_is_ground)
        for (size_t k = 0; k < n_feat; ++k) any_ground = any_ground || feat_is_ground[k];
    d.feat_ground = any_ground ? W.feat_ground : nullptr;

    if (n_pts) HIP_TRY(ctx, hipMemcpyAsync(W.cloud, cloud_xyzi, sizeof(float) * 4 * n_pts, hipMemcpyHostToDevice, s));
    if (n_feat) HIP_TRY(ctx, hipMemcpyAsync(W.feat_uv, feat_uv, sizeof(float) * 2 * n_feat, hipMemcpyHostToDevice, s));
    if (any_ground) HIP_TRY(ctx, hipMemcpyAsync(W.feat_ground, feat_is_ground, n_feat, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(W.cell_count, 0, sizeof(int) * cells, s));
    HIP_TRY(ctx, hipMemsetAsync(W.plane, 0, sizeof(double) * 8, s));
    if (n_pts) hipLaunchKernelGGL(k_project, dim3((unsigned)((n_pts + 255) / 256)), dim3(256), 0, s, d);
    if (any_ground && p.do_use_ransac_plane && n_pts) {
        const int n_hyp = std::max(1, p.ransac_plane_max_iterations);
        const int n_blk = (int)((n_pts + 255) / 256), n_chunk_r = (int)((n_pts + kRansacChunk - 1) / kRansacChunk),
                  n_chunk_f = (int)((n_pts + 1023) / 1024);  // upper bounds: the band size is only known on the device
        hipLaunchKernelGGL(k_band_count, dim3(n_blk), dim3(256), 0, s, d);
        hipLaunchKernelGGL(k_band_scan, dim3(1), dim3(1024), 0, s, d, n_blk);
        hipLaunchKernelGGL(k_band_write, dim3(n_blk), dim3(256), 0, s, d);
        hipLaunchKernelGGL(k_ransac_planes, dim3((n_hyp + 255) / 256), dim3(256), 0, s, d, n_hyp);
        hipLaunchKernelGGL(k_ransac_count, dim3((n_hyp + kHypPerBlock - 1) / kHypPerBlock, n_chunk_r), dim3(256), 0, s, d, n_hyp);
        hipLaunchKernelGGL(k_ransac_pick, dim3(1), dim3(64), 0, s, d, n_hyp);
        if (p.ransac_plane_use_refinement) {
            for (int pass = 0; pass < 2; ++pass) {
                hipLaunchKernelGGL(k_refine_part, dim3(n_chunk_f), dim3(256), 0, s, d, pass);
                hipLaunchKernelGGL(k_refine_sum, dim3(1), dim3(384), 0, s, d, pass);
            }
        }
        hipLaunchKernelGGL(k_refine_finish, dim3(1), dim3(64), 0, s, d);
    }
    if (n_feat) {
        hipLaunchKernelGGL(k_features, dim3((unsigned)((n_feat + 3) / 4)), dim3(256), 0, s, d);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(depth_out, W.out, sizeof(float) * n_feat, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return LIMO_OK;
}

}  // extern "C"
