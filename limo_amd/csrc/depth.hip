// depth.hip — LiDAR -> feature depth assignment on gfx950 (SURVEY §8a rows D1–D6; C-ABI limo_depth_estimate and
// limo_depth_estimate_batch).
//
// Replaces the (un-vendored) mono_lidar_depth DepthEstimator as pinned by
// demo_keyframe_bundle_adjustment_meta/res/mono_lidar_fusion_parameters.yaml (cited as yaml:LINE); output contract
// FeaturePoint::d (matches_msg_types/include/matches_msg_types/feature_point.hpp:24-26).  Algorithmic choices the
// parameter file leaves open are the ones documented in oracle/depth_oracle.cpp (the test oracle of this path).
//
// BIT-EXACT CONTRACT.  Every accept / reject decision and every depth of this file equals the oracle's bit for bit:
//   * this translation unit is compiled with floating-point contraction OFF (pragma below + -ffp-contract=off in the
//     build): every + - * / sqrt is one IEEE-754 operation, in the order the oracle writes it;
//   * nothing derived from a return is stored in fp64: every kernel recomputes the camera-frame position / pixel of a
//     return from its 16-byte record with the same statements (cam_point, pixel_of), so all kernels see the same bits;
//   * the only long sum of the path - the moments of the ground-plane refinement - is accumulated in FIXED POINT
//     (int64, units 2^-30 m and 2^-20 m^2): integer addition is associative, so the sum does not depend on the
//     workgroup / lane / atomic order, and the oracle forms the same integers;
//   * workgroups never hand data to each other inside a launch, with one exception: the look-back scan of k_project,
//     whose messages are single 64-bit words (flag and value in one atomic).  Everything else is ordered by kernel
//     boundaries (no "last workgroup" hand-offs through relaxed counters).
//
// Seven launches per call, every one with the frame as a grid dimension (a call carries 1..kMaxBatch sweeps):
//   k_project    1 lane / lidar return   HBM   D1: lidar->camera, cut z<=0, pinhole projection, in-image test; the index
//                                              of a visible return goes into the list of its 8x8-pixel image cell.  D6a,
//                                              same pass: order-preserving compaction of the indices of the returns
//                                              inside the z band (single-pass decoupled look-back scan over the
//                                              workgroups of a frame).  Also clears the OTHER scratch zone (cell
//                                              counters, inlier counters, moments, scan state) for the next call.
//   k_ransac<first>, k_pick, k_ransac<rest>   workgroup = (64 hypotheses, 1024 returns): planes from seeded draws out
//                of the band list, inlier counts over the sweep (band predicate evaluated on the fly) as integer
//                atomics.  k_pick (one wave per frame) applies the sequential best-so-far / adaptive-iteration-bound
//                semantics to hypotheses 0..63, which is where that loop almost always stops; the workgroups of <rest>
//                leave at once unless it did not.
//   k_refine     fixed-point moments of the band returns near the RANSAC plane (int64 atomics, order-free)
//   k_plane      one wave per frame: centroid + scatter from the moments, smallest eigenvector, orientation
//   k_features   1 wave / feature   D2: the returns of the cells under the 6x9 px rectangle, all cells in one pass
//                (prefix sum over the cell counts, lane = candidate), ordered by return index; D3: depth histogram in
//                LDS + nearest local maximum, D4: largest triangle as a wave reduction over point pairs, plane, ray
//                intersection, D5: gates; D6b: ground features use the inverse-distance weighted patch.
// HBM bytes moved (16 B / return + 20 B / visible return; 212 B / feature): k_project reads 16 B / return
// and writes 4 B / visible return + 4 B / band return; k_ransac / k_refine read 16 B / return; k_features reads
// ~20 B / candidate return (index + record) + 8 B / cell.
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/limo_hip.h"
#include "limo_ctx.hpp"

namespace {

constexpr int kCell = 8;        // pixels per image cell
constexpr int kCellCap = 48;    // returns kept per cell (a spinning 64-beam scanner: 5-13 per cell); more => LIMO_ERR_INVALID
constexpr int kMaxNb = 64;      // neighbours per feature; more => LIMO_ERR_INVALID
constexpr int kMaxBins = 512;   // histogram bins per feature (0.3 m bins => 150 m of depth range)
constexpr int kMaxHyp = 4096;   // RANSAC hypotheses
constexpr int kMaxBatch = 32;   // frames per launch group (per-frame sizes and pointers travel as kernel arguments)
constexpr int kHypPerBlock = 64;   // = lanes of a wave: lane k keeps the count of plane k
constexpr int kChunk = 1024;       // returns per workgroup of k_ransac / k_refine, 4 per lane
constexpr int kMomVals = 10;       // count, sum e (3), sum e e^T (6)
enum { CTR_TICKET = 0, CTR_COUNT = 4 };
enum { OVF_CELL = 1, OVF_NEIGHBOURS = 2 };
// fixed-point units of the refinement moments (oracle/depth_oracle.cpp uses the same constants)
constexpr double kMomScale1 = 1073741824.0;  // 2^30 per metre
constexpr double kMomScale2 = 1048576.0;     // 2^20 per square metre
constexpr double kMomRange = 1024.0;         // returns further than this from the anchor point do not take part

struct DepthView {
    double R[9], t[3];  // camera <- lidar
    double f, cx, cy;
    int img_w, img_h, cells_x, cells_y, n_cells;
    int n_frames, n_hyp;
    uint32_t ground_mask;  // bit f: frame f has ground-labelled features and the RANSAC plane is wanted
    limo_depth_params p;
    // per frame (device pointers: the caller's, or this context's upload buffers)
    const float* cloud[kMaxBatch];
    const float* feat_uv[kMaxBatch];
    const uint8_t* feat_ground[kMaxBatch];  // null: no ground labels
    float* out[kMaxBatch];
    int n_pts[kMaxBatch], n_feat[kMaxBatch];
    // workspace; frame f lives at base + f * stride
    size_t pt_stride;                 // returns
    int* cell_pts;                    // [frame][cell][kCellCap] indices of the visible returns of a cell
    int* band_idx;                    // [frame][pt_stride] indices of the returns inside the z band, in index order
    int* band_n;                      // [frame]
    double* plane;                    // [frame][8]: n(3), d, ok, RANSAC inliers, more hypotheses wanted, iteration bound
    int* pick;                        // [frame][2]: best hypothesis, its count (after k_pick)
    int* overflow;                    // pinned host word: OVF_* bits (a capacity of this file was exceeded)
    // zero-initialised scratch of this call / the zone this call clears for the next one
    int *zone, *zone_next;
    size_t zone_stride, zone_next_clear;  // ints per frame; ints of zone_next this call has to clear
    int off_hyp, off_ctr, off_mom, off_scan;  // layout of a frame's zone: cell_count at 0, inlier counts, counters, moments, scan words
};

struct Cam {
    double x, y, z;
};
// D1, the statements of oracle_depth_estimate: camera-frame position of a return ...
__device__ __forceinline__ Cam cam_point(const DepthView& d, const float4 q) {
    const double x = q.x, y = q.y, z = q.z;
    return {d.R[0] * x + d.R[1] * y + d.R[2] * z + d.t[0], d.R[3] * x + d.R[4] * y + d.R[5] * z + d.t[1],
            d.R[6] * x + d.R[7] * y + d.R[8] * z + d.t[2]};
}
// ... and its pixel; false if it is cut (behind the camera, yaml:168) or outside the image
__device__ __forceinline__ bool pixel_of(const DepthView& d, const Cam& c, double* u, double* v) {
    if (d.p.do_use_cut_behind_camera && !(c.z > 0.0)) return false;
    if (c.z == 0.0) return false;
    *u = d.f * c.x / c.z + d.cx;
    *v = d.f * c.y / c.z + d.cy;
    return *u >= 0.0 && *u < (double)d.img_w && *v >= 0.0 && *v < (double)d.img_h;
}
__device__ __forceinline__ bool in_band(const DepthView& d, const float4 q) {
    const double z = q.z;
    return z >= d.p.ransac_plane_min_z && z <= d.p.ransac_plane_max_z;
}
__device__ __forceinline__ float4 load_return(const float* cloud, int i) { return reinterpret_cast<const float4*>(cloud)[i]; }

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// agent-scope accesses to the scan words of k_project (flag and value travel in ONE 64-bit word, so a relaxed atomic
// load that sees the flag has the value)
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------ D1 + band compaction
constexpr unsigned long long kScanAggregate = 1ull << 32, kScanPrefix = 2ull << 32;

__global__ __launch_bounds__(256) void k_project(DepthView d) {
    const int f = blockIdx.y;
    {  // clear the other zone for the next call
        const size_t gsz = (size_t)gridDim.x * gridDim.y * 256;
        for (size_t k = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; k < d.zone_next_clear; k += gsz) d.zone_next[k] = 0;
    }
    int* zone = d.zone + (size_t)f * d.zone_stride;
    __shared__ int s_ticket, s_base, wave_cnt[4];
    // workgroups take their segment of the sweep in the order they start: a workgroup only ever waits for earlier ones
    if (threadIdx.x == 0) s_ticket = atomicAdd(&zone[d.off_ctr + CTR_TICKET], 1);
    __syncthreads();
    const int blk = s_ticket;
    const int n_pts = d.n_pts[f];
    const int nblk = (n_pts + 255) / 256;
    if (blk >= nblk) {
        if (blk == 0 && threadIdx.x == 0) d.band_n[f] = 0;
        return;
    }
    const int i = blk * 256 + threadIdx.x;
    const bool live = i < n_pts;
    const size_t fo = (size_t)f * d.pt_stride;
    float4 q = {0.f, 0.f, 0.f, 0.f};
    if (live) q = load_return(d.cloud[f], i);  // 16-byte coalesced read
    double u, v;
    if (live && pixel_of(d, cam_point(d, q), &u, &v)) {
        const int cell = ((int)v / kCell) * d.cells_x + (int)u / kCell;
        const int pos = atomicAdd(&zone[cell], 1);
        if (pos < kCellCap) d.cell_pts[((size_t)f * d.n_cells + cell) * kCellCap + pos] = i;
    }
    if (!((d.ground_mask >> f) & 1u)) return;  // no ground plane wanted for this frame (uniform over the workgroup)
    // ---- D6a: returns with lidar z inside [min_z, max_z], kept in index order
    const bool flag = live && in_band(d, q);
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    if (wave == 0) {
        const int agg = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        unsigned long long* st = reinterpret_cast<unsigned long long*>(zone + d.off_scan);
        if (lane == 0) st_agent(&st[blk], (blk == 0 ? kScanPrefix : kScanAggregate) | (unsigned)agg);
        int excl = 0;
        if (blk > 0) {
            // look back over the earlier workgroups, 64 at a time (lane 0 = nearest): add aggregates up to and including
            // the nearest published inclusive prefix
            for (int look = blk - 1;; look -= 64) {
                const int idx = look - lane;
                unsigned long long w;
                do {
                    w = idx >= 0 ? ld_agent(&st[idx]) : kScanPrefix;
                } while (__any((w >> 32) == 0));
                const unsigned long long pm = __ballot((w >> 32) == 2);
                const int first_p = pm ? __ffsll((long long)pm) - 1 : 64;
                int val = lane <= first_p ? (int)(unsigned)(w & 0xffffffffull) : 0;
                for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off, 64);
                excl += val;
                if (pm) break;
            }
            if (lane == 0) st_agent(&st[blk], kScanPrefix | (unsigned)(excl + agg));
        }
        if (lane == 0) {
            s_base = excl;
            if (blk == nblk - 1) d.band_n[f] = excl + agg;
        }
    }
    __syncthreads();
    if (flag) {
        int off = s_base;
        for (int k = 0; k < wave; ++k) off += wave_cnt[k];
        d.band_idx[fo + off + __popcll(m & ((1ull << lane) - 1ull))] = i;
    }
}

// ------------------------------------------------------------------------------------------ D6a ground plane
// vertex v of hypothesis `it`: position in the band list (oracle_ground_plane draws the same numbers)
__device__ __forceinline__ size_t hyp_vertex(const DepthView& d, int nb, int it, int v) {
    const uint64_t h = splitmix64(d.p.ransac_seed * 0x100000001B3ull + (uint64_t)it);
    return splitmix64(h + (uint64_t)v) % (uint64_t)nb;
}
__device__ __forceinline__ bool plane_through(const double* a, const double* b, const double* c, double* pl) {
    const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (!(nn > 1e-9)) return false;
    pl[0] = n[0] / nn;
    pl[1] = n[1] / nn;
    pl[2] = n[2] / nn;
    pl[3] = -(pl[0] * a[0] + pl[1] * a[1] + pl[2] * a[2]);
    return true;
}
__device__ bool hyp_plane(const DepthView& d, int f, int nb, int it, double* pl) {
    if (nb < 3) return false;
    const size_t i0 = hyp_vertex(d, nb, it, 0), i1 = hyp_vertex(d, nb, it, 1), i2 = hyp_vertex(d, nb, it, 2);
    if (i0 == i1 || i0 == i2 || i1 == i2) return false;
    const int* bidx = d.band_idx + (size_t)f * d.pt_stride;
    const Cam A = cam_point(d, load_return(d.cloud[f], bidx[i0])), B = cam_point(d, load_return(d.cloud[f], bidx[i1])),
              C = cam_point(d, load_return(d.cloud[f], bidx[i2]));
    const double a[3] = {A.x, A.y, A.z}, b[3] = {B.x, B.y, B.z}, c[3] = {C.x, C.y, C.z};
    return plane_through(a, b, c, pl);
}

// Inlier counts: workgroup = (kHypPerBlock hypotheses, kChunk returns of the sweep).  The returns are loaded once (4 per
// lane, issued before anything else) and tested for the z band on the fly; 192 lanes draw and gather the 3 vertices of the
// 64 planes, 64 lanes build the planes (LDS); every wave then tests its returns against the planes (ballot + popcount:
// wave-uniform counts, lane k keeps the count of plane k).  Integer atomics make the totals order-independent.
//
// The sequential RANSAC semantics (keep the best so far, stop once the adaptive iteration bound
// k = log(1-p)/log(1-w^3) is reached) almost always stop inside the first 64 hypotheses (inlier ratio 0.5 => bound 35),
// so the counts come in two launches: <true> = hypotheses 0..63, then k_pick; <false> = the other hypotheses, whose
// workgroups leave at once unless k_pick asked for more (plane[6]).
template <bool FIRST>
__global__ __launch_bounds__(256) void k_ransac(DepthView d) {
    const int f = blockIdx.z;
    if (!((d.ground_mask >> f) & 1u)) return;
    if (!FIRST && d.plane[8 * (size_t)f + 6] == 0.0) return;
    const int n_pts = d.n_pts[f];
    const int q0 = blockIdx.y * kChunk;
    if (q0 >= n_pts) return;  // the grid is sized for the largest sweep of the call
    const int nb = d.band_n[f];
    if (nb < 3) return;
    __shared__ double vtx[kHypPerBlock][3][3];
    __shared__ int vidx[kHypPerBlock][3];
    __shared__ double pl[kHypPerBlock][4];
    __shared__ int valid[kHypPerBlock], bc[kHypPerBlock];
    int* hyp_count = d.zone + (size_t)f * d.zone_stride + d.off_hyp;
    const size_t fo = (size_t)f * d.pt_stride;
    const int n_hyp = d.n_hyp;
    const int h0 = (FIRST ? 0 : (int)blockIdx.x + 1) * kHypPerBlock;
    const int lane = threadIdx.x & 63;
    double x[4], y[4], z[4];
    bool live[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = q0 + k * 256 + (int)threadIdx.x;
        float4 r = {0.f, 0.f, 0.f, 0.f};
        if (q < n_pts) r = load_return(d.cloud[f], q);
        live[k] = q < n_pts && in_band(d, r);
        const Cam c = cam_point(d, r);
        x[k] = c.x;
        y[k] = c.y;
        z[k] = c.z;
    }
    if (threadIdx.x < 3 * kHypPerBlock) {
        const int hl = threadIdx.x / 3, v = threadIdx.x % 3;
        const size_t i = hyp_vertex(d, nb, min(h0 + hl, n_hyp - 1), v);
        vidx[hl][v] = (int)i;
        const Cam c = cam_point(d, load_return(d.cloud[f], d.band_idx[fo + i]));
        vtx[hl][v][0] = c.x;
        vtx[hl][v][1] = c.y;
        vtx[hl][v][2] = c.z;
    }
    __syncthreads();
    if (threadIdx.x < kHypPerBlock) {
        const int hl = threadIdx.x;
        double p4[4] = {0.0, 0.0, 0.0, 0.0};
        const bool distinct = vidx[hl][0] != vidx[hl][1] && vidx[hl][0] != vidx[hl][2] && vidx[hl][1] != vidx[hl][2];
        valid[hl] = h0 + hl < n_hyp && distinct && plane_through(vtx[hl][0], vtx[hl][1], vtx[hl][2], p4);
        for (int k = 0; k < 4; ++k) pl[hl][k] = p4[k];
        bc[hl] = 0;
    }
    __syncthreads();
    const double thr = d.p.ransac_plane_distance_treshold;
    int mine = 0;  // lane k: inliers of plane k among this wave's returns
#pragma unroll 4
    for (int k = 0; k < kHypPerBlock; ++k) {
        const double a = pl[k][0], b = pl[k][1], c = pl[k][2], dd = pl[k][3];
        int n = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) n += __popcll(__ballot(live[j] && fabs(a * x[j] + b * y[j] + c * z[j] + dd) < thr));
        if (lane == k) mine = n;
    }
    if (mine) atomicAdd(&bc[lane], mine);
    __syncthreads();
    if (threadIdx.x < kHypPerBlock && bc[threadIdx.x] && valid[threadIdx.x]) atomicAdd(&hyp_count[h0 + threadIdx.x], bc[threadIdx.x]);
}

// The sequential RANSAC loop over pre-computed counts [it0, it1): statements of oracle_ground_plane.
struct PickState {
    int best, bi;
    double k_needed;
};
__device__ __forceinline__ void pick_range(const DepthView& d, const int* hyp_count, int nb, int it0, int it1, PickState& s) {
    for (int it = it0; it < it1; ++it) {
        if (it >= s.k_needed) break;
        const int c = hyp_count[it];
        if (c > s.best) {
            s.best = c;
            s.bi = it;
            const double w = (double)c / (double)nb;
            const double denom = log(fmax(1e-300, 1.0 - w * w * w));
            s.k_needed = denom < 0 ? log(1.0 - d.p.ransac_plane_probability) / denom : 0.0;
        }
    }
}

// One wave per frame, after k_ransac<true>: the pick over hypotheses 0..63 and whether the loop would go on.
__global__ __launch_bounds__(64) void k_pick(DepthView d) {
    const int f = blockIdx.x;
    if (!((d.ground_mask >> f) & 1u)) return;
    __shared__ int cnts[kHypPerBlock];
    const int* hyp_count = d.zone + (size_t)f * d.zone_stride + d.off_hyp;
    cnts[threadIdx.x] = hyp_count[threadIdx.x];
    __syncthreads();
    if (threadIdx.x != 0) return;
    double* plane = d.plane + 8 * (size_t)f;
    const int nb = d.band_n[f], n_hyp = d.n_hyp, it1 = min(n_hyp, kHypPerBlock);
    PickState s = {0, -1, (double)n_hyp};
    if (nb >= 3) pick_range(d, cnts, nb, 0, it1, s);
    plane[5] = s.best;
    plane[6] = (nb >= 3 && it1 < n_hyp && s.k_needed > (double)it1) ? 1.0 : 0.0;  // the sequential loop would go on
    plane[7] = s.k_needed;
    d.pick[2 * f] = s.bi;
    d.pick[2 * f + 1] = s.best;
}

// The RANSAC plane of a frame once all counts exist (any lane may call it; every caller gets the same bits): the pick of
// k_pick, continued over the remaining hypotheses if that pick asked for them.
__device__ bool ransac_plane(const DepthView& d, int f, double* pl, int* inliers) {
    const double* plane = d.plane + 8 * (size_t)f;
    const int nb = d.band_n[f];
    PickState s = {d.pick[2 * f + 1], d.pick[2 * f], plane[7]};
    if (plane[6] != 0.0) pick_range(d, d.zone + (size_t)f * d.zone_stride + d.off_hyp, nb, kHypPerBlock, d.n_hyp, s);
    *inliers = s.best;
    pl[0] = pl[1] = pl[2] = pl[3] = 0.0;
    if (s.best < 3 || s.bi < 0) return false;
    return hyp_plane(d, f, nb, s.bi, pl);
}

// Refinement of the RANSAC plane over the band returns within refinement_treshold of it (yaml:138-140): centroid and
// scatter matrix from the moments of e = p - a around the point a = -d n of the RANSAC plane, accumulated in fixed point
// (round-to-nearest-even of e * 2^30 and of (e_i * e_j) * 2^20 into int64; integer sums are exact and order-free).
__device__ __forceinline__ long long fx(double v, double scale) { return __double2ll_rn(v * scale); }

__global__ __launch_bounds__(256) void k_refine(DepthView d) {
    const int f = blockIdx.y;
    if (!((d.ground_mask >> f) & 1u) || !d.p.ransac_plane_use_refinement) return;
    const int n_pts = d.n_pts[f];
    const int q0 = blockIdx.x * kChunk;
    if (q0 >= n_pts) return;
    __shared__ double s_pl[4];
    __shared__ int s_ok;
    __shared__ long long s_part[4][kMomVals];
    float4 r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = q0 + 4 * (int)threadIdx.x + k;
        r[k] = {0.f, 0.f, 0.f, 0.f};
        if (q < n_pts) r[k] = load_return(d.cloud[f], q);
    }
    if (threadIdx.x == 0) {
        double pl[4];
        int inl;
        s_ok = ransac_plane(d, f, pl, &inl);
        for (int k = 0; k < 4; ++k) s_pl[k] = pl[k];
    }
    __syncthreads();
    if (!s_ok) return;
    const double pl[4] = {s_pl[0], s_pl[1], s_pl[2], s_pl[3]};
    const double a[3] = {-pl[3] * pl[0], -pl[3] * pl[1], -pl[3] * pl[2]};
    long long acc[kMomVals] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = q0 + 4 * (int)threadIdx.x + k;
        if (q >= n_pts || !in_band(d, r[k])) continue;
        const Cam c = cam_point(d, r[k]);
        if (!(fabs(pl[0] * c.x + pl[1] * c.y + pl[2] * c.z + pl[3]) < d.p.ransac_plane_refinement_treshold)) continue;
        const double e[3] = {c.x - a[0], c.y - a[1], c.z - a[2]};
        if (!(fabs(e[0]) < kMomRange && fabs(e[1]) < kMomRange && fabs(e[2]) < kMomRange)) continue;
        acc[0] += 1;
        acc[1] += fx(e[0], kMomScale1);
        acc[2] += fx(e[1], kMomScale1);
        acc[3] += fx(e[2], kMomScale1);
        acc[4] += fx(e[0] * e[0], kMomScale2);
        acc[5] += fx(e[0] * e[1], kMomScale2);
        acc[6] += fx(e[0] * e[2], kMomScale2);
        acc[7] += fx(e[1] * e[1], kMomScale2);
        acc[8] += fx(e[1] * e[2], kMomScale2);
        acc[9] += fx(e[2] * e[2], kMomScale2);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < kMomVals; ++k) {
        long long v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) s_part[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kMomVals) {
        const long long v = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
        unsigned long long* mom = reinterpret_cast<unsigned long long*>(d.zone + (size_t)f * d.zone_stride + d.off_mom);
        if (v) atomicAdd(&mom[threadIdx.x], (unsigned long long)v);
    }
}

// One Jacobi rotation of the symmetric 3x3 matrix in the (I,J) plane, eigenvectors in V; indices are compile-time
// constants so that everything stays in registers.
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

// One wave per frame, after k_refine: the ground plane of the frame (n, d with n.p + d = 0, d >= 0: the camera is on the
// positive side).
__global__ __launch_bounds__(64) void k_plane(DepthView d) {
    const int f = blockIdx.x;
    if (!((d.ground_mask >> f) & 1u) || threadIdx.x != 0) return;
    double* plane = d.plane + 8 * (size_t)f;
    double pl[4];
    int inliers;
    const bool ok = ransac_plane(d, f, pl, &inliers);
    double n[3] = {pl[0], pl[1], pl[2]}, dd = pl[3];
    if (ok && d.p.ransac_plane_use_refinement) {
        const long long* mom = reinterpret_cast<const long long*>(d.zone + (size_t)f * d.zone_stride + d.off_mom);
        const double m0 = (double)mom[0];
        if (m0 >= 3.0) {
            const double a[3] = {-pl[3] * pl[0], -pl[3] * pl[1], -pl[3] * pl[2]};
            const double c[3] = {((double)mom[1] / kMomScale1) / m0, ((double)mom[2] / kMomScale1) / m0, ((double)mom[3] / kMomScale1) / m0};  // centroid - a
            const double S[6] = {(double)mom[4] / kMomScale2 - m0 * c[0] * c[0], (double)mom[5] / kMomScale2 - m0 * c[0] * c[1],
                                 (double)mom[6] / kMomScale2 - m0 * c[0] * c[2], (double)mom[7] / kMomScale2 - m0 * c[1] * c[1],
                                 (double)mom[8] / kMomScale2 - m0 * c[1] * c[2], (double)mom[9] / kMomScale2 - m0 * c[2] * c[2]};
            smallest_eigvec(S, n);
            dd = -(n[0] * (a[0] + c[0]) + n[1] * (a[1] + c[1]) + n[2] * (a[2] + c[2]));
        }
    }
    if (dd < 0) {
        n[0] = -n[0];
        n[1] = -n[1];
        n[2] = -n[2];
        dd = -dd;
    }
    plane[0] = n[0];
    plane[1] = n[1];
    plane[2] = n[2];
    plane[3] = dd;
    plane[4] = ok ? 1.0 : 0.0;
    plane[5] = inliers;
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
    int cell_off[65];       // exclusive prefix of the candidate counts of the cells under the rectangle
    int tmp_idx[kMaxNb];    // neighbours in arrival order
    double tmp_xyz[kMaxNb][3];
    int nb_idx[kMaxNb];     // neighbours ordered by return index
    double nb_xyz[kMaxNb][3];
    double seg[kMaxNb][3];  // points of the selected histogram bin / ground patch
    int bins[kMaxBins];
};

__global__ __launch_bounds__(256) void k_features(DepthView d) {
    __shared__ WaveLds lds[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int fr = blockIdx.y;
    const int k = blockIdx.x * 4 + wave;
    if (k >= d.n_feat[fr]) return;  // whole wave exits together
    WaveLds& L = lds[wave];
    const int* cell_count = d.zone + (size_t)fr * d.zone_stride;
    const int* cell_pts = d.cell_pts + (size_t)fr * d.n_cells * kCellCap;
    const double* plane = d.plane + 8 * (size_t)fr;
    const float* cloud = d.cloud[fr];
    const float* feat_uv = d.feat_uv[fr];
    const uint8_t* feat_ground = d.feat_ground[fr];
    const double fu = feat_uv[2 * (size_t)k], fv = feat_uv[2 * (size_t)k + 1];
    const double hw = 0.5 * d.p.pixelarea_search_width, hh = 0.5 * d.p.pixelarea_search_height;
    const double cu = fu + d.p.pixelarea_search_offset_x, cv = fv + d.p.pixelarea_search_offset_y;
    // ---- D2: candidates = the returns listed in the cells under the rectangle.  All cells in one pass: lane c takes the
    //      count of cell c, a wave prefix sum gives every candidate a lane (three dependent loads per pass - count, index,
    //      record - instead of three per cell); the rectangle test uses the pixel recomputed from the record.
    const int cx0 = max(0, (int)floor((cu - hw) / kCell)), cx1 = min(d.cells_x - 1, (int)floor((cu + hw) / kCell));
    const int cy0 = max(0, (int)floor((cv - hh) / kCell)), cy1 = min(d.cells_y - 1, (int)floor((cv + hh) / kCell));
    const int ncx = max(0, cx1 - cx0 + 1), ncell = ncx * max(0, cy1 - cy0 + 1);
    int n = 0;
    unsigned ovf = 0;
    for (int c0 = 0; c0 < ncell; c0 += 64) {
        const int c = c0 + lane;
        int cell = 0, cnt = 0;
        if (c < ncell) {
            cell = (cy0 + c / ncx) * d.cells_x + cx0 + c % ncx;
            cnt = cell_count[cell];
            if (cnt > kCellCap) {
                ovf |= OVF_CELL;
                cnt = kCellCap;
            }
        }
        int incl = cnt;  // inclusive prefix sum over the lanes
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const int total = __shfl(incl, 63, 64);
        L.cell_off[lane] = incl - cnt;
        if (lane == 63) L.cell_off[64] = total;
        __builtin_amdgcn_wave_barrier();
        for (int j0 = 0; j0 < total; j0 += 64) {
            const int j = j0 + lane;
            bool in = false;
            int idx = -1;
            Cam P = {0.0, 0.0, 0.0};
            if (j < total) {
                int lo = 0, hi = 63;  // the cell of candidate j: last lane whose offset is <= j
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (L.cell_off[mid] <= j) lo = mid; else hi = mid - 1;
                }
                const int cc = c0 + lo;
                const int cl = (cy0 + cc / ncx) * d.cells_x + cx0 + cc % ncx;
                idx = cell_pts[(size_t)cl * kCellCap + (j - L.cell_off[lo])];
                P = cam_point(d, load_return(cloud, idx));
                double u, v;
                in = pixel_of(d, P, &u, &v) && fabs(u - cu) <= hw && fabs(v - cv) <= hh;
            }
            const unsigned long long m = __ballot(in);
            if (in) {
                const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
                if (pos < kMaxNb) {
                    L.tmp_idx[pos] = idx;
                    L.tmp_xyz[pos][0] = P.x;
                    L.tmp_xyz[pos][1] = P.y;
                    L.tmp_xyz[pos][2] = P.z;
                }
            }
            n += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (n > kMaxNb) {
        ovf |= OVF_NEIGHBOURS;
        n = kMaxNb;
    }
    if (ovf) *reinterpret_cast<volatile int*>(d.overflow) = (int)ovf;  // pinned host word; the call then fails with LIMO_ERR_INVALID
    // order by return index (rank sort inside the wave) so every later step sees the lidar order the oracle sees
    __builtin_amdgcn_wave_barrier();
    if (lane < n) {
        const int mine = L.tmp_idx[lane];
        int rank = 0;
        for (int q = 0; q < n; ++q) rank += L.tmp_idx[q] < mine;
        L.nb_idx[rank] = mine;
        L.nb_xyz[rank][0] = L.tmp_xyz[lane][0];
        L.nb_xyz[rank][1] = L.tmp_xyz[lane][1];
        L.nb_xyz[rank][2] = L.tmp_xyz[lane][2];
    }
    __builtin_amdgcn_wave_barrier();
    float result = -1.0f;
    if (n >= d.p.neighbors_count_min) {
        const bool ground_feat = feat_ground && feat_ground[k] && plane[4] != 0.0;
        double depth = -1.0, zlo = 0.0, zhi = 0.0;
        bool have = false;
        if (ground_feat) {
            // ---- D6b: inverse-distance weighted patch over the neighbours close to the sweep's ground plane (lane 0,
            //      sequential in return order: the oracle's summation order)
            if (lane == 0) {
                const double gn[3] = {plane[0], plane[1], plane[2]}, gd = plane[3];
                double sw = 0, c[3] = {0, 0, 0};
                int m = 0;
                zlo = 1.79769313486231570e308;
                zhi = -zlo;
                double* wgt = reinterpret_cast<double*>(L.bins);  // weights parked in the (unused) histogram storage
                for (int q = 0; q < n; ++q) {
                    const double p[3] = {L.nb_xyz[q][0], L.nb_xyz[q][1], L.nb_xyz[q][2]};
                    const double dist = gn[0] * p[0] + gn[1] * p[1] + gn[2] * p[2] + gd;
                    if (fabs(dist) < d.p.ransac_plane_point_distance_treshold) {
                        const double w = d.p.plane_estimator_use_mestimator ? 1.0 / (fabs(dist) + 0.01) : 1.0;
                        L.seg[m][0] = p[0];
                        L.seg[m][1] = p[1];
                        L.seg[m][2] = p[2];
                        wgt[m] = w;
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
                        const double w = wgt[q];
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
                myz = L.nb_xyz[lane][2];
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
                const double span = floor((zmax - zmin) / bw);
                if (!(span < (double)kMaxBins)) {
                    seg_ok = false;  // depth span beyond 150 m inside one 6x9 px window: treat as unsegmentable
                } else {
                    const int nbins = (int)span + 1;
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
                            L.seg[pos][0] = L.nb_xyz[lane][0];
                            L.seg[pos][1] = L.nb_xyz[lane][1];
                            L.seg[pos][2] = L.nb_xyz[lane][2];
                        }
                        nseg = __popcll(m);
                    }
                }
            } else {
                if (lane < n) {
                    L.seg[lane][0] = L.nb_xyz[lane][0];
                    L.seg[lane][1] = L.nb_xyz[lane][1];
                    L.seg[lane][2] = L.nb_xyz[lane][2];
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
    if (lane == 0) d.out[fr][k] = result;
}

// ------------------------------------------------------------------------------------------ workspace
struct DepthWs {
    size_t cap_pts = 0, cap_feat = 0, cap_cells = 0;
    int cap_frames = 0;
    size_t zone_stride = 0;          // ints per frame
    int off_hyp = 0, off_ctr = 0, off_mom = 0, off_scan = 0;
    int zone_used[2] = {0, 0};       // frames of each zone a call has written into since it was last cleared
    int cur = 0;                     // zone of the next call
    int last_frames = 0;             // frames of the last launch group (limo_depth_last_ground_plane)
    uint32_t last_ground_mask = 0;
    bool last_timed = false;         // the last enqueued group recorded its events
    bool open = false;               // a limo_depth_estimate_begin whose _end has not been called
    size_t open_feat = 0;            // its n_feat
    bool timing = false;             // limo_depth_set_timing: HIP events around the kernels of a launch group
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // before k_project | k_ransac.. | k_features | copy back | end
    double last_ms[4] = {0, 0, 0, 0};  // k_project, ground-plane kernels, k_features, all kernels
    float* cloud = nullptr;
    int *cell_pts = nullptr, *band_idx = nullptr, *band_n = nullptr, *pick = nullptr, *zone[2] = {nullptr, nullptr};
    double* plane = nullptr;
    float *feat_uv = nullptr, *out = nullptr;
    uint8_t* feat_ground = nullptr;
    float* h_feat = nullptr;   // pinned staging: uv of every frame, then the ground labels; and the depths coming back
    float* h_out = nullptr;
    int* h_ovf = nullptr;      // pinned word the kernels raise when a capacity of this file is exceeded
    void release() {
        void* ptrs[] = {cloud, cell_pts, band_idx, band_n, pick, zone[0], zone[1], plane, feat_uv, out, feat_ground};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        if (h_feat) (void)hipHostFree(h_feat);
        if (h_out) (void)hipHostFree(h_out);
        if (h_ovf) (void)hipHostFree(h_ovf);
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
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

size_t round_up(size_t v, size_t q) { return (v + q - 1) / q * q; }

// (Re)allocate the workspace for `frames` sweeps of up to n_pts returns / n_feat features on a cells-cell image.
int ensure_capacity(limo_ctx* ctx, DepthWs& W, int frames, size_t n_pts, size_t n_feat, size_t cells) {
    if (frames <= W.cap_frames && n_pts <= W.cap_pts && n_feat <= W.cap_feat && cells <= W.cap_cells) return LIMO_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int F = std::max(frames, W.cap_frames);
    const size_t P = round_up(std::max({n_pts, W.cap_pts, (size_t)1}), 256), Q = round_up(std::max({n_feat, W.cap_feat, (size_t)1}), 64),
                 C = std::max(cells, W.cap_cells);
    const size_t nblk = P / 256;
    int rc = LIMO_OK;
    rc |= grow(ctx, &W.cloud, (size_t)F * P * 4);
    rc |= grow(ctx, &W.band_idx, (size_t)F * P);
    rc |= grow(ctx, &W.cell_pts, (size_t)F * C * kCellCap);
    rc |= grow(ctx, &W.band_n, (size_t)F);
    rc |= grow(ctx, &W.pick, (size_t)F * 2);
    rc |= grow(ctx, &W.plane, (size_t)F * 8);
    rc |= grow(ctx, &W.feat_uv, (size_t)F * Q * 2);
    rc |= grow(ctx, &W.feat_ground, (size_t)F * Q);
    rc |= grow(ctx, &W.out, (size_t)F * Q);
    // zone of a frame: cell counters | inlier counts | counters | moments (64-bit) | scan words (64-bit, 8-byte aligned)
    W.off_hyp = (int)round_up(C, 2);
    W.off_ctr = W.off_hyp + kMaxHyp;
    W.off_mom = W.off_ctr + CTR_COUNT;
    W.off_scan = W.off_mom + 2 * kMomVals;
    W.zone_stride = round_up((size_t)W.off_scan + 2 * nblk, 2);
    for (int z = 0; z < 2; ++z) {
        rc |= grow(ctx, &W.zone[z], (size_t)F * W.zone_stride);
        if (rc == LIMO_OK) HIP_TRY(ctx, hipMemsetAsync(W.zone[z], 0, sizeof(int) * (size_t)F * W.zone_stride, ctx->stream));
        W.zone_used[z] = 0;
    }
    if (W.h_feat) (void)hipHostFree(W.h_feat);
    if (W.h_out) (void)hipHostFree(W.h_out);
    W.h_feat = W.h_out = nullptr;
    if (rc == LIMO_OK) {
        HIP_TRY(ctx, hipHostMalloc((void**)&W.h_feat, (size_t)F * Q * (2 * sizeof(float) + 1)));
        HIP_TRY(ctx, hipHostMalloc((void**)&W.h_out, (size_t)F * Q * sizeof(float)));
        if (!W.h_ovf) HIP_TRY(ctx, hipHostMalloc((void**)&W.h_ovf, sizeof(int)));
    }
    if (rc != LIMO_OK) {
        W.cap_frames = 0;
        W.cap_pts = W.cap_feat = W.cap_cells = 0;
        return LIMO_ERR_RUNTIME;
    }
    W.cap_frames = F;
    W.cap_pts = P;
    W.cap_feat = Q;
    W.cap_cells = C;
    return LIMO_OK;
}

// camera <- lidar rotation from the quaternion (w,x,y,z): the statements of the oracle's quat_to_R (contraction off)
void quat_to_R(const double* q, double* R) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z);
    R[1] = 2 * (x * y - w * z);
    R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);
    R[4] = 1 - 2 * (x * x + z * z);
    R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);
    R[7] = 2 * (y * z + w * x);
    R[8] = 1 - 2 * (x * x + y * y);
}

// One launch group: 1..kMaxBatch frames.  enqueue_group puts the copies and the kernels on the context's stream, finish_group
// waits for them and reports what the kernels flagged; run_group = both + the depths into the caller's arrays.
int enqueue_group(limo_ctx* ctx, int n_frames, const limo_depth_frame* frames, const double* T_cam_lidar, double f, double cx, double cy,
              int32_t img_w, int32_t img_h, const limo_depth_params& p, bool device_ptrs) {
    if (!ctx->depth_ws) {
        ctx->depth_ws = new DepthWs();
        ctx->depth_ws_free = depth_ws_free;
    }
    DepthWs& W = *static_cast<DepthWs*>(ctx->depth_ws);
    if (W.open) {
        ctx->err = "limo_depth_estimate: a limo_depth_estimate_begin is open on this context (limo_depth_estimate_end first)";
        return LIMO_ERR_INVALID;
    }
    hipStream_t s = ctx->stream;
    const int cells_x = (img_w + kCell - 1) / kCell, cells_y = (img_h + kCell - 1) / kCell;
    const size_t cells = (size_t)cells_x * cells_y;
    size_t max_pts = 0, max_feat = 0;
    for (int k = 0; k < n_frames; ++k) {
        max_pts = std::max(max_pts, frames[k].n_pts);
        max_feat = std::max(max_feat, frames[k].n_feat);
    }
    if (max_pts > (1u << 22) || max_feat > 0x7fffff00u) {  // 2^22 returns: the fixed-point moments cannot overflow
        ctx->err = "limo_depth_estimate: more than 2^22 returns in one sweep";
        return LIMO_ERR_INVALID;
    }
    if (int rc = ensure_capacity(ctx, W, n_frames, max_pts, max_feat, cells)) return rc;

    DepthView d;
    std::memset(&d, 0, sizeof(d));
    quat_to_R(T_cam_lidar, d.R);
    for (int i = 0; i < 3; ++i) d.t[i] = T_cam_lidar[4 + i];
    d.f = f;
    d.cx = cx;
    d.cy = cy;
    d.img_w = img_w;
    d.img_h = img_h;
    d.cells_x = cells_x;
    d.cells_y = cells_y;
    d.n_cells = (int)cells;
    d.n_frames = n_frames;
    d.n_hyp = std::max(1, p.ransac_plane_max_iterations);
    d.p = p;
    d.pt_stride = W.cap_pts;
    d.cell_pts = W.cell_pts;
    d.band_idx = W.band_idx;
    d.band_n = W.band_n;
    d.plane = W.plane;
    d.pick = W.pick;
    d.overflow = W.h_ovf;
    const int z = W.cur;
    d.zone = W.zone[z];
    d.zone_next = W.zone[z ^ 1];
    d.zone_stride = W.zone_stride;
    d.zone_next_clear = (size_t)W.zone_used[z ^ 1] * W.zone_stride;
    d.off_hyp = W.off_hyp;
    d.off_ctr = W.off_ctr;
    d.off_mom = W.off_mom;
    d.off_scan = W.off_scan;
    *W.h_ovf = 0;

    const size_t Q = W.cap_feat;
    float* h_uv = W.h_feat;
    uint8_t* h_ground = reinterpret_cast<uint8_t*>(W.h_feat + (size_t)W.cap_frames * Q * 2);
    bool any_ground = false;
    for (int k = 0; k < n_frames; ++k) {
        const limo_depth_frame& fr = frames[k];
        d.n_pts[k] = (int)fr.n_pts;
        d.n_feat[k] = (int)fr.n_feat;
        bool ground = false;
        if (device_ptrs) {
            d.cloud[k] = fr.cloud_xyzi;
            d.feat_uv[k] = fr.feat_uv;
            d.feat_ground[k] = fr.feat_is_ground;
            d.out[k] = fr.depth_out;
            ground = fr.feat_is_ground != nullptr && fr.n_feat > 0;
        } else {
            d.cloud[k] = W.cloud + (size_t)k * W.cap_pts * 4;
            d.feat_uv[k] = W.feat_uv + (size_t)k * Q * 2;
            d.out[k] = W.out + (size_t)k * Q;
            if (fr.n_pts) HIP_TRY(ctx, hipMemcpyAsync(W.cloud + (size_t)k * W.cap_pts * 4, fr.cloud_xyzi, sizeof(float) * 4 * fr.n_pts, hipMemcpyHostToDevice, s));
            if (fr.n_feat) std::memcpy(h_uv + (size_t)k * Q * 2, fr.feat_uv, sizeof(float) * 2 * fr.n_feat);
            if (fr.feat_is_ground)
                for (size_t j = 0; j < fr.n_feat && !ground; ++j) ground = fr.feat_is_ground[j] != 0;
            if (ground) std::memcpy(h_ground + (size_t)k * Q, fr.feat_is_ground, fr.n_feat);
            d.feat_ground[k] = ground ? W.feat_ground + (size_t)k * Q : nullptr;
        }
        if (ground && p.do_use_ransac_plane && fr.n_pts) d.ground_mask |= 1u << k;
        if (!(ground && p.do_use_ransac_plane && fr.n_pts)) d.feat_ground[k] = nullptr;  // no plane => nothing takes the ground path
        any_ground = any_ground || ground;
    }
    if (!device_ptrs && max_feat) {
        HIP_TRY(ctx, hipMemcpyAsync(W.feat_uv, h_uv, sizeof(float) * 2 * Q * n_frames, hipMemcpyHostToDevice, s));
        if (any_ground) HIP_TRY(ctx, hipMemcpyAsync(W.feat_ground, h_ground, Q * n_frames, hipMemcpyHostToDevice, s));
    }
    W.zone_used[z] = std::max(W.zone_used[z], n_frames);
    W.zone_used[z ^ 1] = 0;  // cleared by this call's k_project (or already clean)
    W.cur = z ^ 1;
    W.last_frames = n_frames;
    W.last_ground_mask = d.ground_mask;
    const unsigned F = (unsigned)n_frames;
    auto mark = [&](int k) {
        if (W.timing) (void)hipEventRecord(W.ev[k], s);
    };
    mark(0);
    // the projection kernel also clears the other zone: it runs even for a call without returns
    hipLaunchKernelGGL(k_project, dim3((unsigned)std::max<size_t>(1, (max_pts + 255) / 256), F), dim3(256), 0, s, d);
    mark(1);
    if (d.ground_mask) {
        const unsigned n_chunk = (unsigned)((max_pts + kChunk - 1) / kChunk);
        const unsigned n_groups = (unsigned)((d.n_hyp + kHypPerBlock - 1) / kHypPerBlock);
        hipLaunchKernelGGL(k_ransac<true>, dim3(1, n_chunk, F), dim3(256), 0, s, d);
        hipLaunchKernelGGL(k_pick, dim3(F), dim3(64), 0, s, d);
        if (n_groups > 1) hipLaunchKernelGGL(k_ransac<false>, dim3(n_groups - 1, n_chunk, F), dim3(256), 0, s, d);
        if (p.ransac_plane_use_refinement) hipLaunchKernelGGL(k_refine, dim3(n_chunk, F), dim3(256), 0, s, d);
        hipLaunchKernelGGL(k_plane, dim3(F), dim3(64), 0, s, d);
    }
    mark(2);
    if (max_feat) {
        hipLaunchKernelGGL(k_features, dim3((unsigned)((max_feat + 3) / 4), F), dim3(256), 0, s, d);
        mark(3);
        if (!device_ptrs) HIP_TRY(ctx, hipMemcpyAsync(W.h_out, W.out, sizeof(float) * Q * n_frames, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(ctx, hipGetLastError());
    W.last_timed = W.timing && max_feat;
    return LIMO_OK;
}

int finish_group(limo_ctx* ctx, DepthWs& W) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (W.last_timed) {
        float a = 0.f, b = 0.f, c = 0.f;
        (void)hipEventElapsedTime(&a, W.ev[0], W.ev[1]);
        (void)hipEventElapsedTime(&b, W.ev[1], W.ev[2]);
        (void)hipEventElapsedTime(&c, W.ev[2], W.ev[3]);
        W.last_ms[0] = a;
        W.last_ms[1] = b;
        W.last_ms[2] = c;
        W.last_ms[3] = (double)a + b + c;
    }
    if (*W.h_ovf) {
        ctx->err = std::string("limo_depth_estimate: ") + ((*W.h_ovf & OVF_CELL) ? "more than 48 returns project into one 8x8 px image cell" : "more than 64 returns inside one search rectangle") +
                   " (not a single sweep of a spinning scanner?)";
        return LIMO_ERR_INVALID;
    }
    return LIMO_OK;
}

int run_group(limo_ctx* ctx, int n_frames, const limo_depth_frame* frames, const double* T_cam_lidar, double f, double cx, double cy,
              int32_t img_w, int32_t img_h, const limo_depth_params& p, bool device_ptrs) {
    if (int rc = enqueue_group(ctx, n_frames, frames, T_cam_lidar, f, cx, cy, img_w, img_h, p, device_ptrs)) return rc;
    DepthWs& W = *static_cast<DepthWs*>(ctx->depth_ws);
    if (int rc = finish_group(ctx, W)) return rc;
    if (!device_ptrs)
        for (int k = 0; k < n_frames; ++k)
            if (frames[k].n_feat) std::memcpy(frames[k].depth_out, W.h_out + (size_t)k * W.cap_feat, sizeof(float) * frames[k].n_feat);
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

int limo_depth_estimate_batch(limo_ctx* ctx, int32_t n_frames, const limo_depth_frame* frames, const double* T_cam_lidar, double f,
                              double cx, double cy, int32_t img_w, int32_t img_h, const limo_depth_params* params, uint32_t flags) {
    if (!ctx || n_frames < 0 || (n_frames && !frames) || !T_cam_lidar || img_w <= 0 || img_h <= 0 || (flags & ~(uint32_t)LIMO_DEPTH_DEVICE_POINTERS))
        return LIMO_ERR_INVALID;
    for (int k = 0; k < n_frames; ++k)
        if ((frames[k].n_pts && !frames[k].cloud_xyzi) || (frames[k].n_feat && (!frames[k].feat_uv || !frames[k].depth_out))) return LIMO_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return LIMO_ERR_NO_DEVICE;
    limo_depth_params p;
    if (params)
        p = *params;
    else
        limo_depth_default_params(&p);
    if (p.ransac_plane_max_iterations > kMaxHyp) p.ransac_plane_max_iterations = kMaxHyp;
    for (int k0 = 0; k0 < n_frames; k0 += kMaxBatch)
        if (int rc = run_group(ctx, std::min(kMaxBatch, n_frames - k0), frames + k0, T_cam_lidar, f, cx, cy, img_w, img_h, p,
                               (flags & LIMO_DEPTH_DEVICE_POINTERS) != 0))
            return rc;
    return LIMO_OK;
}

int limo_depth_set_timing(limo_ctx* ctx, int32_t on) {
    if (!ctx) return LIMO_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return LIMO_ERR_NO_DEVICE;
    if (!ctx->depth_ws) {
        ctx->depth_ws = new DepthWs();
        ctx->depth_ws_free = depth_ws_free;
    }
    DepthWs& W = *static_cast<DepthWs*>(ctx->depth_ws);
    if (on && !W.ev[0])
        for (hipEvent_t& e : W.ev) HIP_TRY(ctx, hipEventCreate(&e));
    W.timing = on != 0;
    return LIMO_OK;
}

int limo_depth_last_kernel_ms(limo_ctx* ctx, double* ms4) {
    if (!ctx || !ms4 || !ctx->depth_ws) return LIMO_ERR_INVALID;
    const DepthWs& W = *static_cast<DepthWs*>(ctx->depth_ws);
    for (int k = 0; k < 4; ++k) ms4[k] = W.last_ms[k];
    return LIMO_OK;
}

int limo_depth_last_ground_plane(limo_ctx* ctx, int32_t frame, double* plane4, int32_t* inliers) {
    if (!ctx || !plane4 || !ctx->depth_ws) return LIMO_ERR_INVALID;
    DepthWs& W = *static_cast<DepthWs*>(ctx->depth_ws);
    if (frame < 0 || frame >= W.last_frames) return LIMO_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return LIMO_ERR_NO_DEVICE;
    double pl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((W.last_ground_mask >> frame) & 1u) {
        HIP_TRY(ctx, hipMemcpyAsync(pl, W.plane + 8 * (size_t)frame, sizeof(pl), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    const bool ok = pl[4] != 0.0;
    for (int k = 0; k < 4; ++k) plane4[k] = ok ? pl[k] : 0.0;
    if (inliers) *inliers = ok ? (int32_t)pl[5] : 0;
    return LIMO_OK;
}

int limo_depth_estimate_begin(limo_ctx* ctx, const float* cloud_xyzi, size_t n_pts, const double* T_cam_lidar, double f, double cx,
                              double cy, int32_t img_w, int32_t img_h, const float* feat_uv, size_t n_feat, const uint8_t* feat_is_ground,
                              const limo_depth_params* params) {
    if (!ctx || !T_cam_lidar || img_w <= 0 || img_h <= 0 || (n_pts && !cloud_xyzi) || (n_feat && !feat_uv)) return LIMO_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return LIMO_ERR_NO_DEVICE;
    limo_depth_params p;
    if (params)
        p = *params;
    else
        limo_depth_default_params(&p);
    if (p.ransac_plane_max_iterations > kMaxHyp) p.ransac_plane_max_iterations = kMaxHyp;
    limo_depth_frame fr;
    fr.cloud_xyzi = cloud_xyzi;
    fr.n_pts = n_pts;
    fr.feat_uv = feat_uv;
    fr.n_feat = n_feat;
    fr.feat_is_ground = feat_is_ground;
    fr.depth_out = nullptr;
    if (int rc = enqueue_group(ctx, 1, &fr, T_cam_lidar, f, cx, cy, img_w, img_h, p, false)) return rc;
    DepthWs& W = *static_cast<DepthWs*>(ctx->depth_ws);
    W.open = true;
    W.open_feat = n_feat;
    return LIMO_OK;
}

int limo_depth_estimate_end(limo_ctx* ctx, float* depth_out, size_t n_feat) {
    if (!ctx || !ctx->depth_ws) return LIMO_ERR_INVALID;
    DepthWs& W = *static_cast<DepthWs*>(ctx->depth_ws);
    if (!W.open) {
        ctx->err = "limo_depth_estimate_end: no limo_depth_estimate_begin is open on this context";
        return LIMO_ERR_INVALID;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) return LIMO_ERR_NO_DEVICE;
    W.open = false;  // (whatever happens below, the call is over)
    if (int rc = finish_group(ctx, W)) return rc;
    if (n_feat != W.open_feat || (n_feat && !depth_out)) {
        ctx->err = "limo_depth_estimate_end: n_feat differs from the limo_depth_estimate_begin call";
        return LIMO_ERR_INVALID;
    }
    if (n_feat) std::memcpy(depth_out, W.h_out, sizeof(float) * n_feat);
    return LIMO_OK;
}

int limo_depth_estimate(limo_ctx* ctx, const float* cloud_xyzi, size_t n_pts, const double* T_cam_lidar, double f,
                        double cx, double cy, int32_t img_w, int32_t img_h, const float* feat_uv, size_t n_feat,
                        const uint8_t* feat_is_ground, const limo_depth_params* params, float* depth_out) {
    limo_depth_frame fr;
    fr.cloud_xyzi = cloud_xyzi;
    fr.n_pts = n_pts;
    fr.feat_uv = feat_uv;
    fr.n_feat = n_feat;
    fr.feat_is_ground = feat_is_ground;
    fr.depth_out = depth_out;
    return limo_depth_estimate_batch(ctx, 1, &fr, T_cam_lidar, f, cx, cy, img_w, img_h, params, 0);
}

}  // extern "C"
