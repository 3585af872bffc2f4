// depth.hip — LiDAR -> feature depth assignment on gfx950 (SURVEY §8a rows D1–D6; C-ABI limo_depth_estimate and
// limo_depth_estimate_batch).
//
// Replaces the (un-vendored) mono_lidar_depth DepthEstimator as pinned by
// demo_keyframe_bundle_adjustment_meta/res/mono_lidar_fusion_parameters.yaml (cited as yaml:LINE); output contract
// FeaturePoint::d (matches_msg_types/include/matches_msg_types/feature_point.hpp:24-26).  Algorithmic choices the
// parameter file leaves open are the ones documented in oracle/depth_oracle.cpp (the test oracle of this path).
//
// Five launches per call, every one with the frame as a grid dimension (a call carries 1..kMaxBatch sweeps):
//   k_project    1 lane / lidar return   HBM   D1: lidar->camera, cut z<=0, pinhole projection, in-image test; visible
//                                              returns are binned into 8x8-pixel image cells (index lists).  D6a, same
//                                              pass: order-preserving compaction of the returns inside the z band
//                                              (single-pass decoupled look-back scan over the workgroups of a frame).
//                                              Also clears the OTHER scratch zone (cell counters, inlier counters, scan
//                                              state) for the next call: no memset launches.
//   k_ransac<first>, k_ransac<rest>   workgroup = (64 hypotheses, 1024 band returns): planes from seeded draws,
//                inlier counts as integer atomics; the last workgroup of a frame to finish runs the sequential
//                best-so-far / adaptive iteration bound over the counts.  <first> covers hypotheses 0..63, which is where
//                the sequential loop almost always stops; the workgroups of <rest> leave at once unless it did not.
//   k_refine     moments of the band returns near the RANSAC plane in one pass, fixed two-level summation order
//                (deterministic); the last workgroup adds the chunk sums and finishes the plane (smallest eigenvector of
//                the scatter matrix, sign).
//   k_features   1 wave / feature   D2: gather the returns of the cells under the 6x9 px rectangle (ballot compaction
//                into LDS, ordered by return index), D3: depth histogram in LDS + nearest local maximum, D4: largest
//                triangle as a wave reduction over point pairs, plane, ray intersection, D5: gates; D6b: ground features
//                use the inverse-distance weighted patch.
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
constexpr int kMaxBatch = 32;   // frames per launch group (per-frame sizes and pointers travel as kernel arguments)
constexpr int kHypPerBlock = 64;   // = lanes of a wave: lane k keeps the count of plane k
constexpr int kRansacChunk = 1024;  // band returns per workgroup, 4 per lane
constexpr int kRefineChunk = 1024;
enum { CTR_TICKET = 0, CTR_RANSAC = 1, CTR_RANSAC2 = 2, CTR_REFINE = 3, CTR_COUNT = 8 };

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
    double *pu, *pv, *px, *py, *pz;   // per return, valid for the visible ones (listed in the cells)
    int* cell_pts;                    // [frame][cell][kCellCap]
    int* band_idx;                    // compacted indices of the returns inside the z band, in index order
    double *bx, *by, *bz;             // camera-frame coordinates of the band returns (same order)
    double* ref_part;                 // [frame][chunk of kRefineChunk band returns][10] partial moments of the refinement
    size_t ref_stride;
    int* band_n;                      // [frame]
    double* plane;                    // [frame][8]: n(3), d, ok, inliers, more hypotheses wanted, iteration bound
    double* red;                      // [frame][16]: refinement moments (10); [14] = index of the best hypothesis
    // zero-initialised scratch of this call / the zone this call clears for the next one
    int *zone, *zone_next;
    size_t zone_stride, zone_next_clear;  // ints per frame; ints of zone_next this call has to clear
    int off_hyp, off_ctr, off_scan;       // layout of a frame's zone: cell_count at 0, inlier counts, counters, scan words
};

struct FrameView {
    const double *pu, *pv, *px, *py, *pz;
    const int *cell_count, *cell_pts;
    const double* plane;
};
__device__ __forceinline__ FrameView frame_view(const DepthView& d, int f) {
    const size_t o = (size_t)f * d.pt_stride;
    return {d.pu + o, d.pv + o, d.px + o, d.py + o, d.pz + o, d.zone + (size_t)f * d.zone_stride,
            d.cell_pts + (size_t)f * d.n_cells * kCellCap, d.plane + 8 * (size_t)f};
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// agent-scope accesses to the words workgroups of one launch exchange (scan state, done counters, inlier counts)
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// "last workgroup of the frame".  Everything workgroups of one launch hand to each other travels through agent-scope
// atomics (inlier counts, partial sums stored / loaded with st_agent_f64 / ld_agent_f64), so the counter needs no
// release / acquire fence: an agent-scope fence per workgroup writes back and invalidates the XCD's L2 and costs more
// than the kernels' work.  After the barrier every atomic of the workgroup has been performed; the last workgroup's
// loads are issued after its increment has returned.
__device__ __forceinline__ bool last_block_done(int* counter, int total) {
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1;
    __syncthreads();
    return s_last != 0;
}
__device__ __forceinline__ void st_agent_f64(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent_f64(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

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
    if (live) q = reinterpret_cast<const float4*>(d.cloud[f])[i];  // 16-byte coalesced read
    const double x = q.x, y = q.y, z = q.z;
    const double cxp = d.R[0] * x + d.R[1] * y + d.R[2] * z + d.t[0];
    const double cyp = d.R[3] * x + d.R[4] * y + d.R[5] * z + d.t[1];
    const double czp = d.R[6] * x + d.R[7] * y + d.R[8] * z + d.t[2];
    if (live && !(d.p.do_use_cut_behind_camera && !(czp > 0.0)) && czp != 0.0) {
        const double u = d.f * cxp / czp + d.cx;
        const double v = d.f * cyp / czp + d.cy;
        if (u >= 0.0 && u < (double)d.img_w && v >= 0.0 && v < (double)d.img_h) {
            d.pu[fo + i] = u;
            d.pv[fo + i] = v;
            d.px[fo + i] = cxp;
            d.py[fo + i] = cyp;
            d.pz[fo + i] = czp;
            const int cell = ((int)v / kCell) * d.cells_x + (int)u / kCell;
            const int pos = atomicAdd(&zone[cell], 1);
            if (pos < kCellCap) d.cell_pts[((size_t)f * d.n_cells + cell) * kCellCap + pos] = i;
        }
    }
    if (!((d.ground_mask >> f) & 1u)) return;  // no ground plane wanted for this frame (uniform over the workgroup)
    // ---- D6a: returns with lidar z inside [min_z, max_z], kept in index order
    const bool flag = live && z >= d.p.ransac_plane_min_z && z <= d.p.ransac_plane_max_z;
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
                int v = lane <= first_p ? (int)(unsigned)(w & 0xffffffffull) : 0;
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                excl += v;
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
        const size_t pos = fo + off + __popcll(m & ((1ull << lane) - 1ull));
        d.band_idx[pos] = i;
        d.bx[pos] = cxp;
        d.by[pos] = cyp;
        d.bz[pos] = czp;
    }
}

// ------------------------------------------------------------------------------------------ D6a ground plane
// hypothesis `it` of a frame: plane through three seeded band returns; false for a degenerate draw
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
__device__ bool hyp_plane(const DepthView& d, size_t fo, int nb, int it, double* pl) {
    if (nb < 3) return false;
    const size_t i0 = hyp_vertex(d, nb, it, 0), i1 = hyp_vertex(d, nb, it, 1), i2 = hyp_vertex(d, nb, it, 2);
    if (i0 == i1 || i0 == i2 || i1 == i2) return false;
    const double a[3] = {d.bx[fo + i0], d.by[fo + i0], d.bz[fo + i0]}, b[3] = {d.bx[fo + i1], d.by[fo + i1], d.bz[fo + i1]},
                 c[3] = {d.bx[fo + i2], d.by[fo + i2], d.bz[fo + i2]};
    return plane_through(a, b, c, pl);
}

// Inlier counts: workgroup = (kHypPerBlock hypotheses, kRansacChunk band returns).  The returns are loaded once (4 per
// lane, issued before anything else); 192 lanes draw and gather the 3 vertices of the 64 planes, 64 lanes build the planes
// (LDS); every wave then tests its returns against the planes (ballot + popcount: wave-uniform counts, lane k keeps the
// count of plane k).  Integer atomics make the totals order-independent.  The last workgroup of the frame applies the
// sequential RANSAC semantics to the pre-computed hypotheses: keep the best so far, stop once the adaptive iteration
// bound k = log(1-p)/log(1-w^3) is reached.
//
// The sequential semantics almost always stop inside the first 64 hypotheses (inlier ratio 0.5 => bound 35), so the
// counts come in two launches: <true> = hypotheses 0..63 + the pick over them; <false> = the other hypotheses, whose
// workgroups leave at once unless the first pick asked for more (plane[6]), + the continued pick.
template <bool FIRST>
__global__ __launch_bounds__(256) void k_ransac(DepthView d) {
    const int f = blockIdx.z;
    if (!((d.ground_mask >> f) & 1u)) return;
    double* plane = d.plane + 8 * (size_t)f;
    if (!FIRST && plane[6] == 0.0) return;
    __shared__ double vtx[kHypPerBlock][3][3];
    __shared__ int vidx[kHypPerBlock][3];
    __shared__ double pl[kHypPerBlock][4];
    __shared__ int valid[kHypPerBlock], bc[kHypPerBlock];
    __shared__ int cnts[FIRST ? kHypPerBlock : kMaxHyp];
    int* zone = d.zone + (size_t)f * d.zone_stride;
    int* hyp_count = zone + d.off_hyp;
    const size_t fo = (size_t)f * d.pt_stride;
    const int n_hyp = d.n_hyp;
    const int h0 = (FIRST ? 0 : (int)blockIdx.x + 1) * kHypPerBlock;
    const int nb = d.band_n[f];
    const int q0 = blockIdx.y * kRansacChunk;
    const int n_chunk = max(1, (nb + kRansacChunk - 1) / kRansacChunk);  // the grid is sized for the whole sweep
    if ((int)blockIdx.y >= n_chunk) return;
    const int lane = threadIdx.x & 63;
    if (nb >= 3) {
        double x[4], y[4], z[4];
        bool live[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = q0 + k * 256 + (int)threadIdx.x;
            live[k] = q < nb;
            x[k] = live[k] ? d.bx[fo + q] : 0.0;
            y[k] = live[k] ? d.by[fo + q] : 0.0;
            z[k] = live[k] ? d.bz[fo + q] : 0.0;
        }
        if (threadIdx.x < 3 * kHypPerBlock) {
            const int hl = threadIdx.x / 3, v = threadIdx.x % 3;
            const size_t i = hyp_vertex(d, nb, min(h0 + hl, n_hyp - 1), v);
            vidx[hl][v] = (int)i;
            vtx[hl][v][0] = d.bx[fo + i];
            vtx[hl][v][1] = d.by[fo + i];
            vtx[hl][v][2] = d.bz[fo + i];
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
    if (!last_block_done(&zone[d.off_ctr + (FIRST ? CTR_RANSAC : CTR_RANSAC2)], gridDim.x * n_chunk)) return;
    const int it0 = FIRST ? 0 : kHypPerBlock, it1 = FIRST ? min(n_hyp, kHypPerBlock) : n_hyp;
    for (int h = it0 + threadIdx.x; h < it1; h += 256) cnts[h - it0] = __hip_atomic_load(&hyp_count[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x != 0) return;
    double* red = d.red + 16 * (size_t)f;
    int best = FIRST ? 0 : (int)plane[5], bi = FIRST ? -1 : (int)red[14];
    double k_needed = FIRST ? (double)n_hyp : plane[7];
    for (int it = it0; it < it1; ++it) {
        if (it >= k_needed) break;
        const int c = cnts[it - it0];
        if (c > best) {
            best = c;
            bi = it;
            const double w = (double)c / (double)nb;
            const double denom = log(fmax(1e-300, 1.0 - w * w * w));
            k_needed = denom < 0 ? log(1.0 - d.p.ransac_plane_probability) / denom : 0.0;
        }
    }
    double p4[4] = {0.0, 0.0, 0.0, 0.0};
    if (bi >= 0) hyp_plane(d, fo, nb, bi, p4);
    for (int k = 0; k < 4; ++k) plane[k] = p4[k];
    plane[4] = (best >= 3) ? 1.0 : 0.0;
    plane[5] = best;
    plane[6] = (FIRST && it1 < n_hyp && k_needed > (double)it1) ? 1.0 : 0.0;  // the sequential loop would go on
    plane[7] = k_needed;
    red[14] = bi;
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

// least-squares refinement of the RANSAC plane over the band returns within refinement_treshold of it: centroid c and
// scatter matrix S = sum (p-c)(p-c)^T, then the smallest eigenvector of S.  One pass: moments m0 = n, m1 = sum e,
// m2 = sum e e^T of e = p - a around a point a ON the RANSAC plane (a = -d n; the returns lie around it, so nothing
// cancels), then c = a + m1/m0 and S = m2 - m0 (m1/m0)(m1/m0)^T.  Two levels in a FIXED order (deterministic): a
// workgroup reduces a chunk of kRefineChunk returns (4 consecutive ones per lane, then a tree); the last workgroup of the
// frame adds the chunk sums in a fixed partition + tree and finishes the plane (d >= 0).
constexpr int kRefVals = 10;
__global__ __launch_bounds__(256) void k_refine(DepthView d) {
    const int f = blockIdx.y;
    if (!((d.ground_mask >> f) & 1u)) return;
    double* plane = d.plane + 8 * (size_t)f;
    if (plane[4] == 0.0) return;  // no plane: the same for every workgroup of the frame
    const double pl[4] = {plane[0], plane[1], plane[2], plane[3]};
    if (!d.p.ransac_plane_use_refinement) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && pl[3] < 0)
            for (int k = 0; k < 4; ++k) plane[k] = -pl[k];
        return;
    }
    __shared__ double sh[256];
    int* zone = d.zone + (size_t)f * d.zone_stride;
    const size_t fo = (size_t)f * d.pt_stride;
    const int nb = d.band_n[f];
    const int q0 = blockIdx.x * kRefineChunk;
    double* part = d.ref_part + (size_t)f * d.ref_stride;
    const int n_chunk = max(1, (nb + kRefineChunk - 1) / kRefineChunk);  // the grid is sized for the whole sweep
    if ((int)blockIdx.x >= n_chunk) return;
    const double a[3] = {-pl[3] * pl[0], -pl[3] * pl[1], -pl[3] * pl[2]};
    if (q0 < nb) {
        double acc[kRefVals] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < 4; ++k) {
            const int q = q0 + 4 * threadIdx.x + k;
            if (q >= nb) break;
            const double p[3] = {d.bx[fo + q], d.by[fo + q], d.bz[fo + q]};
            if (!(fabs(pl[0] * p[0] + pl[1] * p[1] + pl[2] * p[2] + pl[3]) < d.p.ransac_plane_refinement_treshold)) continue;
            const double e[3] = {p[0] - a[0], p[1] - a[1], p[2] - a[2]};
            acc[0] += 1.0;
            acc[1] += e[0];
            acc[2] += e[1];
            acc[3] += e[2];
            acc[4] += e[0] * e[0];
            acc[5] += e[0] * e[1];
            acc[6] += e[0] * e[2];
            acc[7] += e[1] * e[1];
            acc[8] += e[1] * e[2];
            acc[9] += e[2] * e[2];
        }
        for (int k = 0; k < kRefVals; ++k) {
            sh[threadIdx.x] = acc[k];
            __syncthreads();
            for (int st = 128; st > 0; st >>= 1) {
                if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
                __syncthreads();
            }
            if (threadIdx.x == 0) st_agent_f64(&part[(size_t)blockIdx.x * kRefVals + k], sh[0]);
            __syncthreads();
        }
    }
    if (!last_block_done(&zone[d.off_ctr + CTR_REFINE], n_chunk)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int v = wave; v < kRefVals; v += 4) {  // one wave per value
        double s = 0.0;
        for (int b = lane; b < n_chunk; b += 64) s += ld_agent_f64(&part[(size_t)b * kRefVals + v]);  // fixed partition ...
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);                        // ... fixed tree
        if (lane == 0) sh[v] = s;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double n[3] = {pl[0], pl[1], pl[2]}, dd = pl[3];
    const double m0 = sh[0];
    if (m0 >= 3.0) {
        const double c[3] = {sh[1] / m0, sh[2] / m0, sh[3] / m0};  // centroid - a
        const double S[6] = {sh[4] - m0 * c[0] * c[0], sh[5] - m0 * c[0] * c[1], sh[6] - m0 * c[0] * c[2],
                             sh[7] - m0 * c[1] * c[1], sh[8] - m0 * c[1] * c[2], sh[9] - m0 * c[2] * c[2]};
        smallest_eigvec(S, n);
        dd = -(n[0] * (a[0] + c[0]) + n[1] * (a[1] + c[1]) + n[2] * (a[2] + c[2]));
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
    double* red = d.red + 16 * (size_t)f;
    for (int k = 0; k < kRefVals; ++k) red[k] = sh[k];
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
    const int fr = blockIdx.y;
    const int k = blockIdx.x * 4 + wave;
    if (k >= d.n_feat[fr]) return;  // whole wave exits together
    WaveLds& L = lds[wave];
    const FrameView F = frame_view(d, fr);
    const float* feat_uv = d.feat_uv[fr];
    const uint8_t* feat_ground = d.feat_ground[fr];
    const double fu = feat_uv[2 * (size_t)k], fv = feat_uv[2 * (size_t)k + 1];
    const double hw = 0.5 * d.p.pixelarea_search_width, hh = 0.5 * d.p.pixelarea_search_height;
    const double cu = fu + d.p.pixelarea_search_offset_x, cv = fv + d.p.pixelarea_search_offset_y;
    // ---- D2: candidates from the cells under the rectangle, ballot compaction
    const int cx0 = max(0, (int)floor((cu - hw) / kCell)), cx1 = min(d.cells_x - 1, (int)floor((cu + hw) / kCell));
    const int cy0 = max(0, (int)floor((cv - hh) / kCell)), cy1 = min(d.cells_y - 1, (int)floor((cv + hh) / kCell));
    int n = 0;
    for (int cy = cy0; cy <= cy1; ++cy)
        for (int cx = cx0; cx <= cx1; ++cx) {
            const int cell = cy * d.cells_x + cx;
            const int cnt = min(kCellCap, F.cell_count[cell]);
            bool in = false;
            int idx = -1;
            if (lane < cnt) {
                idx = F.cell_pts[cell * kCellCap + lane];
                in = fabs(F.pu[idx] - cu) <= hw && fabs(F.pv[idx] - cv) <= hh;
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
        const bool ground_feat = feat_ground && feat_ground[k] && F.plane[4] != 0.0;
        double depth = -1.0, zlo = 0.0, zhi = 0.0;
        bool have = false;
        if (ground_feat) {
            // ---- D6b: inverse-distance weighted patch over the neighbours close to the sweep's ground plane (lane 0,
            //      sequential in return order: identical summation order to the oracle)
            if (lane == 0) {
                const double gn[3] = {F.plane[0], F.plane[1], F.plane[2]}, gd = F.plane[3];
                double sw = 0, c[3] = {0, 0, 0};
                int m = 0;
                zlo = 1.79769313486231570e308;
                zhi = -zlo;
                for (int q = 0; q < n; ++q) {
                    const int i = L.nb_idx[q];
                    const double p[3] = {F.px[i], F.py[i], F.pz[i]};
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
                myz = F.pz[L.nb_idx[lane]];
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
                            L.seg[pos][0] = F.px[i];
                            L.seg[pos][1] = F.py[i];
                            L.seg[pos][2] = F.pz[i];
                        }
                        nseg = __popcll(m);
                    }
                }
            } else {
                if (lane < n) {
                    const int i = L.nb_idx[lane];
                    L.seg[lane][0] = F.px[i];
                    L.seg[lane][1] = F.py[i];
                    L.seg[lane][2] = F.pz[i];
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
    int off_hyp = 0, off_ctr = 0, off_scan = 0;
    int zone_used[2] = {0, 0};       // frames of each zone a call has written into since it was last cleared
    int cur = 0;                     // zone of the next call
    float* cloud = nullptr;
    double *pu = nullptr, *pv = nullptr, *px = nullptr, *py = nullptr, *pz = nullptr;
    int *cell_pts = nullptr, *band_idx = nullptr, *band_n = nullptr, *zone[2] = {nullptr, nullptr};
    double *plane = nullptr, *red = nullptr, *bx = nullptr, *by = nullptr, *bz = nullptr, *ref_part = nullptr;
    float *feat_uv = nullptr, *out = nullptr;
    uint8_t* feat_ground = nullptr;
    float* h_feat = nullptr;   // pinned staging: uv of every frame, then the ground labels; and the depths coming back
    float* h_out = nullptr;
    void release() {
        void* ptrs[] = {cloud, pu, pv, px, py, pz, cell_pts, band_idx, band_n, zone[0], zone[1], plane, red, bx, by, bz, ref_part,
                        feat_uv, out, feat_ground};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        if (h_feat) (void)hipHostFree(h_feat);
        if (h_out) (void)hipHostFree(h_out);
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
    rc |= grow(ctx, &W.pu, (size_t)F * P);
    rc |= grow(ctx, &W.pv, (size_t)F * P);
    rc |= grow(ctx, &W.px, (size_t)F * P);
    rc |= grow(ctx, &W.py, (size_t)F * P);
    rc |= grow(ctx, &W.pz, (size_t)F * P);
    rc |= grow(ctx, &W.band_idx, (size_t)F * P);
    rc |= grow(ctx, &W.bx, (size_t)F * P);
    rc |= grow(ctx, &W.by, (size_t)F * P);
    rc |= grow(ctx, &W.bz, (size_t)F * P);
    rc |= grow(ctx, &W.ref_part, (size_t)F * (P / kRefineChunk + 1) * kRefVals);
    rc |= grow(ctx, &W.cell_pts, (size_t)F * C * kCellCap);
    rc |= grow(ctx, &W.band_n, (size_t)F);
    rc |= grow(ctx, &W.plane, (size_t)F * 8);
    rc |= grow(ctx, &W.red, (size_t)F * 16);
    rc |= grow(ctx, &W.feat_uv, (size_t)F * Q * 2);
    rc |= grow(ctx, &W.feat_ground, (size_t)F * Q);
    rc |= grow(ctx, &W.out, (size_t)F * Q);
    // zone of a frame: cell counters | inlier counts | counters | scan words (64-bit, 8-byte aligned)
    W.off_hyp = (int)round_up(C, 2);
    W.off_ctr = W.off_hyp + kMaxHyp;
    W.off_scan = W.off_ctr + CTR_COUNT;
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

// One launch group: 1..kMaxBatch frames.
int run_group(limo_ctx* ctx, int n_frames, const limo_depth_frame* frames, const double* T_cam_lidar, double f, double cx, double cy,
              int32_t img_w, int32_t img_h, const limo_depth_params& p, bool device_ptrs) {
    if (!ctx->depth_ws) {
        ctx->depth_ws = new DepthWs();
        ctx->depth_ws_free = depth_ws_free;
    }
    DepthWs& W = *static_cast<DepthWs*>(ctx->depth_ws);
    hipStream_t s = ctx->stream;
    const int cells_x = (img_w + kCell - 1) / kCell, cells_y = (img_h + kCell - 1) / kCell;
    const size_t cells = (size_t)cells_x * cells_y;
    size_t max_pts = 0, max_feat = 0;
    for (int k = 0; k < n_frames; ++k) {
        max_pts = std::max(max_pts, frames[k].n_pts);
        max_feat = std::max(max_feat, frames[k].n_feat);
    }
    if (max_pts > 0x7fffff00u || max_feat > 0x7fffff00u) return LIMO_ERR_INVALID;
    if (int rc = ensure_capacity(ctx, W, n_frames, max_pts, max_feat, cells)) return rc;

    DepthView d;
    std::memset(&d, 0, sizeof(d));
    kba::quat_R(T_cam_lidar, d.R);
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
    d.pu = W.pu;
    d.pv = W.pv;
    d.px = W.px;
    d.py = W.py;
    d.pz = W.pz;
    d.cell_pts = W.cell_pts;
    d.band_idx = W.band_idx;
    d.bx = W.bx;
    d.by = W.by;
    d.bz = W.bz;
    d.ref_part = W.ref_part;
    d.ref_stride = (W.cap_pts / kRefineChunk + 1) * kRefVals;
    d.band_n = W.band_n;
    d.plane = W.plane;
    d.red = W.red;
    const int z = W.cur;
    d.zone = W.zone[z];
    d.zone_next = W.zone[z ^ 1];
    d.zone_stride = W.zone_stride;
    d.zone_next_clear = (size_t)W.zone_used[z ^ 1] * W.zone_stride;
    d.off_hyp = W.off_hyp;
    d.off_ctr = W.off_ctr;
    d.off_scan = W.off_scan;

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
    const unsigned F = (unsigned)n_frames;
    // the projection kernel also clears the other zone: it runs even for a call without returns
    hipLaunchKernelGGL(k_project, dim3((unsigned)std::max<size_t>(1, (max_pts + 255) / 256), F), dim3(256), 0, s, d);
    if (d.ground_mask) {
        const unsigned n_chunk_r = (unsigned)((max_pts + kRansacChunk - 1) / kRansacChunk), n_chunk_f = (unsigned)((max_pts + kRefineChunk - 1) / kRefineChunk);
        // upper bounds: the band sizes are only known on the device
        const unsigned n_groups = (unsigned)((d.n_hyp + kHypPerBlock - 1) / kHypPerBlock);
        hipLaunchKernelGGL(k_ransac<true>, dim3(1, n_chunk_r, F), dim3(256), 0, s, d);
        if (n_groups > 1) hipLaunchKernelGGL(k_ransac<false>, dim3(n_groups - 1, n_chunk_r, F), dim3(256), 0, s, d);
        hipLaunchKernelGGL(k_refine, dim3(n_chunk_f, F), dim3(256), 0, s, d);
    }
    if (max_feat) {
        hipLaunchKernelGGL(k_features, dim3((unsigned)((max_feat + 3) / 4), F), dim3(256), 0, s, d);
        if (!device_ptrs) HIP_TRY(ctx, hipMemcpyAsync(W.h_out, W.out, sizeof(float) * Q * n_frames, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (!device_ptrs)
        for (int k = 0; k < n_frames; ++k)
            if (frames[k].n_feat) std::memcpy(frames[k].depth_out, W.h_out + (size_t)k * Q, sizeof(float) * frames[k].n_feat);
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
