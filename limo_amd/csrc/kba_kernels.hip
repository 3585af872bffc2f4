// kba_kernels.hip — gfx950 kernels of the batched keyframe-BA pipeline (CDNA4, wave64).
//
// Kernel          lanes                      bound   what it replaces (reference / Ceres)
// k_linearize     1 per observation          HBM     AutoDiff evaluation of ReprojectionErrorWithQuaternions +
//                                                    LandmarkDepthError incl. loss corrector and local
//                                                    parameterisation (cost_functors_ceres.hpp:53-222,
//                                                    bundle_adjuster_keyframes.cpp:584-620) + F^T F / F^T r block sums
// k_cost          1 per observation          HBM     Evaluator::Evaluate(cost only) at the candidate point
// k_gp            1 per ground-plane row     -       GroundPlaneHeightRegularization (cost_functors_ceres.hpp:355-392)
// k_lm_accum      1 per landmark             HBM     E^T E, E^T r (SchurEliminator chunk), Jacobi column scale
// k_lm_damp       1 per landmark             HBM     (E^T E + D^2) Cholesky inverse per landmark
// k_schur         workgroup per 256 lm       MFMA    S -= sum_i Y'_i Y'_i^T   (v_mfma_f64_16x16x4_f64 SYRK from LDS tiles)
// k_cam_assemble  workgroup per window       -       camera-camera blocks, regularisers, IterationZero / step tail
// k_cam_solve     workgroup per window       -       reduced camera system: dense Cholesky in LDS, camera step
// k_backsub       1 per landmark             HBM     BackSubstitute + candidate point + model-cost-change parts
// k_step_decide   1 per window               -       TrustRegionMinimizer step acceptance (kba_lm.hpp)
// k_trim_*        1 per obs / lm / window    HBM     robust_optimization::solveTrimmed residual evaluation + quantile
//
// Every workgroup first looks at its window's LM state and returns if the window is not iterating, so one launch
// sequence serves a whole batch of windows that converge at different iterations.
#include <hip/hip_runtime.h>

#define KBA_SYNC() __syncthreads()
#include "kba_items.hpp"

namespace kba {

// ------------------------------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// Sum N per-lane values over the 256-lane workgroup; result valid in lanes < N of wave 0 as return of lane i.
template <int N>
__device__ __forceinline__ void block_sum(const double* vals, double* lds /* [4*N] */, double* out_global) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double s = wave_sum(vals[i]);
        if (lane == 0) lds[wave * N + i] = s;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        out_global[threadIdx.x] = (lds[threadIdx.x] + lds[N + threadIdx.x]) + (lds[2 * N + threadIdx.x] + lds[3 * N + threadIdx.x]);
    }
}

// ------------------------------------------------------------------------------------------ LM control
__global__ void k_solve_init(BatchView bv, SolveConsts c, int max_iter, int select) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= bv.n_win) return;
    WinState& s = bv.st[w];
    bool sel = true;
    if (select >= 1) sel = bv.win[w].do_trim != 0;
    if (select == 2) sel = sel && (s.solve_initial_cost - s.solve_final_cost <= 0.0);
    lm_solve_init(s, sel, max_iter, c);
}

// activity flags for host-side re-batching (worklists of the windows that still iterate)
__global__ void k_export_active(BatchView bv, int32_t* flags) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < bv.n_win) flags[w] = bv.st[w].active;
}

__global__ void k_expire(BatchView bv) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= bv.n_win) return;
    if (bv.st[w].active) lm_terminate(bv.st[w], LIMO_NO_CONVERGENCE);
}

// ------------------------------------------------------------------------------------------ observations
__global__ __launch_bounds__(kBlock) void k_linearize(BatchView bv, SolveConsts c, const int32_t* wl) {
    const int b = wl ? wl[blockIdx.x] : blockIdx.x;
    const int w = bv.view_win[bv.blk_view[b]];
    const WinState& st = bv.st[w];
    if (!st.active || !st.need_lin) return;
    __shared__ double lds[4 * kLinPartial];
    LinLane l;
    linearize_lane(bv, c, b, threadIdx.x, l);
    double vals[kLinPartial];
    vals[0] = l.cost;
#pragma unroll
    for (int i = 0; i < 21; ++i) vals[1 + i] = l.U[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) vals[22 + i] = l.g[i];
    const int any_fail = __syncthreads_or(l.fail);
    if (c.pad == 21) {  // profiling aid: skip the workgroup reduction
        if (threadIdx.x < kLinPartial) bv.blk_part[(int64_t)b * kLinPartial + threadIdx.x] = vals[0];
    } else {
        block_sum<kLinPartial>(vals, lds, bv.blk_part + (int64_t)b * kLinPartial);
    }
    if (threadIdx.x == 0) bv.blk_fail[b] = any_fail;
}

__global__ __launch_bounds__(kBlock) void k_cost(BatchView bv, SolveConsts c, const int32_t* wl) {
    const int b = wl ? wl[blockIdx.x] : blockIdx.x;
    const int w = bv.view_win[bv.blk_view[b]];
    if (!bv.st[w].active) return;
    __shared__ double lds[4];
    double cost;
    int fail;
    cost_lane(bv, c, b, threadIdx.x, cost, fail);
    const int any_fail = __syncthreads_or(fail);
    block_sum<1>(&cost, lds, bv.blk_cost_c + b);
    if (threadIdx.x == 0) bv.blk_fail_c[b] = any_fail;
}

__global__ void k_gp(BatchView bv, int candidate) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= bv.TG) return;
    const int w = bv.lm_win[bv.gp_lm[g]];
    const WinState& st = bv.st[w];
    if (!st.active) return;
    if (!candidate && !st.need_lin) return;
    gp_lane(bv, g, candidate != 0, candidate ? bv.gp_cost_c : bv.gp_cost);
}

// ------------------------------------------------------------------------------------------ landmarks
__global__ __launch_bounds__(kBlock) void k_lm_accum(BatchView bv, SolveConsts c, const int32_t* wl) {
    const int b = wl ? wl[blockIdx.x] : blockIdx.x;
    const int w = bv.lblk_win[b];
    const WinState& st = bv.st[w];
    if (!st.active || !st.need_lin) return;
    __shared__ double lds[8];
    double part[2] = {0.0, 0.0};
    if ((int)threadIdx.x < bv.lblk_n[b]) lm_accum_lane(bv, c, bv.lblk_lm0[b] + threadIdx.x, part);
    const double m = wave_max(part[0]);
    const double s = wave_sum(part[1]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        lds[wave] = m;
        lds[4 + wave] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bv.lblk_part[(int64_t)b * 8 + 0] = fmax(fmax(lds[0], lds[1]), fmax(lds[2], lds[3]));
        bv.lblk_part[(int64_t)b * 8 + 1] = (lds[4] + lds[5]) + (lds[6] + lds[7]);
    }
}

__global__ __launch_bounds__(kBlock) void k_lm_damp(BatchView bv, SolveConsts c, const int32_t* wl) {
    const int b = wl ? wl[blockIdx.x] : blockIdx.x;
    const int w = bv.lblk_win[b];
    if (!bv.st[w].active) return;
    int fail = 0;
    if ((int)threadIdx.x < bv.lblk_n[b]) fail = lm_damp_lane(bv, c, bv.lblk_lm0[b] + threadIdx.x);
    const int any = __syncthreads_or(fail);
    if (threadIdx.x == 0) bv.lblk_part[(int64_t)b * 8 + 5] = any ? 1.0 : 0.0;
}

__global__ __launch_bounds__(kBlock) void k_backsub(BatchView bv, const int32_t* wl) {
    const int b = wl ? wl[blockIdx.x] : blockIdx.x;
    const int w = bv.lblk_win[b];
    if (!bv.st[w].active) return;
    __shared__ double lds[12];
    double part[8];
    part[2] = part[3] = part[4] = 0.0;
    if ((int)threadIdx.x < bv.lblk_n[b]) backsub_lane(bv, bv.lblk_lm0[b] + threadIdx.x, part);
    block_sum<3>(part + 2, lds, bv.lblk_part + (int64_t)b * 8 + 2);
}

// ------------------------------------------------------------------------------------------ Schur complement (MFMA)
typedef double v4f64 __attribute__((ext_vector_type(4)));

// LDS row stride (doubles) of the Schur tile.  ODD on purpose: the fill phase has the 32 lanes of a group write rows
// k = 3*li + c (stride 3*ld doubles): with ld = 48 that is 288 dwords == 0 (mod 32 banks), a 32-way conflict on every
// store; with ld = 49 the lanes land on distinct banks.  The MFMA operand reads (lanes along a row, 4 k-rows per
// instruction) then overlap on two banks only.
__host__ __device__ inline int schur_ld(int ncp) {
    return ncp + 1;
}

constexpr int kSchurMaxTilesPerWave = 9;  // upper-triangular 16x16 tiles of a 128x128 system over 4 waves
constexpr int kSchurMaxViews = 64;

__host__ __device__ inline int schur_lds_doubles(int nfp) {
    return 3 * kSchurLm * schur_ld(nfp) + 3 * kSchurLm + kMaxNc + (kMaxNc + kSchurMaxViews + 1) / 2;
}

__global__ __launch_bounds__(kBlock) void k_schur(BatchView bv, const int32_t* wl, int dbg) {
    const int sb = wl ? wl[blockIdx.x] : blockIdx.x;
    const int w = bv.sblk_win[sb];
    if (!bv.st[w].active) return;
    const WinDesc& wd = bv.win[w];
    const int nc = wd.nc, nf = wd.nf, nfp = wd.nf_pad, ld = schur_ld(nfp);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Z = smem;                        // [3*kSchurLm][ld]
    double* tt = Z + 3 * kSchurLm * ld;      // [3*kSchurLm]
    double* sc_s = tt + 3 * kSchurLm;        // [kMaxNc] camera column scale by full local slot
    int* cs_s = reinterpret_cast<int*>(sc_s + kMaxNc);  // [kMaxNc] compact slot or -1
    int* vkl = cs_s + kMaxNc;                // [kSchurMaxViews] keyframe (local) of each view
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < nc; i += kBlock) {
        sc_s[i] = bv.scale_c[wd.cam0 + i];
        cs_s[i] = bv.cslot[wd.cam0 + i];
    }
    for (int j = threadIdx.x; j < wd.n_view; j += kBlock) vkl[j] = bv.view_kf[wd.view0 + j] - wd.kf0;
    const int T = nfp / 16;
    const int n_upper = T * (T + 1) / 2;
    v4f64 acc[kSchurMaxTilesPerWave];
#pragma unroll
    for (int i = 0; i < kSchurMaxTilesPerWave; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
    double rhs_acc = 0.0;
    const int li = threadIdx.x & (kSchurLm - 1);
    const int grp = threadIdx.x / kSchurLm;  // 0..7 : owns keyframes kf_local == grp (mod 8)
    const int n_lm_blk = bv.sblk_n[sb];
    for (int l0 = 0; l0 < n_lm_blk; l0 += kSchurLm) {
        const int nl = min(kSchurLm, n_lm_blk - l0);
        if (dbg != 34) for (int i = threadIdx.x; i < 3 * kSchurLm * ld; i += kBlock) Z[i] = 0.0;
        if (threadIdx.x < 3 * kSchurLm) tt[threadIdx.x] = 0.0;
        __syncthreads();
        if (li < nl && dbg != 31) {
            const int gl = bv.sblk_lm0[sb] + l0 + li;
            if (bv.lm_state[gl] == 1) {
                const int gg = bv.lm_gp[gl];
                const int gkl = gg >= 0 ? bv.gp_kf[gg] - wd.kf0 : -1;
                bool mine = (gkl >= 0 && (gkl & 7) == grp) || grp == 0;
                for (int j = 0; j < wd.n_view && !mine; ++j) mine = (vkl[j] & 7) == grp;
                if (mine) {
                    double lmk[9];
                    schur_load_lm(bv, gl, lmk);
                    for (int j = 0; j < wd.n_view; ++j) {
                        const int kl = vkl[j];
                        if ((kl & 7) != grp) continue;
                        schur_fill_view(bv, gl, li, j, kl, lmk, cs_s, sc_s, Z, ld);
                    }
                    if (gkl >= 0 && (gkl & 7) == grp) schur_fill_gp(bv, gg, li, gkl, lmk, cs_s, sc_s, Z, ld);
                    if (grp == 0) {
                        tt[3 * li + 0] = bv.lm_t[0 * bv.SL + gl];
                        tt[3 * li + 1] = bv.lm_t[1 * bv.SL + gl];
                        tt[3 * li + 2] = bv.lm_t[2 * bv.SL + gl];
                    }
                }
            }
        }
        __syncthreads();
        // SYRK on the tile: D[tr][tc] += sum_k Z[k][tr*16+i] * Z[k][tc*16+j]
        const int ksteps = (3 * nl + 3) / 4;
#pragma unroll
        for (int q = 0; q < kSchurMaxTilesPerWave; ++q) {
            const int tile = wave + 4 * q;
            if (tile < n_upper && dbg != 32) {
                int tr = 0, rem = tile;  // upper-triangular tile index -> (tr, tc)
                while (rem >= T - tr) {
                    rem -= T - tr;
                    ++tr;
                }
                const int tc = tr + rem;
                const double* za = Z + (lane >> 4) * ld + tr * 16 + (lane & 15);
                const double* zb = Z + (lane >> 4) * ld + tc * 16 + (lane & 15);
                v4f64 a4 = acc[q];
                for (int ks = 0; ks < ksteps; ++ks) {
                    const double a = za[ks * 4 * ld];
                    const double bb = zb[ks * 4 * ld];
                    a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, a4, 0, 0, 0);
                }
                acc[q] = a4;
            }
        }
        if ((int)threadIdx.x < nf && dbg != 33) {
            double s = 0.0;
            for (int k = 0; k < 3 * nl; ++k) s += Z[k * ld + threadIdx.x] * tt[k];
            rhs_acc += s;
        }
        __syncthreads();
    }
    const int slab = nfp * nfp + nfp;
    double* out = bv.S_part + wd.spart_off + (int64_t)(sb - wd.sblk0) * slab;
#pragma unroll
    for (int q = 0; q < kSchurMaxTilesPerWave; ++q) {
        const int tile = wave + 4 * q;
        if (tile < n_upper) {
            int tr = 0, rem = tile;
            while (rem >= T - tr) {
                rem -= T - tr;
                ++tr;
            }
            const int tc = tr + rem;
            const int col = tc * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = tr * 16 + (lane >> 4) + 4 * r;  // f64 16x16x4 C/D layout: row = (lane>>4) + 4*reg
                const double v = acc[q][r];
                out[row * nfp + col] = v;
                if (tr != tc) out[col * nfp + row] = v;
            }
        }
    }
    if ((int)threadIdx.x < nfp) out[nfp * nfp + threadIdx.x] = ((int)threadIdx.x < nf) ? rhs_acc : 0.0;
}

// ------------------------------------------------------------------------------------------ camera system
__global__ __launch_bounds__(kBlock) void k_cam_assemble(BatchView bv, SolveConsts c, const int32_t* wl) {
    const int w = wl ? wl[blockIdx.x] : blockIdx.x;
    WinState& st = bv.st[w];
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (st.active && st.need_lin) {
        cam_assemble(bv, c, w, threadIdx.x, blockDim.x, smem);
        __syncthreads();
        if (threadIdx.x == 0) lm_decide_lin(st, bv.red[w], bv.reg_cost[2 * w + 1], c);
    }
    __syncthreads();
    if (threadIdx.x == 0 && st.active) atomicAdd(bv.n_active, 1);
}

__global__ __launch_bounds__(kBlock) void k_cam_solve(BatchView bv, SolveConsts c, const int32_t* wl) {
    const int w = wl ? wl[blockIdx.x] : blockIdx.x;
    if (!bv.st[w].active) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int flag;
    cam_solve(bv, c, w, threadIdx.x, blockDim.x, smem, &flag);
}

__global__ __launch_bounds__(64) void k_step_decide(BatchView bv, SolveConsts c, const int32_t* wl) {
    const int w = wl ? wl[blockIdx.x] : blockIdx.x;
    if (!bv.st[w].active) return;
    __shared__ double red[64];
    reduce_step(bv, w, threadIdx.x, blockDim.x, red);
    __syncthreads();
    if (threadIdx.x == 0) lm_decide_step(bv.st[w], bv.red[w], c);
}

// candidate -> current for accepted windows (keyframe part: first TK threads, landmark part: the rest)
__global__ void k_accept(BatchView bv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bv.TK) {
        if (!bv.st[bv.kf_win[i]].accept) return;
        for (int q = 0; q < 7; ++q) bv.pose[7 * (int64_t)i + q] = bv.pose_c[7 * (int64_t)i + q];
        for (int q = 0; q < 3; ++q) bv.pdir[3 * (int64_t)i + q] = bv.pdir_c[3 * (int64_t)i + q];
        bv.pdist[i] = bv.pdist_c[i];
        return;
    }
    const int l = i - bv.TK;
    if (l >= bv.TL) return;
    if (!bv.st[bv.lm_win[l]].accept) return;
    for (int q = 0; q < 3; ++q) bv.lm[3 * (int64_t)l + q] = bv.lm_c[3 * (int64_t)l + q];
}

// ------------------------------------------------------------------------------------------ trimming
__global__ __launch_bounds__(kBlock) void k_trim_residual(BatchView bv, double* plane_rep, double* plane_dep) {
    const int b = blockIdx.x;
    const int w = bv.view_win[bv.blk_view[b]];
    if (!bv.win[w].do_trim) return;
    trim_residual_lane(bv, b, threadIdx.x, plane_rep, plane_dep);
}

__global__ void k_trim_max(BatchView bv, const double* plane_rep, const double* plane_dep) {
    const int gl = blockIdx.x * blockDim.x + threadIdx.x;
    if (gl >= bv.TL) return;
    if (!bv.win[bv.lm_win[gl]].do_trim) return;
    trim_max_lane(bv, gl, plane_rep, plane_dep);
}

// Quantile selection per window and list: bitonic sort of (value, id) in LDS, outliers = sorted positions
// >= int(n_groups * quantile) (TrimmerQuantile::getOutliers; std::nth_element ties resolved by id).  Falls back to
// the O(n^2) rank count (same result) when the padded list does not fit in LDS.
constexpr int kTrimMaxSort = 8192;

__device__ __forceinline__ bool trim_less(double ka, int ia, double kb, int ib) {
    return ka < kb || (ka == kb && ia < ib);
}

__global__ __launch_bounds__(kBlock) void k_trim_select(BatchView bv, SolveConsts c) {
    const int w = blockIdx.x;
    const WinDesc& wd = bv.win[w];
    if (!wd.do_trim) return;
    const int n = wd.n_lm;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int n_valid;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    int removed = 0;
    if (np2 <= kTrimMaxSort) {
        double* keys = smem;                                   // [np2]
        int* ids = reinterpret_cast<int*>(smem + np2);          // [np2]
        unsigned char* flags = reinterpret_cast<unsigned char*>(ids + np2);  // [n]
        for (int i = threadIdx.x; i < n; i += kBlock) flags[i] = 0;
        for (int list = 0; list < 2; ++list) {
            const double* vals = (list == 0 ? bv.trim_dep : bv.trim_rep) + wd.lm0;
            const double q = list == 0 ? c.depth_quantile : c.reprojection_quantile;
            if (threadIdx.x == 0) n_valid = 0;
            __syncthreads();
            int mine = 0;
            for (int i = threadIdx.x; i < np2; i += kBlock) {
                double v = i < n ? vals[i] : -1.0;
                const bool valid = v >= 0.0;  // also true for +inf (failed functor)
                keys[i] = valid ? v : INFINITY;
                ids[i] = valid ? i : (i | (1 << 30));  // invalid entries sort after every valid one
                mine += valid;
            }
            if (mine) atomicAdd(&n_valid, mine);
            __syncthreads();
            for (int k = 2; k <= np2; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = threadIdx.x; t < np2; t += kBlock) {
                        const int p = t ^ j;
                        if (p > t) {
                            const double ka = keys[t], kb = keys[p];
                            const int ia = ids[t], ib = ids[p];
                            const bool up = (t & k) == 0;
                            const bool swap = up ? trim_less(kb, ib, ka, ia) : trim_less(ka, ia, kb, ib);
                            if (swap) {
                                keys[t] = kb;
                                keys[p] = ka;
                                ids[t] = ib;
                                ids[p] = ia;
                            }
                        }
                    }
                    __syncthreads();
                }
            }
            const int ng = n_valid;
            if (ng >= c.min_groups) {
                const int num = (int)((double)ng * q);
                for (int p = num + threadIdx.x; p < ng; p += kBlock) flags[ids[p]] = 1;
            }
            __syncthreads();
        }
        for (int l = threadIdx.x; l < n; l += kBlock) {
            if (flags[l] && bv.lm_state[wd.lm0 + l]) {
                bv.lm_state[wd.lm0 + l] = 0;
                ++removed;
            }
        }
    } else {
        for (int l = threadIdx.x; l < n; l += kBlock) {
            const int out = trim_is_outlier(bv.trim_dep + wd.lm0, n, l, c.depth_quantile, c.min_groups) ||
                            trim_is_outlier(bv.trim_rep + wd.lm0, n, l, c.reprojection_quantile, c.min_groups);
            if (out && bv.lm_state[wd.lm0 + l]) {
                bv.lm_state[wd.lm0 + l] = 0;
                ++removed;
            }
        }
    }
    if (removed) atomicAdd(&bv.st[w].n_trimmed, removed);
}

// ------------------------------------------------------------------------------------------ evaluate (Problem::Evaluate)
// Writes per-observation residuals / Jacobians into the planes and per-observation cost / valid flags.
__global__ __launch_bounds__(kBlock) void k_evaluate(BatchView bv, SolveConsts c, int apply_loss, double* obs_cost,
                                                     uint8_t* obs_valid) {
    const int b = blockIdx.x;
    const int t = threadIdx.x;
    if (t >= bv.blk_n[b]) return;
    const int view = bv.blk_view[b];
    const int64_t o = bv.blk_obs0[b] + t;
    const int gl = bv.obs_lm[o];
    const double* cam = bv.view_cam + 16 * (int64_t)view;
    ObsOut oo;
    const bool ok = obs_residual_jacobian(bv.pose + 7 * (int64_t)bv.view_kf[view], cam + 4, cam + 13, cam[0], cam[1], cam[2],
                                          bv.lm + 3 * (int64_t)gl, bv.obs_u[o], bv.obs_v[o], bv.obs_d[o], bv.lm_weight[gl],
                                          c.a_rep, c.a_dep, apply_loss != 0, &oo);
    if (!ok) {
        for (int i = 0; i < 3; ++i) oo.r[i] = 0.0;
        for (int i = 0; i < 18; ++i) oo.Jp[i] = 0.0;
        for (int i = 0; i < 9; ++i) oo.Jl[i] = 0.0;
        oo.cost = 0.0;
    }
    for (int i = 0; i < 3; ++i) bv.obs_r[i * bv.SO + o] = oo.r[i];
    for (int i = 0; i < 18; ++i) bv.obs_Jp[i * bv.SO + o] = oo.Jp[i];
    for (int i = 0; i < 9; ++i) bv.obs_Jl[i * bv.SO + o] = oo.Jl[i];
    obs_cost[o] = oo.cost;
    obs_valid[o] = ok ? 1 : 0;
}

}  // namespace kba
