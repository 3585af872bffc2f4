// kba_kernels.hip — gfx950 kernels of the batched keyframe-BA pipeline (CDNA4, wave64).
//
// Kernel          lanes                      bound   what it replaces (reference / Ceres)
// k_lin_lm        1 per landmark, loop views fp64    AutoDiff evaluation of ReprojectionErrorWithQuaternions + LandmarkDepthError
//                                            VALU    incl. loss corrector and local parameterisation (cost_functors_ceres.hpp:
//                                                    53-222, bundle_adjuster_keyframes.cpp:584-620): factored Jacobian planes,
//                                                    F^T F / F^T r camera sums; the landmark's ground-plane row
//                                                    (GroundPlaneHeightRegularization, cost_functors_ceres.hpp:355-392); E^T E,
//                                                    E^T r (SchurEliminator chunk), Jacobi column scale, damped 3x3 Cholesky
// k_lm_damp /     1 per landmark             HBM     (E^T E + D^2) Cholesky inverse per landmark after a REJECTED step (k_after_step: the streaming
// k_after_step                                       solve's form, at the end of the round)
// k_schur_lean/_wide  wave / 512 lanes       MFMA    S -= sum_i Y'_i Y'_i^T   (v_mfma_f64_16x16x4_f64 SYRK from LDS tiles)
// k_cam_assemble  workgroup per window       -       camera-camera blocks, regularisers, IterationZero / step tail
// k_cam_solve     workgroup per window       -       reduced camera system: dense Cholesky in LDS, camera step
// k_backsub       1 per landmark             HBM     BackSubstitute + candidate point + model-cost-change parts +
//                                                    Evaluator::Evaluate(cost only) of its observations and of its
//                                                    ground-plane row at the candidate
// k_step_decide   1 per window               -       TrustRegionMinimizer step acceptance (kba_lm.hpp)
// k_trim_*        1 per obs / lm / window    HBM     robust_optimization::solveTrimmed residual evaluation + quantile
// k_evaluate      1 per obs, wave = 64       HBM     Problem::Evaluate: residuals, J_pose, J_point of every observation written out (the
//                 aligned observations       stores  MATERIALISED pass SURVEY 8d grades); k_eval_rows: the ground-plane / regulariser rows
//
// Every workgroup first looks at its window's LM state and returns if the window is not iterating, so one launch
// sequence serves a whole batch of windows that converge at different iterations.
#include <hip/hip_runtime.h>

#include <type_traits>

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

// ---- 28 per-lane values summed over the 256-lane workgroup (k_linearize: cost | U 21 | g 6).
// Reduce-scatter instead of 28 butterflies: every exchange step halves the number of values a lane carries, so the
// wave reduction costs 28/2 + 14/2 + 4 + 2 + 1 + 1 = 29 exchange-adds instead of 28 * 6 = 168 (the butterflies via
// ds_bpermute were a quarter of the kernel).  Strides 32 / 16 use gfx950's v_permlane32_swap / v_permlane16_swap (one
// instruction moves the two halves both ways), strides 8 / 4 / 2 / 1 DPP row operations.  Fixed order: deterministic.
__device__ __forceinline__ double dpp_mov(double x, int ctrl_sel) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    switch (ctrl_sel) {  // compile-time constant after inlining
        case 0:  // row_ror:8  (lane ^ 8 within a row of 16)
            lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, false);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, false);
            break;
        case 1:  // row_half_mirror (lane -> 7 - lane within 8: the partner differs in bit 2)
            lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, false);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, false);
            break;
        case 2:  // quad_perm [2,3,0,1]  (lane ^ 2)
            lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, false);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, false);
            break;
        default:  // quad_perm [1,0,3,2]  (lane ^ 1)
            lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false);
            break;
    }
    return __hiloint2double(hi, lo);
}
// a + (a's other half), b likewise: lanes 0-31 end with sum pairs of a, lanes 32-63 with sum pairs of b
__device__ __forceinline__ double swap32_add(double a, double b) {
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// even rows (of 16 lanes) end with pair sums of a, odd rows with pair sums of b
__device__ __forceinline__ double swap16_add(double a, double b) {
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// lanes whose `bit` is clear keep a (and receive the partner's a), the others keep b
template <int CTRL>
__device__ __forceinline__ double halve_dpp(double a, double b, bool bit) {
    const double keep = bit ? b : a, send = bit ? a : b;
    return keep + dpp_mov(send, CTRL);
}
// one value summed over the wave with the same exchanges (every lane ends with the total): 6 exchange-adds, no LDS
__device__ __forceinline__ double wave_sum_all(double v) {
    v = swap32_add(v, v);
    v = swap16_add(v, v);
    v += dpp_mov(v, 0);
    v += dpp_mov(v, 1);
    v += dpp_mov(v, 2);
    return v + dpp_mov(v, 3);
}
// Index of the value whose wave total lane `lane` holds after wave_reduce_scatter28 (even lanes; -1: duplicate).
__device__ __forceinline__ int rs28_index(int lane) {
    const int row = lane >> 4, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1, b1 = (lane >> 1) & 1;
    int j;
    if (!b1 && !b2) j = b3 ? 4 : 0;
    else if (!b1 && b2) j = b3 ? 6 : 2;
    else if (b1 && !b2) j = b3 ? 5 : 1;
    else j = b3 ? -1 : 3;
    return (lane & 1) || j < 0 ? -1 : 7 * row + j;
}
__device__ __forceinline__ double wave_reduce_scatter28(const double* v, int lane) {
    double s[14], u[7];
#pragma unroll
    for (int i = 0; i < 14; ++i) s[i] = swap32_add(v[i], v[i + 14]);  // lanes < 32: values 0..13, lanes >= 32: 14..27
#pragma unroll
    for (int i = 0; i < 7; ++i) u[i] = swap16_add(s[i], s[i + 7]);     // row r of 16 lanes: values 7 r + i
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
    const double w0 = halve_dpp<0>(u[0], u[4], b3), w1 = halve_dpp<0>(u[1], u[5], b3), w2 = halve_dpp<0>(u[2], u[6], b3);
    const double w3 = u[3] + dpp_mov(u[3], 0);
    const double x0 = halve_dpp<1>(w0, w2, b2), x1 = halve_dpp<1>(w1, w3, b2);
    const double y = halve_dpp<2>(x0, x1, b1);
    return y + dpp_mov(y, 3);
}
// The same for 14 values (7 + 4 + 2 + 1 + 1 + 1 = 16 exchange-adds).  k_lin_lm reduces the 28 camera-side sums of a view as two
// halves: with all 28 (56 registers) live next to the Jacobian the kernel needed 167 registers - three waves per SIMD; in halves it
// fits into 128 - four.  After the call the lane with rs14_index(lane) = i >= 0 holds the wave total of v[i].
__device__ __forceinline__ int rs14_index(int lane) {
    const int half = lane >> 5, rowp = (lane >> 4) & 1, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1;
    const int j = b2 ? (b3 ? 3 : 1) : (b3 ? 2 : 0);
    if ((lane & 3) || (j == 3 && rowp)) return -1;
    return 7 * half + (j == 3 ? 3 : j + 4 * rowp);
}
__device__ __forceinline__ double wave_reduce_scatter14(const double* v, int lane) {
    double s[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) s[i] = swap32_add(v[i], v[i + 7]);  // lanes < 32: values 0..6, lanes >= 32: 7..13
    const double u0 = swap16_add(s[0], s[4]), u1 = swap16_add(s[1], s[5]), u2 = swap16_add(s[2], s[6]);  // even rows i, odd rows i + 4
    const double u3 = swap16_add(s[3], s[3]);                                                           // both rows: 3
    const bool b3 = lane & 8, b2 = lane & 4;
    const double w0 = halve_dpp<0>(u0, u2, b3), w1 = halve_dpp<0>(u1, u3, b3);
    const double x = halve_dpp<1>(w0, w1, b2);
    const double y = x + dpp_mov(x, 2);
    return y + dpp_mov(y, 3);
}
// out_global[i] = sum over the workgroup of vals[i], i < 28;  lds: 4 * 28 doubles
__device__ __forceinline__ void block_sum28(const double* vals, double* lds, double* out_global) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double tot = wave_reduce_scatter28(vals, lane);
    const int idx = rs28_index(lane);
    if (idx >= 0) lds[wave * 28 + idx] = tot;
    __syncthreads();
    if (threadIdx.x < 28) out_global[threadIdx.x] = (lds[threadIdx.x] + lds[28 + threadIdx.x]) + (lds[56 + threadIdx.x] + lds[84 + threadIdx.x]);
}

// Worklist entry i: plain lists (lock-step solve; null = the identity) or counted lists rebuilt by k_sched every round
// (streaming solve: the launch grid is the list's capacity, entry [-1] its length).  -1 = nothing to do.
__device__ __forceinline__ int wl_at(const BatchView& bv, const int32_t* wl, int i) {
    if (bv.counted) return i < wl[-1] ? wl[i] : -1;
    return wl ? wl[i] : i;
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

// ------------------------------------------------------------------------------------------ streaming solve: scheduler
// At the start of every round (<= 4096 windows in flight):
//   1. a window whose solve ended moves on in its schedule (kba_lm.hpp:sched_advance); finished windows leave their
//      slot and the next pending window of the batch moves in - the windows of a batch converge after 5 .. 100
//      iterations, so in a lock-step solve most launch rounds work on a fraction of the batch;
//   2. the worklists of the round are rebuilt with a workgroup scan: observation / landmark / Schur workgroups and
//      windows that iterate, and - separately - the ones whose trimming solve just ended: those are trimmed on a side
//      stream during this round (k_trim_select is a latency-bound sort) and iterate again from the next round on.  Kernels are launched over the lists' capacities and return
//      where blockIdx is past the count (wl_at).
// Which slot a window lands in does not influence any result: every per-window quantity lives at the window's own
// offsets.
// Three small kernels (one serial workgroup doing all of it took 0.2 ms per round at 4096 slots):
//   k_sched_advance  one lane per slot: phase machine, slot refill, the slot's nine list counts
//   k_sched_scan     one workgroup: exclusive scan of the counts over the slots -> offsets, list lengths, progress word
//   k_sched_fill     one wave per slot: writes the slot's entries at its offsets
constexpr int kSchedThreads = 1024;
constexpr int kSchedSlotsPerLane = 4;
constexpr int kSchedMaxSlots = kSchedThreads * kSchedSlotsPerLane;

// bv.slot_cnt: [n_slots][SL_COUNT + 1] - counts of the slot's window per list, [SL_COUNT] = kind (0 nothing, 1 iterates,
// 2 is trimmed this round); overwritten with the offsets by k_sched_scan.
__global__ void k_sched_advance(BatchView bv, SolveConsts c) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= bv.n_slots) return;
    int32_t* cnt = bv.slot_cnt + (int64_t)s * (SL_COUNT + 1);
#pragma unroll
    for (int k = 0; k <= SL_COUNT; ++k) cnt[k] = 0;
    // pending windows left?  (plain read first: once the batch is handed out, thousands of empty slots would otherwise
    // hammer the cursor with atomics every round)
    const bool pending = *(volatile int32_t*)bv.sched_ctl < bv.n_win;
    int w = bv.slot_win[s];
    const int w_in = w;
    int kind = 0, n_fin = 0;
    for (int tries = 0; tries < 2; ++tries) {
        if (w < 0) {  // free slot: next pending window of the batch
            if (!pending) break;
            const int nxt = atomicAdd(bv.sched_ctl, 1);
            if (nxt >= bv.n_win) break;
            w = nxt;
            bv.st[w].phase = PH_IDLE;
        }
        const int r = sched_advance(bv.st[w], bv.win[w], c);
        if (r == 2) {
            ++n_fin;
            w = -1;
            continue;  // the slot is free again: refill it in this round
        }
        kind = r == 1 ? (bv.st[w].phase == PH_TRIM ? 2 : 1) : 0;
        break;
    }
    if (n_fin) atomicAdd(bv.sched_ctl + 1, n_fin);
    if (w != w_in) bv.slot_win[s] = w;
    if (w < 0 || !kind) return;
    const WinDesc& wd = bv.win[w];
    cnt[SL_COUNT] = kind;
    if (kind == 2) {
        cnt[SL_TBLK] = wd.n_blk;
        cnt[SL_TLBLK] = wd.n_lblk;
        cnt[SL_TWIN] = 1;
        return;
    }
    const int gp_groups = (wd.n_sblk - wd.n_sblk_plain + c.schur_span_gp - 1) / c.schur_span_gp;
    const int pl_groups = (wd.n_sblk_plain + c.schur_span - 1) / c.schur_span;
    cnt[SL_LBLK] = wd.n_lblk;
    if (wd.schur_fast) {
        cnt[SL_SPLAIN] = pl_groups;
        cnt[SL_SFGP] = gp_groups;
    } else {
        cnt[SL_SGEN] = pl_groups + gp_groups;
    }
    cnt[SL_WIN] = 1;
}

__global__ __launch_bounds__(kSchedThreads) void k_sched_scan(BatchView bv, int round) {
    __shared__ int wave_tot[16][SL_COUNT];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int cnt[kSchedSlotsPerLane][SL_COUNT], tot[SL_COUNT];
#pragma unroll
    for (int k = 0; k < SL_COUNT; ++k) tot[k] = 0;
#pragma unroll
    for (int q = 0; q < kSchedSlotsPerLane; ++q) {
        const int s = t * kSchedSlotsPerLane + q;
#pragma unroll
        for (int k = 0; k < SL_COUNT; ++k) {
            cnt[q][k] = s < bv.n_slots ? bv.slot_cnt[(int64_t)s * (SL_COUNT + 1) + k] : 0;
            tot[k] += cnt[q][k];
        }
    }
    int off[SL_COUNT];
#pragma unroll
    for (int k = 0; k < SL_COUNT; ++k) {
        int x = tot[k];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        off[k] = x - tot[k];
        if (lane == 63) wave_tot[wave][k] = x;
    }
    __syncthreads();
    if (t < SL_COUNT) {
        int run = 0;
        for (int q = 0; q < 16; ++q) {
            const int v = wave_tot[q][t];
            wave_tot[q][t] = run;
            run += v;
        }
        bv.sched_lists[bv.sched_off[t]] = run;  // the list's count word
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kSchedSlotsPerLane; ++q) {
        const int s = t * kSchedSlotsPerLane + q;
        if (s >= bv.n_slots) continue;
#pragma unroll
        for (int k = 0; k < SL_COUNT; ++k) {
            bv.slot_cnt[(int64_t)s * (SL_COUNT + 1) + k] = wave_tot[wave][k] + off[k];
            off[k] += cnt[q][k];
        }
    }
    if (t == 0) {
        const int done = bv.sched_ctl[1];
        *(volatile int32_t*)(bv.sched_done_host + (round & 3)) = done;
        __threadfence_system();
    }
}

__global__ __launch_bounds__(256) void k_sched_fill(BatchView bv, SolveConsts c, int view_consts_here) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= bv.n_slots) return;
    const int32_t* off = bv.slot_cnt + (int64_t)s * (SL_COUNT + 1);
    const int kind = off[SL_COUNT];
    if (!kind) return;
    const int w = bv.slot_win[s];
    const WinDesc& wd = bv.win[w];
    auto L = [&](int k) { return bv.sched_lists + bv.sched_off[k] + 1 + off[k]; };
    if (kind == 2) {
        for (int i = lane; i < wd.n_blk; i += 64) L(SL_TBLK)[i] = wd.blk0 + i;
        for (int i = lane; i < wd.n_lblk; i += 64) L(SL_TLBLK)[i] = wd.lblk0 + i;
        if (lane == 0) L(SL_TWIN)[0] = w;
        return;
    }
    for (int i = lane; i < wd.n_lblk; i += 64) L(SL_LBLK)[i] = wd.lblk0 + i;
    const int n_pg = (wd.n_sblk_plain + c.schur_span - 1) / c.schur_span;
    const int n_gg = (wd.n_sblk - wd.n_sblk_plain + c.schur_span_gp - 1) / c.schur_span_gp;
    int32_t* lp = wd.schur_fast ? L(SL_SPLAIN) : L(SL_SGEN);
    int32_t* lg = wd.schur_fast ? L(SL_SFGP) : L(SL_SGEN) + n_pg;
    for (int i = lane; i < n_pg; i += 64) lp[i] = wd.sblk0 + i * c.schur_span;
    for (int i = lane; i < n_gg; i += 64) lg[i] = wd.sblk0 + wd.n_sblk_plain + i * c.schur_span_gp;
    if (lane == 0) L(SL_WIN)[0] = w;
    // the per-view constants of a window that is linearised this round (the former k_view_consts launch: the wave is here anyway)
    const WinState& st = bv.st[w];
    if (view_consts_here && st.active && st.need_lin && lane < wd.n_view) view_consts_item(bv, wd.view0 + lane);
}

// ------------------------------------------------------------------------------------------ observations
// per-view constants of the current poses (kba_items.hpp:view_consts_item), one lane per view
// (streaming solve: one 64-lane workgroup per listed window, lanes over its <= kMaxViews views)
__global__ void k_view_consts(BatchView bv) {
    int v;
    if (bv.counted) {
        const int32_t* wl = bv.sched_lists + bv.sched_off[SL_WIN] + 1;
        if ((int)blockIdx.x >= wl[-1]) return;
        const WinDesc& wd = bv.win[wl[blockIdx.x]];
        if ((int)threadIdx.x >= wd.n_view) return;
        v = wd.view0 + threadIdx.x;
    } else {
        v = blockIdx.x * blockDim.x + threadIdx.x;
        if (v >= bv.TV) return;
    }
    const WinState& st = bv.st[bv.view_win[v]];
    if (!st.active || !st.need_lin) return;
    view_consts_item(bv, v);
}

// ------------------------------------------------------------------------------------------ linearisation, landmark-major
// k_lin_lm: Jacobian evaluation (B1, B2, B5, B6) AND the landmark blocks V = sum E^T E, g = sum E^T r in one pass
// (kba_items.hpp:lin_lm_lane has the plain statements).  Lane = landmark, loop over the window's views:
//   * the view's 37 constants (kba_items.hpp:view_consts_item) are wave-uniform: the workgroup copies its window's constants
//     into LDS once and reads them from there (VLDS, the default since round 5); the other variant reads view_lin through the
//     constant address space (written by k_view_consts, the launch before: scalar loads although plane stores precede them);
//   * the observation of the pair (slot table -> index s -> u, v, d: 12 B) is fetched one view ahead, the slot two ahead;
//   * branch-free: a landmark that does not see the view (or is out of the problem) runs the same arithmetic on a valid
//     dummy observation, contributes zeros and stores into the dump area behind the planes;
//   * the 28 camera-side sums of the view leave each WAVE through one reduce-scatter into the wave's LDS slice (no
//     barrier in the loop); after the last view one barrier, then lane (view, entry) adds the four slices.
// Bytes per pair: 4 (slot) + 12 (u, v, d) read, 16 (planes au, sd) written; per landmark 32 read + 72 (V, g) + 48 (Bt) written -
// 79 B per observation under counters (round 5); the separate view-major linearise + landmark-major accumulate pair of round 1
// moved 101 + 93.
typedef const double __attribute__((address_space(4))) cdouble;
constexpr int kLinAccl = 13;  // doubles per lane that the ACCL variants of k_lin_lm park in LDS: V 6 | g 3 | Jacobi scale 3 | ground-plane row
// (the body of k_lin_lm for landmark workgroup b; k_solve_wg runs it for the workgroups of its window one after the other)
// KVIEW: the view constants are read through the constant address space (scalar loads) - only valid when they were
// written by an EARLIER launch (k_view_consts); k_solve_wg writes them in the same launch and reads them as plain memory.
template <bool KVIEW>
struct ViewPtr {
    typedef cdouble* type;
};
template <>
struct ViewPtr<false> {
    typedef const double* type;
};
// ACCL: the landmark block's nine running sums (V 6 | g 3) live in LDS ([9][kBlock] doubles in front of the view slices, one
// column per lane: conflict-free) instead of registers - read, three rows added, written back once per view.  18 registers
// less across the view loop: what the 128-register build (four waves per SIMD) spilled; the sums and their order are the same.
// VLDS: the window's view constants are copied into LDS once per workgroup ([view][kLinVlds], behind the ACCL region) and read from
// there (ds_read, broadcast) instead of through scalar loads: LDS reads return in order and are waited for one by one, scalar loads
// return out of order - every use of one waits for ALL of them (s_waitcnt lgkmcnt(0)), ten times per pair.
#ifndef KBA_COOP_VLDS
#define KBA_COOP_VLDS 1  // the one-launch kernels (k_solve_wg, k_solve_coop) linearise through LDS as well (0: A/B builds)
#endif
constexpr int kLinVlds = 38;  // doubles of a view's constants the linearisation reads (view_consts_item: 37)
template <bool KVIEW, bool ACCL = false, bool VLDS = false>
__device__ __forceinline__ void lin_lm_block(const BatchView& bv, const SolveConsts& c, int b) {
    const int w = bv.lblk_win[b];
    const WinState& st = bv.st[w];
    if (!st.active || !st.need_lin) return;
    const bool want_cost = st.first != 0;  // workgroup-uniform
    const WinDesc& wd = bv.win[w];
    const int n_view = wd.n_view;
    const int n = bv.lblk_n[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in_block = (int)threadIdx.x < n;
    const int gl = bv.lblk_lm0[b] + (in_block ? (int)threadIdx.x : n - 1);  // lanes past the end shadow the last landmark
    const int state = in_block ? bv.lm_state[gl] : 0;
    LinIn in;
    // A window whose last step was ACCEPTED linearises at the candidate point: the landmark is read from lm_c and moves into lm
    // right here (the streaming solve has no pass of its own for that: k_after_step; the other paths copy it themselves - the same
    // values, so reading lm_c instead of lm changes nothing for them).
    const bool take_candidate = st.accept != 0;  // (workgroup-uniform)
    const double* lm_src = take_candidate ? bv.lm_c : bv.lm;
    in.p[0] = lm_src[3 * (int64_t)gl];
    in.p[1] = lm_src[3 * (int64_t)gl + 1];
    in.p[2] = lm_src[3 * (int64_t)gl + 2];
    if (take_candidate && in_block) {
        bv.lm[3 * (int64_t)gl] = in.p[0];
        bv.lm[3 * (int64_t)gl + 1] = in.p[1];
        bv.lm[3 * (int64_t)gl + 2] = in.p[2];
    }
    in.w = bv.lm_weight[gl];
    in.sw = sqrt(in.w);
    LmTailIn tail;  // (fetched here, in front of the view loop's stores: kba_items.hpp:LmTailIn)
    lin_lm_tail_fetch(bv, w, gl, tail);
    const int32_t* slot = bv.lm_slot + gl;
    double* out = bv.lv_part + wd.lvpart_off + (int64_t)(b - wd.lblk0) * n_view * kLinPartial;
    extern __shared__ __attribute__((aligned(16))) double lin_lds[];  // ACCL: [kLinAccl][kBlock] sums, tail inputs | VLDS: [view][kLinVlds] | [view][wave][kLinPartial]
    double* const vlds = lin_lds + (ACCL ? kLinAccl * kBlock : 0);
    double* const lv_lds = vlds + (VLDS ? n_view * kLinVlds : 0);
    double* const accl = lin_lds + threadIdx.x;
    typedef typename std::conditional<VLDS, const double*, typename ViewPtr<KVIEW>::type>::type VT;
    constexpr int kVStride = VLDS ? kLinVlds : kViewLin;
    VT vc;
    if constexpr (VLDS) {
        const double* src = bv.view_lin + (int64_t)kViewLin * wd.view0;
        for (int i = threadIdx.x; i < n_view * kLinVlds; i += kBlock) vlds[i] = src[(i / kLinVlds) * kViewLin + i % kLinVlds];
        __syncthreads();
        vc = vlds;
    } else {
        vc = (VT)(bv.view_lin + (int64_t)kViewLin * wd.view0);
    }
    const int64_t dump = bv.SO - kObsBlock + threadIdx.x;
    LmAcc acc;
#pragma unroll
    for (int i = 0; i < 6; ++i) acc.V[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) acc.g[i] = 0.0;
    if constexpr (ACCL) {
#pragma unroll
        for (int i = 0; i < 9; ++i) accl[i * kBlock] = 0.0;
        // the tail's per-landmark inputs wait in LDS as well (fetched up here, in front of the loop's stores: LmTailIn)
#pragma unroll
        for (int i = 0; i < 3; ++i) accl[(9 + i) * kBlock] = tail.sc[i];
        accl[12 * kBlock] = __hiloint2double(tail.gg, 0);
    }
    // (a lane only ever touches its own column of the sums: no barrier between these accesses)
    auto accum = [&](auto vl, const double* r3, const double* c4) {
        if constexpr (ACCL) {
            LmAcc a;
#pragma unroll
            for (int i = 0; i < 6; ++i) a.V[i] = accl[i * kBlock];
#pragma unroll
            for (int i = 0; i < 3; ++i) a.g[i] = accl[(6 + i) * kBlock];
            lin_lm_accum(vl, r3, c4, a);
#pragma unroll
            for (int i = 0; i < 6; ++i) accl[i * kBlock] = a.V[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) accl[(6 + i) * kBlock] = a.g[i];
        } else {
            lin_lm_accum(vl, r3, c4, acc);
        }
    };
    int fail = 0;
    // pipeline: slot of view j + 2, measurement of view j + 1 in flight while view j computes
    int s_cur = slot[0];
    int s_nxt = n_view > 1 ? slot[bv.SL] : -1;
    float u_n, v_n, d_n;
    {
        const int64_t o = s_cur >= 0 ? s_cur : 0;
        u_n = bv.obs_u[o];
        v_n = bv.obs_v[o];
        d_n = bv.obs_d[o];
    }
    const int rs14_idx = rs14_index(lane);
    // ---- the leading views of keyframes WITHOUT a free pose block (WinDesc::n_view_fixed0: the Pose-fixed oldest keyframe of
    //      a sliding window): same pipeline, same planes, same landmark-block terms, but no pose Jacobian, no U / g, and only the
    //      cost leaves the wave (the slice's other 27 entries are zeros).  A loop of its own: the general loop below stays one
    //      straight-line body (a uniform branch inside it cost more than the skipped third of the arithmetic gave back).
    const int j_cam0 = wd.n_view_fixed0;
    for (int j = 0; j < j_cam0; ++j) {
        // (the constants are read where they are used - scalar loads the compiler places next to their instructions; a local
        // copy of all 55 would not fit the scalar register file)
        VT vl = vc + (int64_t)j * kVStride;
        const int s = s_cur;
        in.u = u_n;
        in.v = v_n;
        in.d = d_n;
        s_cur = s_nxt;
        s_nxt = j + 2 < n_view ? slot[(int64_t)(j + 2) * bv.SL] : -1;
        {
            const int64_t o = s_cur >= 0 ? s_cur : 0;
            u_n = bv.obs_u[o];
            v_n = bv.obs_v[o];
            d_n = bv.obs_d[o];
        }
        const bool have = state != 0 && s >= 0;
        in.live = have;
        LinLane l;
        double r3[3], c4[4];
        if (!lin_obs<false>(vl, c, in, want_cost, r3, c4, l)) fail = 1;
        {
            const int64_t o = have ? s : dump;
            bv.obs_c[o] = c4[0];
            bv.obs_c[bv.SO + o] = c4[3];
        }
        accum(vl, r3, c4);
        const double tot = wave_sum_all(l.e[0]);
        if (lane < kLinPartial) lv_lds[(j * kLinWaves + wave) * kLinPartial + lane] = lane == 0 ? tot : 0.0;
    }
    for (int j = j_cam0; j < n_view; ++j) {
        VT vl = vc + (int64_t)j * kVStride;
        const int s = s_cur;
        in.u = u_n;
        in.v = v_n;
        in.d = d_n;
        s_cur = s_nxt;
        s_nxt = j + 2 < n_view ? slot[(int64_t)(j + 2) * bv.SL] : -1;
        {
            const int64_t o = s_cur >= 0 ? s_cur : 0;
            u_n = bv.obs_u[o];
            v_n = bv.obs_v[o];
            d_n = bv.obs_d[o];
        }
        const bool have = state != 0 && s >= 0;
        in.live = have;
        LinLane l;
        double r3[3], c4[4];
        if (!lin_obs_core(vl, c, in, want_cost, r3, c4, l.e[0])) fail = 1;
        {   // of the four scalars of the factored Jacobian the Schur / back-substitution kernels read au and sd (16 B per pair) and
            // rebuild xn, yn from the landmark and the view (kba_math.hpp:view_xy: the same statements as here); the residual
            // r3 has done its work inside this lane (g += E^T r, camera-side g) - nobody reads it from memory in a solve
            const int64_t o = have ? s : dump;
            bv.obs_c[o] = c4[0];
            bv.obs_c[bv.SO + o] = c4[3];
        }
        accum(vl, r3, c4);  // zeros where the pair does not exist
        static_assert(kLinPartial == 28 && kLinWaves * 64 == kBlock, "wave_reduce_scatter14 x 2");
        // the camera-side RAW sums of the view (kba_items.hpp:lin_cam_half0 / _half1: 26 values - the 3 x 6 pose Jacobian is never
        // formed) leave the wave in two halves of 14 (register pressure: wave_reduce_scatter14)
        CamTmp ct;
        {
            double vals[14];
            lin_cam_half0(vl, in.p, c4, r3, l.e[0], ct, vals);
            const double tot = wave_reduce_scatter14(vals, lane);
            if (rs14_idx >= 0) lv_lds[(j * kLinWaves + wave) * kLinPartial + rs14_idx] = tot;
        }
        {
            double vals[14];
            lin_cam_half1(ct, vals);
            const double tot = wave_reduce_scatter14(vals, lane);
            if (rs14_idx >= 0) lv_lds[(j * kLinWaves + wave) * kLinPartial + 14 + rs14_idx] = tot;
        }
    }
    if constexpr (ACCL) {
#pragma unroll
        for (int i = 0; i < 6; ++i) acc.V[i] = accl[i * kBlock];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc.g[i] = accl[(6 + i) * kBlock];
#pragma unroll
        for (int i = 0; i < 3; ++i) tail.sc[i] = accl[(9 + i) * kBlock];
        tail.gg = __double2hiint(accl[12 * kBlock]);
    }
    double part[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    // the landmark's ground-plane row (B3) is linearised by its own lane right here, before lin_lm_finish adds it to the
    // landmark block (a separate k_gp launch per linearisation did this before: one launch per iteration less)
    if (in_block && tail.gg >= 0) gp_lane(bv, tail.gg, false, bv.gp_cost);
    if (state == 1) lin_lm_finish(bv, c, gl, in.p, tail, acc, part);
    __shared__ double lds[8];
    const double m = wave_max(part[0]);
    const double sm = wave_sum(part[1]);
    if (lane == 0) {
        lds[wave] = m;
        lds[4 + wave] = sm;
    }
    const int any_fail = __syncthreads_or(fail);                // a functor failed (the OR is logical, not bitwise)
    const int any_damp_fail = __syncthreads_or(part[5] != 0.0);  // a damped landmark block is not positive definite
    for (int e = threadIdx.x; e < n_view * kLinPartial; e += kBlock) {
        const double* q = lv_lds + (e / kLinPartial) * kLinWaves * kLinPartial + e % kLinPartial;
        out[e] = (q[0] + q[kLinPartial]) + (q[2 * kLinPartial] + q[3 * kLinPartial]);
    }
    if (threadIdx.x == 0) {
        bv.lblk_part[(int64_t)b * 8 + 0] = fmax(fmax(lds[0], lds[1]), fmax(lds[2], lds[3]));
        bv.lblk_part[(int64_t)b * 8 + 1] = (lds[4] + lds[5]) + (lds[6] + lds[7]);
        bv.lblk_part[(int64_t)b * 8 + 5] = any_damp_fail ? 1.0 : 0.0;
        bv.lblk_linfail[b] = any_fail ? 1.0 : 0.0;
    }
}
// k_lin_lm<WAVES, VLDS>: <4, true> is the default (128 registers: view constants, landmark sums and tail inputs in LDS, 33 KB per
// workgroup at five views); <3, true> the same at three waves per SIMD (KBA_LIN_WAVES=3); <., false> reads the view constants through
// scalar loads (windows with so many views that the LDS copy would cost occupancy; <3, false> keeps the sums in registers).  Same
// statements, same order, same bits in all of them.
template <int WAVES, bool VLDS>
#ifdef KBA_NOATTR_LIN
__global__ __launch_bounds__(kBlock) void k_lin_lm(
#else
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_lin_lm(
#endif
    BatchView bv, SolveConsts c, const int32_t* wl) {
    const int b = wl_at(bv, wl, blockIdx.x);
    if (b < 0) return;
    lin_lm_block<true, WAVES >= 4 || VLDS, VLDS>(bv, c, b);
}
// accl: the launch keeps its landmark sums in LDS (k_lin_lm<4, .>, k_lin_lm<., true>); vlds: ... and a copy of the view constants
__host__ __device__ inline int lin_lm_lds_bytes(int n_view_max, bool accl = false, bool vlds = false) {
    return (n_view_max * (kLinWaves * kLinPartial + (vlds ? kLinVlds : 0)) + (accl || vlds ? kLinAccl * kBlock : 0)) * (int)sizeof(double);
}

// ------------------------------------------------------------------------------------------ landmarks
__device__ __forceinline__ void lm_damp_block(const BatchView& bv, const SolveConsts& c, int b) {
    const int w = bv.lblk_win[b];
    if (!bv.st[w].active || !bv.st[w].redamp) return;  // after a linearisation k_lin_lm has damped already
    int fail = 0;
    if ((int)threadIdx.x < bv.lblk_n[b]) fail = lm_damp_lane(bv, c, w, bv.lblk_lm0[b] + threadIdx.x);
    const int any = __syncthreads_or(fail);
    if (threadIdx.x == 0) bv.lblk_part[(int64_t)b * 8 + 5] = any ? 1.0 : 0.0;
}
__global__ __launch_bounds__(kBlock) void k_lm_damp(BatchView bv, SolveConsts c, const int32_t* wl) {
    const int b = wl_at(bv, wl, blockIdx.x);
    if (b < 0) return;
    lm_damp_block(bv, c, b);
}

// back-substitution of the landmarks + the cost of their observations at the candidate point (kba_items.hpp:backsub_lane)
__device__ __forceinline__ void backsub_block(const BatchView& bv, const SolveConsts& c, int b) {
    const int w = bv.lblk_win[b];
    if (!bv.st[w].active) return;
    __shared__ double lds[16];
    double part[8];
    part[2] = part[3] = part[4] = part[6] = part[7] = 0.0;
    if ((int)threadIdx.x < bv.lblk_n[b]) {
        const int gl = bv.lblk_lm0[b] + threadIdx.x;
        backsub_lane(bv, c, w, gl, part);
        const int gg = bv.lm_gp[gl];  // cost of the landmark's ground-plane row at the candidate point (former k_gp(candidate) launch)
        if (gg >= 0) gp_lane(bv, gg, true, bv.gp_cost_c);
    }
    const int any_fail = __syncthreads_or(part[7] != 0.0);
    const double v4[4] = {part[2], part[3], part[4], part[6]};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double sum = wave_sum(v4[i]);
        if (lane == 0) lds[wave * 4 + i] = sum;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int slot = threadIdx.x < 3 ? 2 + threadIdx.x : 6;
        bv.lblk_part[(int64_t)b * 8 + slot] = (lds[threadIdx.x] + lds[4 + threadIdx.x]) + (lds[8 + threadIdx.x] + lds[12 + threadIdx.x]);
    }
    if (threadIdx.x == 0) bv.lblk_part[(int64_t)b * 8 + 7] = any_fail ? 1.0 : 0.0;
}
__global__ __launch_bounds__(kBlock) void k_backsub(BatchView bv, SolveConsts c, const int32_t* wl) {
    const int b = wl_at(bv, wl, blockIdx.x);
    if (b < 0) return;
    backsub_block(bv, c, b);
}

// ------------------------------------------------------------------------------------------ Schur complement (MFMA)
//   S -= sum_i Y'_i Y'_i^T  over the landmarks of a window, Y' = S_c F^T E S_l L^-T (kba_items.hpp).  Per tile of 16
//   landmarks the 48 x (nf + 1) matrix Z (rows = landmark coordinates, columns = free camera slots + the rhs) is built
//   in LDS and Z^T Z is accumulated with v_mfma_f64_16x16x4_f64 (A and B operands share the lane mapping: lane
//   (i = lane & 15, k = lane >> 4) supplies Z[4 ks + k][16 t + i]).  Two kernels:
//     k_schur_lean  windows with <= 4 free keyframes and one view per keyframe (the live configuration): one wave per
//                   group of blocks, everything sized for occupancy (below);
//     k_schur_wide  every other window (more free keyframes - up to kMaxKf - or several cameras per keyframe): a
//                   512-lane workgroup shares one Z tile, its eight waves own the output tiles.
typedef double v4f64 __attribute__((ext_vector_type(4)));

// ---- k_schur_wide<NPW>: 8 waves, wave w owns the upper tiles (tr <= tc) number w, w + 8, ... (NPW of them at most).
//   fill:  thread (li = t & 15, q = t >> 4) builds the 3 x 10 block of landmark li and the q-th free keyframe (q + 32
//          ... for more than 32) with kba_items.hpp:schur_pair_block (any number of views per keyframe) and writes it,
//          zeros included, into Z[48][nfp + 1];
//   syrk:  12 k-steps; per owned tile two panel reads and one MFMA.
// LDS: 48 (nfp + 1) doubles + the window's scales / columns / view table (98 KB at nfp = 256).
constexpr int kWideWaves = 8;
__host__ __device__ inline int schur_wide_lds_bytes(int nfp, int nc, int n_view) {
    return (3 * kSchurLm * (nfp + 1) + nc) * (int)sizeof(double) + (nc + n_view + kMaxKf + 2 + 2 * 160) * (int)sizeof(int);
}

template <int NPW>
__global__ __launch_bounds__(64 * kWideWaves) void k_schur_wide(BatchView bv, const int32_t* wl, int span, int span_gp) {
    const int sb = wl_at(bv, wl, blockIdx.x);  // first Schur block of this workgroup's group
    if (sb < 0) return;
    const int w = bv.sblk_win[sb];
    if (!bv.st[w].active) return;
    const WinDesc& wd = bv.win[w];
    const int nc = wd.nc, nfp = wd.nf_pad, nfq = wd.nfq, ld = nfp + 1;  // nfp is a multiple of 16: ld is odd
    const int T = nfp / 16, Tq = (nfq + 16) / 16;  // panels of a ground-plane tile / of a plain tile (columns 0..nfq)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Z = smem;                                      // [48][ld]
    double* sc_s = Z + 3 * kSchurLm * ld;                  // [nc] Jacobi scale by local slot
    int* zc_s = reinterpret_cast<int*>(sc_s + nc);         // [nc] tile column of a local slot or -1
    int* vkl = zc_s + nc;                                  // [n_view] local keyframe of each view
    int* fk = vkl + wd.n_view;                             // [kMaxKf] local keyframes with a free slot, [kMaxKf] their number
    int* ttr = fk + kMaxKf + 2;                            // [<= 160] tile -> panel row / column
    int* ttc = ttr + 160;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < nc; i += blockDim.x) {
        sc_s[i] = bv.scale_c[wd.cam0 + i];
        const int ci = bv.cslot[wd.cam0 + i];
        zc_s[i] = ci < 0 ? -1 : schur_col(ci, nfq);
    }
    for (int j = t; j < wd.n_view; j += blockDim.x) vkl[j] = bv.view_kf[wd.view0 + j] - wd.kf0;
    for (int i = t; i < 3 * kSchurLm * ld; i += blockDim.x) Z[i] = 0.0;
    if (t == 0) {
        int n = 0;
        for (int tr = 0; tr < T; ++tr)
            for (int tc = tr; tc < T; ++tc) {
                ttr[n] = tr;
                ttc[n] = tc;
                ++n;
            }
    }
    __syncthreads();
    if (t == 0) {
        int n = 0;
        for (int k = 0; k < wd.n_kf; ++k)
            if (zc_s[k * kCamSlots] >= 0 || zc_s[k * kCamSlots + 6] >= 0) fk[n++] = k;
        fk[kMaxKf] = n;
    }
    __syncthreads();
    const int nfk = fk[kMaxKf], n_tiles = T * (T + 1) / 2;
    v4f64 acc[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int li = t & 15;
    const int sb_last = schur_group_last(wd, sb, span, span_gp);  // blocks of one class (plain / ground-plane) only
    const int lm_first = bv.sblk_lm0[sb];
    const int n_lm_blk = bv.sblk_lm0[sb_last] + bv.sblk_n[sb_last] - lm_first;
    const bool tile_gp = sb - wd.sblk0 >= wd.n_sblk_plain;  // workgroup-uniform: the group's class
    const int Tt = tile_gp ? T : Tq;
    for (int l0 = 0; l0 < n_lm_blk; l0 += kSchurLm) {
        const int nl = min(kSchurLm, n_lm_blk - l0);
        const int gl = lm_first + l0 + li;
        const bool live = li < nl && bv.lm_state[gl] == 1;
        double lmk[6];
        if (live) schur_load_lm(bv, gl, lmk);
        if (t < kSchurLm) {
            double t3[3] = {0.0, 0.0, 0.0};
            if (live) {
                const double g3[3] = {bv.lm_g[gl], bv.lm_g[bv.SL + gl], bv.lm_g[2 * bv.SL + gl]};
                lm_t_of(lmk, g3, t3);
            }
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) Z[(3 * li + cc) * ld + nfq] = t3[cc];
        }
        for (int q = t >> 4; q < nfk; q += (64 * kWideWaves) >> 4) {
            const int kl = fk[q];
            double Y[3 * kCamSlots];
            if (live) {
                schur_pair_block(bv, wd, gl, kl, lmk, sc_s, vkl, tile_gp, Y);
            } else {
#pragma unroll
                for (int i = 0; i < 3 * kCamSlots; ++i) Y[i] = 0.0;
            }
#pragma unroll
            for (int a = 0; a < kCamSlots; ++a) {
                if (a < 6 || tile_gp) {
                    const int zc = zc_s[kl * kCamSlots + a];
                    if (zc >= 0) {
                        Z[(3 * li + 0) * ld + zc] = Y[a * 3 + 0];
                        Z[(3 * li + 1) * ld + zc] = Y[a * 3 + 1];
                        Z[(3 * li + 2) * ld + zc] = Y[a * 3 + 2];
                    }
                }
            }
        }
        __syncthreads();
        // Z^T Z, upper tiles owned by this wave: D[tr][tc] += sum_k Z[k][16 tr + i] Z[k][16 tc + j]
        const double* zp = Z + (lane >> 4) * ld + (lane & 15);
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int ti = wave + kWideWaves * j;
            if (ti < n_tiles) {  // wave-uniform
                const int tr = ttr[ti], tc = ttc[ti];
                if (tc < Tt) {
#pragma unroll
                    for (int ks = 0; ks < 12; ++ks)
                        acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(zp[ks * 4 * ld + 16 * tr], zp[ks * 4 * ld + 16 * tc], acc[j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    double* out = bv.S_part + wd.spart_off + (int64_t)schur_slab_of(wd, sb, span, span_gp) * ((int64_t)nfp * nfp);
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int ti = wave + kWideWaves * j;
        if (ti < n_tiles) {
            const int tr = ttr[ti], tc = ttc[ti];
#pragma unroll
            for (int r = 0; r < 4; ++r)  // f64 16x16x4 C/D layout: row = (lane>>4) + 4*reg, col = lane&15
                out[(tr * 16 + (lane >> 4) + 4 * r) * nfp + tc * 16 + (lane & 15)] = acc[j][r];
        }
    }
}

// ------------------------------------------------------------------------------------------ Schur complement, lean kernel
// k_schur_lean<TM, GP>: the Schur blocks of the fast class (WinDesc::schur_fast: <= 4 free keyframes, one view each),
// sized for OCCUPANCY - the general kernel above runs 2 waves / SIMD on ~250 registers and 18.8 KB of LDS and its
// per-tile chain fill -> LDS -> 12 k-steps is latency-bound.
//   GP = false: blocks of landmarks WITHOUT a ground-plane row (about 80 % of a window).  Their tiles only touch the
//               pose columns [0, nfq]: Z tile [48][nfq + 1] (25 columns for four free keyframes: 9.6 KB), TM <= 2 panels
//               -> 3 accumulator tiles, 128 registers -> 4 waves / SIMD;
//   GP = true : blocks of landmarks WITH a ground-plane row: all nf + 1 columns, TM <= 3 panels -> 6 accumulator tiles.
//   * No padding columns: a panel read past the last column runs into the next row - finite values that only reach
//     output entries nobody reads (masked to zero at the store).
//   * Keyframe constants (R, Rc, q, scales, columns) sit in LDS; the landmark-side inputs of ONE tile in registers: the
//     loads of the next tile are issued right after the fill and complete under the MFMAs of the current one; the
//     indices (slot, state, ground-plane row) are fetched two tiles ahead.
// One wave per workgroup, no cross-wave synchronisation; `span` consecutive blocks of one class per wave.
constexpr int kSpBatch = 4;  // k-steps whose panel reads are in flight together
constexpr int kSpKf = 72;    // doubles per free keyframe in LDS: R (9) | Rc (9) | q (4) | scale of its 10 slots | |q|^2 - 1 (+ 26 unused: the B_k of
                             // rounds 3-5, M(q, p) = sum_k p_k B_k - now M = -2 [Rh p]_x from R, kba_math.hpp:rot_tangent_from_R) |
                             // H (9), h0 (3) of its view (view_xy: xn, yn of an observation are rebuilt from the landmark)

__host__ __device__ inline int schur_lean_ld(int ncol) { return ncol | 1; }  // odd row stride: conflict-free fill
__host__ __device__ inline int schur_lean_lds_bytes(int ncol) {
    return (3 * kSchurLm * schur_lean_ld(ncol) + 16 + 4 * kSpKf) * (int)sizeof(double) + 4 * 12 * (int)sizeof(int);
}

// The group of Schur blocks that starts at block sb, by ONE wave.  COOP = false: the wave is the whole workgroup
// (k_schur_lean).  COOP = true: the wave is one of several in a workgroup that work on different groups at the same time
// (k_solve_coop) - it may not meet the others at a workgroup barrier, so the LDS hand-offs inside the wave are ordered by
// fences alone (the LDS operations of one wave complete in order), and `smem` is the wave's own LDS region.
template <bool COOP>
__device__ __forceinline__ void schur_wave_sync() {
    if constexpr (COOP) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        __syncthreads();
    }
}
template <int TM, bool GP, bool COOP>
__device__ __forceinline__ void schur_lean_group(const BatchView& bv, int sb, int span, int span_gp, double* smem) {
    const int w = bv.sblk_win[sb];
    // (COOP: k_solve_coop only gets here for a window that iterates - and workgroup 0 may be writing the window's LM state at
    // this very moment (lm_decide_lin runs beside the Schur phase), so the state is not read here)
    if (!COOP && !bv.st[w].active) return;
    const WinDesc& wd = bv.win[w];
    const int nfq = wd.nfq, nfp = wd.nf_pad;
    const int ncol = GP ? wd.nf + 1 : nfq + 1;  // columns of the tile, the rhs (column nfq) included
    const int ld = schur_lean_ld(ncol);
    const int Tt = (ncol + 15) / 16;
    // Two-tile Gram (plain blocks, 17..25 columns): the <= 24 pose columns are covered by TWO 16x16 products instead of
    // the three of a 2 x 2 upper tiling - rows {0..15} x columns {8..23}, then {0..7, 16..23} squared: every pair
    // (a <= b) of 24 columns lies in one of them - and the rhs column leaves the Gram: every (landmark, keyframe) lane
    // keeps Y'^T t' of its six slots over the tiles of the group, one 16-lane reduction at the end.  2 x 64 instead of
    // 3 x 64 MFMA cycles per k-step for 18 FMAs per lane and tile on the idle vector pipe.  The Gram entries keep their
    // bits (same products, same k order); the rhs entries are summed in another order.
    constexpr bool kTT = !GP && TM == 2;  // (plain blocks of fast windows have <= 4 free keyframes: nfq <= 24)
    const bool two_tile = kTT && Tt == 2;
    double* Z = smem;                                   // [48][ld] + 16 zeros (the last row's panel overrun)
    double* kc = Z + 3 * kSchurLm * ld + 16;            // [4][kSpKf]
    int* zcs = reinterpret_cast<int*>(kc + 4 * kSpKf);  // [4][12] tile column of the keyframe's slots (or -1)
    const int lane = COOP ? (int)(threadIdx.x & 63) : (int)threadIdx.x, li = lane & 15, kq = lane >> 4;
    const int n_fk = wd.n_fk;
    int my_view = -1, my_kl = -1;
    if (kq < n_fk) {
        my_kl = wd.fk[kq];
        my_view = wd.fk_view[kq];
        double* mine = kc + kq * kSpKf;
        const double* pose = bv.pose + 7 * (int64_t)(wd.kf0 + my_kl);
        if (li == 0) {
            double R[9];
            quat_R(pose, R);
#pragma unroll
            for (int i = 0; i < 9; ++i) mine[i] = R[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) mine[18 + i] = pose[i];
            mine[32] = quat_norm2_minus_1(pose);
        } else if (li < 10) {
            mine[9 + li - 1] = my_view >= 0 ? bv.view_cam[16 * (int64_t)my_view + 4 + li - 1] : 0.0;
        } else if (li == 13) {
            for (int i = 0; i < 12; ++i) mine[59 + i] = my_view >= 0 ? bv.view_lin[(int64_t)kViewLin * my_view + i] : 0.0;
        }
        if (li < kCamSlots) {
            const int slot = wd.cam0 + my_kl * kCamSlots + li;
            mine[22 + li] = bv.scale_c[slot];
            const int ci = bv.cslot[slot];
            zcs[kq * 12 + li] = ci < 0 ? -1 : schur_col(ci, nfq);
        }
    }
    if (lane < 16) Z[3 * kSchurLm * ld + lane] = 0.0;
    schur_wave_sync<COOP>();
    const bool have = kq < n_fk && zcs[kq * 12] >= 0;  // the keyframe's pose block is free (its six slots together)
    const double* mine = kc + (kq < n_fk ? kq : 0) * kSpKf;
    const int32_t* my_slots = bv.lm_slot + (int64_t)(my_view >= 0 ? my_view - wd.view0 : 0) * bv.SL;
    const int sb_last = schur_group_last(wd, sb, span, span_gp);
    const int lm_first = bv.sblk_lm0[sb];
    const int n_lm_blk = bv.sblk_lm0[sb_last] + bv.sblk_n[sb_last] - lm_first;

    constexpr int NT = (!GP && TM == 2) ? 2 : TM * (TM + 1) / 2;
    v4f64 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};

    // ---- software pipeline
    // stage 1 (two tiles ahead): landmark state, observation slot in this lane's keyframe, ground-plane row attached to it
    auto fetch_index = [&](int l0, int& st, int& slot, int& gg) {
        st = 0;
        slot = -1;
        gg = -1;
        if (l0 + li < n_lm_blk) {
            const int gl = lm_first + l0 + li;
            st = bv.lm_state[gl];
            if (have && my_view >= 0) slot = my_slots[gl];
            if (GP && have && st == 1) {
                const int g = bv.lm_gp[gl];
                if (g >= 0 && bv.gp_kf[g] - wd.kf0 == my_kl) gg = g;
            }
        }
    };
    // stage 2 (one tile ahead): the landmark-side inputs
    // (g3 is written by loads only: with a variable that the fill also overwrites - t = Bt g in place - the compiler copied the loaded
    // values into it right behind the loads, and that wait exposed the whole prefetch of a tile before its MFMAs)
    double c4[4], p[3], Bt[6], g3[3], gE[3], gF[kCamSlots];
    bool live = false, seen = false, att = false;
    auto fetch_data = [&](int l0, int st, int slot, int gg) {
        const int gl = lm_first + l0 + li;
        live = st == 1;
        seen = live && slot >= 0;
        att = GP && live && gg >= 0;
        if (seen) {
            c4[0] = bv.obs_c[slot];
            c4[3] = bv.obs_c[bv.SO + slot];
#pragma unroll
            for (int i = 0; i < 3; ++i) p[i] = bv.lm[3 * (int64_t)gl + i];
        }
        const bool want_t = live && (kq == 0 || two_tile);  // this lane supplies (its share of) the rhs column: t = Bt g
        if (seen || att || want_t) {
#pragma unroll
            for (int i = 0; i < 6; ++i) Bt[i] = bv.lm_Li[i * bv.SL + gl];
        }
        if (want_t) {
#pragma unroll
            for (int i = 0; i < 3; ++i) g3[i] = bv.lm_g[i * bv.SL + gl];  // (turned into t = Bt g at the fill)
        }
        if constexpr (GP) {
            if (att) {
#pragma unroll
                for (int i = 0; i < 3; ++i) gE[i] = bv.gp_E[i * bv.SG + gg];
#pragma unroll
                for (int i = 0; i < kCamSlots; ++i) gF[i] = bv.gp_F[i * bv.SG + gg];
            }
        }
    };
    int n_st, n_slot, n_gg;
    {
        int st0, slot0, gg0;
        fetch_index(0, st0, slot0, gg0);
        fetch_index(kSchurLm, n_st, n_slot, n_gg);
        fetch_data(0, st0, slot0, gg0);
    }
    constexpr int NS = GP ? kCamSlots : 6;  // slots of a keyframe this kernel fills
    double yt[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // two-tile path: this lane's share of the rhs, slots of its keyframe
    const int mq = li < 8 ? li : li + 8;            // two-tile path: column of lane li in the second operand {0..7, 16..23}
    for (int l0 = 0; l0 < n_lm_blk; l0 += kSchurLm) {
        // ---- fill: this lane's 3 x NS block of Y' (zeros where the landmark has no row with the keyframe)
        double t3[3] = {0.0, 0.0, 0.0};
        if (live && (kq == 0 || two_tile)) lm_t_of(Bt, g3, t3);
        if (kq == 0 && !two_tile) {
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) Z[(3 * li + cc) * ld + nfq] = live ? t3[cc] : 0.0;
        }
        if (have) {
            double Y[3 * NS];
#pragma unroll
            for (int i = 0; i < 3 * NS; ++i) Y[i] = 0.0;
            if (seen) {
                double M[9], Ft[9];
                rot_tangent_from_R(mine, mine[32], p, M);  // M(q, p) = -2 [Rh(q) p]_x: three entries of -2 Rh p instead of 27 products with the B_k
                view_xy(mine + 59, p, &c4[1], &c4[2]);
                ft_build(c4, mine + 9, Ft);
                schur_pose_block<true>(Ft, mine, M, Bt, mine + 22, Y);
                if (two_tile) {
#pragma unroll
                    for (int a = 0; a < 6; ++a) yt[a] += Y[a * 3 + 0] * t3[0] + Y[a * 3 + 1] * t3[1] + Y[a * 3 + 2] * t3[2];
                }
            }
            if constexpr (GP) {
                if (att) {  // the landmark's ground-plane row hangs on this keyframe: rank-one term over its free slots
                    const double y0 = gE[0] * Bt[0], y1 = gE[0] * Bt[1] + gE[1] * Bt[2], y2 = gE[0] * Bt[3] + gE[1] * Bt[4] + gE[2] * Bt[5];
#pragma unroll
                    for (int a = 0; a < kCamSlots; ++a) {
                        const double fa = zcs[kq * 12 + a] >= 0 ? gF[a] * mine[22 + a] : 0.0;
                        Y[a * 3 + 0] += fa * y0;
                        Y[a * 3 + 1] += fa * y1;
                        Y[a * 3 + 2] += fa * y2;
                    }
                }
            }
            double* zrow = Z + 3 * li * ld + zcs[kq * 12];  // the six pose slots of a keyframe are consecutive columns
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                zrow[a] = Y[a * 3 + 0];
                zrow[ld + a] = Y[a * 3 + 1];
                zrow[2 * ld + a] = Y[a * 3 + 2];
            }
            if constexpr (GP) {
#pragma unroll
                for (int a = 6; a < kCamSlots; ++a) {
                    const int zc = zcs[kq * 12 + a];
                    if (zc >= 0) {
                        Z[(3 * li + 0) * ld + zc] = Y[a * 3 + 0];
                        Z[(3 * li + 1) * ld + zc] = Y[a * 3 + 1];
                        Z[(3 * li + 2) * ld + zc] = Y[a * 3 + 2];
                    }
                }
            }
        }
        // ---- loads of the next tile (they complete under the MFMAs below), indices of the one after
        {
            const int st = n_st, slot = n_slot, gg = n_gg;
            fetch_index(l0 + 2 * kSchurLm, n_st, n_slot, n_gg);
            fetch_data(l0 + kSchurLm, st, slot, gg);
        }
        schur_wave_sync<COOP>();
        // ---- Z^T Z over the 48 rows (rows of absent landmarks are zero): 12 k-steps, upper tiles
        const double* zp = Z + kq * ld + li;
        if (TM == 1 || Tt == 1) {  // wave-uniform
#pragma unroll
            for (int h = 0; h < 12; h += kSpBatch) {
                double pa[kSpBatch];
#pragma unroll
                for (int j = 0; j < kSpBatch; ++j) pa[j] = zp[(h + j) * 4 * ld];
#pragma unroll
                for (int j = 0; j < kSpBatch; ++j) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[j], pa[j], acc[0], 0, 0, 0);
            }
        } else if (kTT) {
            if constexpr (kTT) {
                constexpr int kB = 3;  // k-steps whose three panel reads are in flight together
#pragma unroll
                for (int h = 0; h < 12; h += kB) {
                    double pa[kB], pb[kB], pq[kB];
#pragma unroll
                    for (int j = 0; j < kB; ++j) {
                        pa[j] = zp[(h + j) * 4 * ld];
                        pb[j] = zp[(h + j) * 4 * ld + 8];
                        pq[j] = Z[((h + j) * 4 + kq) * ld + mq];
                    }
#pragma unroll
                    for (int j = 0; j < kB; ++j) {
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[j], pb[j], acc[0], 0, 0, 0);  // rows 0..15 x cols 8..23
                        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(pq[j], pq[j], acc[1], 0, 0, 0);  // {0..7,16..23}^2
                    }
                }
            }
        } else if (TM == 2 || Tt == 2) {
            if constexpr (TM >= 2 && !kTT) {
                // tile order of acc for TM panels: (0,0) (0,1) .. (0,TM-1) (1,1) ..
                constexpr int i11 = TM;
#pragma unroll
                for (int h = 0; h < 12; h += kSpBatch) {
                    double pa[kSpBatch], pb[kSpBatch];
#pragma unroll
                    for (int j = 0; j < kSpBatch; ++j) {
                        pa[j] = zp[(h + j) * 4 * ld];
                        pb[j] = zp[(h + j) * 4 * ld + 16];
                    }
#pragma unroll
                    for (int j = 0; j < kSpBatch; ++j) {
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[j], pa[j], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[j], pb[j], acc[1], 0, 0, 0);
                        acc[i11] = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[j], pb[j], acc[i11], 0, 0, 0);
                    }
                }
            }
        } else {
            if constexpr (TM >= 3) {
#pragma unroll
                for (int h = 0; h < 12; h += 2) {
                    double pn[2][3];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int t = 0; t < 3; ++t) pn[j][t] = zp[(h + j) * 4 * ld + 16 * t];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(pn[j][0], pn[j][0], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(pn[j][0], pn[j][1], acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(pn[j][0], pn[j][2], acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(pn[j][1], pn[j][1], acc[3], 0, 0, 0);
                        acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(pn[j][1], pn[j][2], acc[4], 0, 0, 0);
                        acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(pn[j][2], pn[j][2], acc[5], 0, 0, 0);
                    }
                }
            }
        }
        schur_wave_sync<COOP>();
    }
    // ---- the slab of this group: tiles (tr <= tc) of the nfp x nfp matrix.  A ground-plane group writes all of them (zeros
    //      outside its columns included).  A plain group only writes the tiles that hold its pose columns and the rhs (rows and
    //      columns <= nfq): the readers of the slabs - cam_solve, slab_reduce_entry, k_solve_coop's slab sum - never look at a
    //      plain slab outside them (round 5: 6 KB instead of 12 KB per plain slab at C2), and a slab keeps its class for the
    //      life of the batch (schur_slab_of: plain groups first, fixed spans).
    double* out = bv.S_part + wd.spart_off + (int64_t)schur_slab_of(wd, sb, span, span_gp) * ((int64_t)nfp * nfp);
    const int T = nfp / 16;
    if (two_tile) {
        // the two products and the rhs sums meet in a 32 x 32 staging matrix in LDS (the Z tile is free now), from where
        // the slab is written in the layout of the three-tile path (upper tiles, zeros outside the Gram)
        double* St = Z;  // 1024 doubles <= 48 * ld
        for (int i = lane; i < 1024; i += 64) St[i] = 0.0;
        schur_wave_sync<COOP>();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = kq + 4 * r;
            {   // first product: entry (row, 8 + li)
                const int col = 8 + li;
                if (row <= col && col < nfq) St[row * 32 + col] = acc[0][r];
            }
            {   // second product: entry (m(row), m(li)); the block {0..7} x {16..23} is already there
                const int a = row < 8 ? row : row + 8, b = mq;
                if (a <= b && b < nfq && !(a < 8 && b >= 16)) St[a * 32 + b] = acc[1][r];
            }
        }
        if (have) {  // rhs: sum of this keyframe's six slot values over the 16 landmark lanes
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double v = yt[a];
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 8, 64);
                if (li == 0) St[(zcs[kq * 12] + a) * 32 + nfq] = v;
            }
        }
        schur_wave_sync<COOP>();
        for (int tc = 0; tc < (T < 2 ? T : 2); ++tc)
            for (int tr = 0; tr <= tc; ++tr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = tr * 16 + kq + 4 * r, col = tc * 16 + li;
                    out[row * nfp + col] = St[row * 32 + col];
                }
            }
        return;
    }
    if constexpr (kTT) {  // (one 16-column tile: windows of this batch with <= 2 free keyframes)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = kq + 4 * r, col = li;
            out[row * nfp + col] = (row < ncol && col < ncol) ? acc[0][r] : 0.0;
        }
    } else {
        int idx = 0;
#pragma unroll
        for (int tr = 0; tr < TM; ++tr)
#pragma unroll
            for (int tc = tr; tc < TM; ++tc) {
                if (tc < T) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // f64 16x16x4 C/D layout: row = (lane>>4) + 4*reg, col = lane&15
                        const int row = tr * 16 + kq + 4 * r, col = tc * 16 + li;
                        out[row * nfp + col] = (row < ncol && col < ncol) ? acc[idx][r] : 0.0;
                    }
                }
                ++idx;
            }
        if constexpr (GP) {
            for (int tc = TM; tc < T; ++tc)
                for (int tr = 0; tr <= tc; ++tr) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[(tr * 16 + kq + 4 * r) * nfp + tc * 16 + li] = 0.0;
                }
        }
    }
}
template <int TM, bool GP, int WAVES>
#ifdef KBA_NOATTR_SCHUR
__global__ __launch_bounds__(64) void k_schur_lean(
#else
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_schur_lean(
#endif
    BatchView bv, const int32_t* wl, int span, int span_gp) {
    const int sb = wl_at(bv, wl, blockIdx.x);
    if (sb < 0) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    schur_lean_group<TM, GP, false>(bv, sb, span, span_gp, smem);
}

// The plain and the ground-plane groups of a round in ONE launch (workgroups [0, n_plain_cap) take the plain list, the others the
// ground-plane list): for rounds with few windows in flight, where each of the two launches is one tile chain of latency (21 + 25 us
// at 16 windows: scripts/gpu_round_timeline.py) and the two lists are independent.  The kernel carries the larger variant's registers
// (218: two waves per SIMD), so it is only used while the batch drains (limo_hip.hip: kSchurPairBound); same device functions, same
// slabs, same bits.
template <int TMP, int TMG>
__global__ __launch_bounds__(64) void k_schur_lean_pair(BatchView bv, const int32_t* wl_plain, int n_plain_cap, const int32_t* wl_gp, int span, int span_gp) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if ((int)blockIdx.x < n_plain_cap) {
        const int sb = wl_at(bv, wl_plain, blockIdx.x);
        if (sb >= 0) schur_lean_group<TMP, false, false>(bv, sb, span, span_gp, smem);
    } else {
        const int sb = wl_at(bv, wl_gp, blockIdx.x - n_plain_cap);
        if (sb >= 0) schur_lean_group<TMG, true, false>(bv, sb, span, span_gp, smem);
    }
}

// ------------------------------------------------------------------------------------------ camera system
// (three waves per SIMD = three windows per CU: 168 registers; the ground-plane Gram tile of round 5 took the allocation to 172)
#ifndef KBA_CAM_ASM_WAVES
#define KBA_CAM_ASM_WAVES 3  // (0: no attribute - 172 registers, two waves per SIMD)
#endif
#ifndef KBA_CAM_SOLVE_WAVES
#define KBA_CAM_SOLVE_WAVES 0  // 0: no occupancy attribute on k_cam_solve (128 registers).  With amdgpu_waves_per_eu(3, 3) - or (4, 4) - the
                               // compiler's schedule is 27 % slower (287 vs 226 us per round, profiles/r05_experiment_cam_solve_occupancy.txt)
#endif
#if KBA_CAM_ASM_WAVES > 0
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(KBA_CAM_ASM_WAVES, KBA_CAM_ASM_WAVES))) void k_cam_assemble(
#else
__global__ __launch_bounds__(kBlock) void k_cam_assemble(
#endif
    BatchView bv, SolveConsts c, const int32_t* wl) {
    const int w = wl_at(bv, wl, blockIdx.x);
    if (w < 0) return;
    WinState& st = bv.st[w];
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (st.active && st.need_lin) {
        // large windows (> ~12 keyframes) assemble in global memory.  Two calls, not one with a selected pointer: the
        // scratch accesses of the common case must stay LDS instructions (a pointer that may be either makes them flat)
        const int64_t so = bv.win[w].cam_scr_off;
        if (so >= 0)
            cam_assemble(bv, c, w, threadIdx.x, blockDim.x, bv.cam_scratch + so);
        else
            cam_assemble(bv, c, w, threadIdx.x, blockDim.x, smem);
        __syncthreads();
#ifdef KBA_PROFILE_TICKS
        if (w == 0 && threadIdx.x == 0)
            for (int l = 0; l < 2; ++l)
                printf("[ticks cam_assemble %s lane] observation blocks %lld, ground-plane rows %lld, regulariser rows %lld, their sums %lld, mask + store %lld, reductions %lld\n",
                       l ? "last" : "first", kba_ticks[l][1] - kba_ticks[l][0], kba_ticks[l][2] - kba_ticks[l][1], kba_ticks[l][3] - kba_ticks[l][2],
                       kba_ticks[l][4] - kba_ticks[l][3], kba_ticks[l][5] - kba_ticks[l][4], kba_ticks[l][6] - kba_ticks[l][5]);
#endif
        if (threadIdx.x == 0) lm_decide_lin(st, bv.red[w], bv.reg_cost[2 * w + 1], c);
    }
    __syncthreads();
    // Count the windows that go on iterating; the workgroup that finishes last publishes the count straight into
    // pinned host memory (the host polls it one iteration behind) and re-arms the counters - no memset / copy
    // commands between the kernels of an iteration.
    if (threadIdx.x == 0 && !bv.counted) {
        if (st.active) atomicAdd(bv.n_active, 1);
        __threadfence();
        if (atomicAdd(bv.n_active + 1, 1) == (int)gridDim.x - 1) {
            __threadfence();
            const int total = atomicAdd(bv.n_active, 0);
            bv.n_active[0] = 0;
            bv.n_active[1] = 0;
            *(volatile int32_t*)bv.n_active_host = total;
            __threadfence_system();
        }
    }
}

#if KBA_CAM_SOLVE_WAVES > 0
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(KBA_CAM_SOLVE_WAVES, KBA_CAM_SOLVE_WAVES))) void k_cam_solve(
#else
__global__ __launch_bounds__(kBlock) void k_cam_solve(
#endif
    BatchView bv, SolveConsts c, const int32_t* wl) {
    const int w = wl_at(bv, wl, blockIdx.x);
    if (w < 0) return;
    if (!bv.st[w].active) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int flag;
    const int64_t so = bv.win[w].cam_scr_off;  // (two calls: see k_cam_assemble)
    if (so >= 0)
        cam_solve(bv, c, w, threadIdx.x, blockDim.x, bv.cam_scratch + so, &flag);
    else
        cam_solve(bv, c, w, threadIdx.x, blockDim.x, smem, &flag);
#ifdef KBA_PROFILE_TICKS
    __syncthreads();
    if (w == 0 && threadIdx.x == 0)
        for (int l = 0; l < 2; ++l)
            printf("[ticks cam_solve %s lane] slab sum %lld, cholesky %lld, back-substitution %lld, step %lld, reduction %lld\n", l ? "last" : "first",
                   kba_ticks[l][9] - kba_ticks[l][8], kba_ticks[l][10] - kba_ticks[l][9], kba_ticks[l][11] - kba_ticks[l][10],
                   kba_ticks[l][12] - kba_ticks[l][11], kba_ticks[l][13] - kba_ticks[l][12]);
#endif
}

__global__ __launch_bounds__(64) void k_step_decide(BatchView bv, SolveConsts c, const int32_t* wl) {
    const int w = wl_at(bv, wl, blockIdx.x);
    if (w < 0) return;
    if (!bv.st[w].active) return;
    __shared__ double red[64];
    reduce_step(bv, w, threadIdx.x, blockDim.x, red);
    __syncthreads();
    if (threadIdx.x == 0) lm_decide_step(bv.st[w], bv.red[w], c);
    if (!bv.counted) return;
    // streaming solve: the accepted keyframe parameters move here, per window (k_accept works per landmark workgroup and
    // a window WITHOUT landmark workgroups - every landmark filtered out, regularisers only - would never get them)
    __syncthreads();
    const WinDesc& wd = bv.win[w];
    if (bv.st[w].accept && (int)threadIdx.x < wd.n_kf) {
        const int64_t i = wd.kf0 + threadIdx.x;
        for (int q = 0; q < 7; ++q) bv.pose[7 * i + q] = bv.pose_c[7 * i + q];
        for (int q = 0; q < 3; ++q) bv.pdir[3 * i + q] = bv.pdir_c[3 * i + q];
        bv.pdist[i] = bv.pdist_c[i];
    }
}

// candidate -> current for accepted windows (keyframe part: first TK threads, landmark part: the rest)
// (streaming solve: one workgroup per listed landmark workgroup; the keyframes moved in k_step_decide)
__global__ void k_accept(BatchView bv) {
    if (bv.counted) {
        const int32_t* wl = bv.sched_lists + bv.sched_off[SL_LBLK] + 1;
        if ((int)blockIdx.x >= wl[-1]) return;
        const int b = wl[blockIdx.x], w = bv.lblk_win[b];
        if (!bv.st[w].accept) return;
        if ((int)threadIdx.x < bv.lblk_n[b]) {
            const int64_t l = bv.lblk_lm0[b] + threadIdx.x;
            for (int q = 0; q < 3; ++q) bv.lm[3 * l + q] = bv.lm_c[3 * l + q];
        }
        return;
    }
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

// Streaming solve: what a REJECTED step asks of a window's landmarks - the landmark blocks are damped again with the new radius, before
// the next round's Schur complement (the former k_lm_damp, at the end of the round instead of in the middle of the next one).  After
// an ACCEPTED step there is nothing to do here: the relinearisation (lin_lm_block) reads the candidate landmarks, moves them into lm
// and damps (the former k_accept: 48 B per landmark read and written by a pass of its own).  (Tried and dropped: the decision itself REPLICATED in every landmark workgroup instead of k_step_decide's launch - eight
// workgroups per window each summing the ground-plane costs and evaluating the regulariser rows cost 70 us per round more than the
// launch they saved: profiles/r06_experiment_launch_train.txt.)
__global__ __launch_bounds__(kBlock) void k_after_step(BatchView bv, SolveConsts c, const int32_t* wl) {
    const int b = wl_at(bv, wl, blockIdx.x);
    if (b < 0) return;
    const int w = bv.lblk_win[b];
    const WinState& st = bv.st[w];
    if (!st.accept && st.active && st.redamp) {
        int fail = 0;
        if ((int)threadIdx.x < bv.lblk_n[b]) fail = lm_damp_lane(bv, c, w, bv.lblk_lm0[b] + threadIdx.x);
        const int any = __syncthreads_or(fail);
        if (threadIdx.x == 0) bv.lblk_part[(int64_t)b * 8 + 5] = any ? 1.0 : 0.0;
    }
}

// ------------------------------------------------------------------------------------------ trimming
__global__ __launch_bounds__(kBlock) void k_trim_residual(BatchView bv, double* plane_rep, double* plane_dep, const int32_t* wl) {
    const int b = wl_at(bv, wl, blockIdx.x);
    if (b < 0) return;
    const int w = bv.view_win[bv.blk_view[b]];
    if (!bv.win[w].do_trim) return;
    for (int q = 0; q < kObsPerLane; ++q) trim_residual_lane(bv, b, threadIdx.x + q * kBlock, plane_rep, plane_dep);
}

// (streaming solve: one 256-lane workgroup per landmark workgroup of the windows being trimmed this round)
__global__ void k_trim_max(BatchView bv, const double* plane_rep, const double* plane_dep, int shard, int n_shards) {
    int gl;
    if (bv.counted) {
        const int32_t* wl = bv.sched_lists + bv.sched_off[SL_TLBLK] + 1;
        if ((int)blockIdx.x >= wl[-1]) return;
        const int b = wl[blockIdx.x];
        if ((int)threadIdx.x >= bv.lblk_n[b]) return;
        gl = bv.lblk_lm0[b] + threadIdx.x;
    } else {
        gl = blockIdx.x * blockDim.x + threadIdx.x;
    }
    if (gl >= bv.TL) return;
    if (n_shards > 1 && bv.lm_id[gl] % n_shards != shard) return;
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

// (streaming solve: the windows whose trimming solve ended this round; afterwards their next solve is armed)
__device__ __forceinline__ void trim_select_win(const BatchView& bv, const SolveConsts& c, int w) {
    const WinDesc& wd = bv.win[w];
    if (!wd.do_trim) return;
    const int n = wd.n_lm;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int n_valid;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    int removed = 0;
    if (np2 <= kTrimMaxSort) {
        // TrimmerQuantile::getOutliers (trimmer_quantile.hpp:40-63): the groups from rank num = (int)(n_groups * q) on, in
        // the order (value, id).  Only the rank SPLIT is needed, not the order: a radix select over the 96-bit key
        // [bits of the non-negative double | id], one byte per pass from the top - histogram of the byte among the
        // entries that match the bytes decided so far, one wave finds the bin that holds rank num - until a bin holds a
        // single entry (typically after 3-4 passes: the values are distinct) - then every entry at or above that key is an
        // outlier.  (Two full bitonic sorts of 2048 pairs, 66 barrier stages each, took 60 % of this kernel's time.)
        unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);  // [np2]
        unsigned* ids = reinterpret_cast<unsigned*>(smem + np2);                 // [np2]
        unsigned char* flags = reinterpret_cast<unsigned char*>(ids + np2);      // [n]
        __shared__ int hist[256];
        __shared__ int s_bin, s_k, s_cnt;
        __shared__ unsigned long long s_thr_hi;
        __shared__ unsigned s_thr_lo;
        for (int i = threadIdx.x; i < n; i += kBlock) flags[i] = 0;
        for (int list = 0; list < 2; ++list) {
            const double* vals = (list == 0 ? bv.trim_dep : bv.trim_rep) + wd.lm0;
            const int32_t* lid = bv.lm_id + wd.lm0;
            const double q = list == 0 ? c.depth_quantile : c.reprojection_quantile;
            if (threadIdx.x == 0) n_valid = 0;
            __syncthreads();
            int mine = 0;
            for (int i = threadIdx.x; i < n; i += kBlock) {
                const double v = vals[i];
                const bool valid = v >= 0.0;  // also true for +inf (failed functor)
                // non-negative doubles order like their bit patterns; entries without a residual sort after every valid one
                keys[i] = valid ? (v == 0.0 ? 0ull : (unsigned long long)__double_as_longlong(v)) : ~0ull;
                ids[i] = valid ? (unsigned)lid[i] : ((unsigned)i | (1u << 30));
                mine += valid;
            }
            if (mine) atomicAdd(&n_valid, mine);
            __syncthreads();
            const int ng = n_valid;
            const int num = (int)((double)ng * q);
            if (ng >= c.min_groups && num < ng) {  // (uniform over the workgroup)
                unsigned long long pre_hi = 0ull;
                unsigned pre_lo = 0u;
                int k = num;
                bool found = false;
                for (int byte = 11; byte >= 0 && !found; --byte) {
                    hist[threadIdx.x] = 0;  // kBlock == 256 bins
                    __syncthreads();
                    for (int i = threadIdx.x; i < n; i += kBlock) {
                        const unsigned long long kh = keys[i];
                        const unsigned kl = ids[i];
                        bool m;
                        unsigned bv8;
                        if (byte >= 4) {
                            const int sh = 8 * (byte - 4);
                            m = byte == 11 || (kh >> (sh + 8)) == (pre_hi >> (sh + 8));
                            bv8 = (unsigned)(kh >> sh) & 255u;
                        } else {
                            const int sh = 8 * byte;
                            m = kh == pre_hi && (((unsigned long long)kl >> (sh + 8)) == ((unsigned long long)pre_lo >> (sh + 8)));
                            bv8 = (kl >> sh) & 255u;
                        }
                        if (m) atomicAdd(&hist[bv8], 1);
                    }
                    __syncthreads();
                    if (threadIdx.x < 64) {  // one wave: the bin that holds rank k
                        const int l = threadIdx.x;
                        const int h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
                        const int tot = h0 + h1 + h2 + h3;
                        int incl = tot;
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) {
                            const int y = __shfl_up(incl, d, 64);
                            if (l >= d) incl += y;
                        }
                        const int excl = incl - tot;
                        if (excl <= k && k < incl) {
                            int r = k - excl, bin = 4 * l, cnt = h0;
                            if (r >= h0) {
                                r -= h0;
                                bin = 4 * l + 1;
                                cnt = h1;
                                if (r >= h1) {
                                    r -= h1;
                                    bin = 4 * l + 2;
                                    cnt = h2;
                                    if (r >= h2) {
                                        r -= h2;
                                        bin = 4 * l + 3;
                                        cnt = h3;
                                    }
                                }
                            }
                            s_bin = bin;
                            s_k = r;
                            s_cnt = cnt;
                        }
                    }
                    __syncthreads();
                    if (byte >= 4)
                        pre_hi |= (unsigned long long)s_bin << (8 * (byte - 4));
                    else
                        pre_lo |= (unsigned)s_bin << (8 * byte);
                    k = s_k;
                    if (s_cnt == 1 && byte > 0) {  // a single entry left: it is the threshold, fetch its full key
                        for (int i = threadIdx.x; i < n; i += kBlock) {
                            const unsigned long long kh = keys[i];
                            const unsigned kl = ids[i];
                            bool m;
                            if (byte >= 4)
                                m = (kh >> (8 * (byte - 4))) == (pre_hi >> (8 * (byte - 4)));
                            else
                                m = kh == pre_hi && (kl >> (8 * byte)) == (pre_lo >> (8 * byte));
                            if (m) {
                                s_thr_hi = kh;
                                s_thr_lo = kl;
                            }
                        }
                        __syncthreads();
                        pre_hi = s_thr_hi;
                        pre_lo = s_thr_lo;
                        found = true;
                    }
                }
                for (int i = threadIdx.x; i < n; i += kBlock) {
                    const unsigned long long kh = keys[i];
                    if (kh != ~0ull && (kh > pre_hi || (kh == pre_hi && ids[i] >= pre_lo))) flags[ids[i]] = 1;
                }
            }
            __syncthreads();
        }
        for (int l = threadIdx.x; l < n; l += kBlock) {
            if (flags[bv.lm_id[wd.lm0 + l]] && bv.lm_state[wd.lm0 + l]) {
                bv.lm_state[wd.lm0 + l] = 0;
                ++removed;
            }
        }
    } else {
        for (int l = threadIdx.x; l < n; l += kBlock) {
            const int out = trim_is_outlier(bv.trim_dep + wd.lm0, bv.lm_id + wd.lm0, n, l, c.depth_quantile, c.min_groups) ||
                            trim_is_outlier(bv.trim_rep + wd.lm0, bv.lm_id + wd.lm0, n, l, c.reprojection_quantile, c.min_groups);
            if (out && bv.lm_state[wd.lm0 + l]) {
                bv.lm_state[wd.lm0 + l] = 0;
                ++removed;
            }
        }
    }
    if (removed) atomicAdd(&bv.st[w].n_trimmed, removed);
}
__global__ __launch_bounds__(kBlock) void k_trim_select(BatchView bv, SolveConsts c) {
    int w = blockIdx.x;
    if (bv.counted) {
        const int32_t* wl = bv.sched_lists + bv.sched_off[SL_TWIN] + 1;
        if ((int)blockIdx.x >= wl[-1]) return;
        w = wl[blockIdx.x];
    }
    if (!bv.win[w].do_trim) return;
    trim_select_win(bv, c, w);
    if (bv.counted) {
        __syncthreads();
        if (threadIdx.x == 0) sched_after_trim(bv.st[w], c);
    }
}

// ------------------------------------------------------------------------------------------ a whole solve in one launch
// k_solve_wg: ONE 256-lane workgroup per window runs the complete solveTrimmed schedule of its window (kba_pack.cpp:
// run_schedule: trimming solves, retries, quantile trimming, final solve) - the phases the lock-step solve launches as
// kernels are the same device functions here, separated by workgroup barriers instead of launch boundaries, the landmark
// workgroups of the window taken one after the other.  For windows WITHOUT free landmarks (adjustPoseOnly,
// bundle_adjuster_keyframes.cpp:769-904: one keyframe against fixed landmarks, a few hundred observations): their
// solve is 9 LM iterations of 8 launches that each run for 4-10 us and wait 3 us for the next one; here the iteration is a
// few microseconds of work and six barriers.  Same arithmetic, same summation orders as the launches it replaces:
// the results are bit-identical (tests/test_gpu_parity.py compares the two paths).
//   * the view constants are written and read in the same launch: the workgroup copies them into LDS with plain loads
//     (lin_lm_block<false, true, true>), not through the scalar cache;
//   * the step reduction of k_step_decide is a 64-lane sum: reduce_step(.., n_work = 64) forms exactly that one;
//   * cap_ticks > 0: wall-clock cap of each solve in ticks of the 100 MHz constant clock (Solver::Options::
//     max_solver_time_in_seconds as run_schedule applies it: checked once per iteration after the linearisation).
__global__ __launch_bounds__(kBlock) void k_solve_wg(BatchView bv, SolveConsts c, long long cap_ticks, double* plane_rep, double* plane_dep) {
    const int w = blockIdx.x;
    const int tid = threadIdx.x;
    const WinDesc& wd = bv.win[w];
    WinState& st = bv.st[w];
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int flag;
    __shared__ double red[6 * (kBlock / 64)];
    const int64_t so = wd.cam_scr_off;
    const int lb0 = wd.lblk0, lb1 = wd.lblk0 + wd.n_lblk;
#ifdef KBA_WG_TICKS  // debug build: where the time of window 0 goes (100 MHz ticks, lane 0)
    long long tk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tl = wall_clock64();
    int n_it = 0, n_lin = 0;
#define KBA_WTICK(i)                         \
    do {                                     \
        const long long n_ = wall_clock64(); \
        tk[i] += n_ - tl;                    \
        tl = n_;                             \
    } while (0)
#else
#define KBA_WTICK(i)
#endif
    // The window's LM state is written by lane 0 only, and only behind a barrier that every lane passes AFTER its last
    // read of the state: all lanes take the same branches.
    // run_schedule: per trimming round [solve(trim_iters, windows that trim), solve(3 trim_iters, of those the ones whose
    // cost did not drop), trim], then solve(max_iters, all) - one loop over the solves, so the body exists once.
    const int n_solves = 2 * c.num_trim_rounds + 1;
    for (int si = 0; si < n_solves; ++si) {
        const bool final_solve = si == n_solves - 1, retry = !final_solve && (si & 1);
        const int max_iter = final_solve ? c.max_iters : (retry ? 3 * c.trim_iters : c.trim_iters);
        __syncthreads();
        if (tid == 0) {  // k_solve_init
            bool sel = true;
            if (!final_solve) sel = wd.do_trim != 0;
            if (retry) sel = sel && (st.solve_initial_cost - st.solve_final_cost <= 0.0);
            lm_solve_init(st, sel, max_iter, c);
        }
        __syncthreads();
        KBA_WTICK(0);
        const long long t0 = cap_ticks > 0 ? (long long)wall_clock64() : 0ll;
        for (;;) {
            // ---- linearize(): k_view_consts, k_lin_lm, k_cam_assemble
            if (st.active && st.need_lin) {
                if (tid < wd.n_view) view_consts_item(bv, wd.view0 + tid);
                __syncthreads();
                KBA_WTICK(1);
                for (int b = lb0; b < lb1; ++b) {
                    lin_lm_block<false, KBA_COOP_VLDS != 0, KBA_COOP_VLDS != 0>(bv, c, b);  // (view constants, sums and tail inputs in LDS: lin_lm_lds_bytes(., true, true))
                    __syncthreads();
                }
                KBA_WTICK(2);
                if (so >= 0)
                    cam_assemble(bv, c, w, tid, kBlock, bv.cam_scratch + so);
                else
                    cam_assemble(bv, c, w, tid, kBlock, smem);
                __syncthreads();
                KBA_WTICK(3);
                if (tid == 0) lm_decide_lin(st, bv.red[w], bv.reg_cost[2 * w + 1], c);
                __syncthreads();
                KBA_WTICK(4);
#ifdef KBA_WG_TICKS
                ++n_lin;
#endif
            }
            int active = st.active;
            if (cap_ticks > 0) {
                __syncthreads();
                if (tid == 0 && active && (long long)wall_clock64() - t0 >= cap_ticks) lm_terminate(st, LIMO_NO_CONVERGENCE);  // k_expire
                __syncthreads();
                active = st.active;
            }
            if (!active) break;
            // ---- step(): k_lm_damp, (no Schur blocks: no free landmark), k_cam_solve, k_backsub, k_step_decide, k_accept
            if (st.redamp) {
                for (int b = lb0; b < lb1; ++b) lm_damp_block(bv, c, b);
                __syncthreads();
            }
            KBA_WTICK(5);
            if (so >= 0)
                cam_solve(bv, c, w, tid, kBlock, bv.cam_scratch + so, &flag);
            else
                cam_solve(bv, c, w, tid, kBlock, smem, &flag);
            __syncthreads();
            KBA_WTICK(6);
            for (int b = lb0; b < lb1; ++b) {
                backsub_block(bv, c, b);
                __syncthreads();
            }
            KBA_WTICK(7);
            reduce_step(bv, w, tid, kBlock, red, 64);
            __syncthreads();
            if (tid == 0) lm_decide_step(st, bv.red[w], c);
            __syncthreads();
            KBA_WTICK(8);
#ifdef KBA_WG_TICKS
            ++n_it;
#endif
            if (st.accept) {
                if (tid < wd.n_kf) {
                    const int64_t i = wd.kf0 + tid;
                    for (int q = 0; q < 7; ++q) bv.pose[7 * i + q] = bv.pose_c[7 * i + q];
                    for (int q = 0; q < 3; ++q) bv.pdir[3 * i + q] = bv.pdir_c[3 * i + q];
                    bv.pdist[i] = bv.pdist_c[i];
                }
                for (int64_t l = wd.lm0 + tid; l < wd.lm0 + wd.n_lm; l += kBlock)
                    for (int q = 0; q < 3; ++q) bv.lm[3 * l + q] = bv.lm_c[3 * l + q];
            }
            __syncthreads();
            KBA_WTICK(9);
        }
        if (retry && wd.do_trim) {  // trim(): k_trim_residual, k_trim_max, k_trim_select
            __syncthreads();
            for (int b = wd.blk0; b < wd.blk0 + wd.n_blk; ++b)
                for (int q = 0; q < kObsPerLane; ++q) trim_residual_lane(bv, b, tid + q * kBlock, plane_rep, plane_dep);
            __syncthreads();
            for (int l = wd.lm0 + tid; l < wd.lm0 + wd.n_lm; l += kBlock) trim_max_lane(bv, l, plane_rep, plane_dep);
            __syncthreads();
            trim_select_win(bv, c, w);
            KBA_WTICK(10);
        }
    }
#ifdef KBA_WG_TICKS
    if (w == 0 && tid == 0)
        printf("[wg ticks] %d obs, %d landmark blocks, %d iterations, %d linearisations: init %lld | view consts %lld | lin %lld | assemble %lld | decide-lin %lld | damp %lld | cam solve %lld | backsub %lld | reduce+decide %lld | accept %lld | trim %lld (x10 ns)\n",
               (int)wd.n_obs, lb1 - lb0, n_it, n_lin, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5], tk[6], tk[7], tk[8], tk[9], tk[10]);
#endif
}

// ------------------------------------------------------------------------------------------ one window, one launch, G workgroups
// k_solve_coop: the complete solveTrimmed schedule of a window WITH free landmarks in one cooperative launch - G workgroups
// of 256 lanes per window that meet at device-wide barriers (coop_sync: 2.7 us at 9 workgroups, scripts/micro/
// grid_barrier.hip) where the lock-step solve has launch boundaries.  The single-window mode is what the reference runs
// (mono_lidar.cpp:203,255: one solve() per keyframe); its LM iteration was ten dependent launches, 165 us of kernels that
// each end on the critical path of one workgroup + 40 us between them.  Here an iteration is five barriers:
//     [re-linearise: k_lin_lm bodies, or re-damp]  |  [workgroup 0: camera system (k_cam_assemble body) WHILE the waves
//     of the other workgroups form the Schur complement (k_schur_lean bodies, one group of blocks per wave)]  |
//     [workgroup 0: k_cam_solve body]  |  [k_backsub bodies]  |  [workgroup 0: step decision, accepted poses, view
//     constants]
// (the first linearisation of a solve computes the Jacobi scaling of the camera columns the Schur blocks are scaled with:
// there the camera system comes first, one more barrier).  Same device functions, same partitions, same summation orders
// as the launches: bit-identical results (tests/test_gpu_ba.py).  Fast-class windows only (WinDesc::schur_fast, camera
// system in LDS); a barrier that is not met within half a second aborts the launch (pinned flag), it cannot hang the GPU.
struct CoopParams {
    int32_t G;                 // workgroups per window
    int32_t xcd_map;           // 1: the workgroups of a window share an XCD (grid = 8 G ceil(n_win / 8))
    int32_t vp, vg;            // k_schur_lean variant (TM) of the plain / ground-plane groups
    int32_t schur_lds;         // doubles of LDS per wave in the Schur phase
    long long cap_ticks;       // wall-clock cap of a solve (100 MHz ticks), 0: none
    long long timeout_ticks;   // a barrier that waits longer aborts the launch (the host then takes the launch sequence)
    int32_t* bar;              // [n_win][2][4] {arrived, generation, abort, -} x {all workgroups, all but the first}, zeroed before the launch
    int32_t* abort_host;       // pinned: set when a barrier timed out
    double *plane_rep, *plane_dep;
    double* red;               // [n_win][kCoopRedStride] the window's Schur slabs summed (see k_cam_solve below)
};
constexpr int kCoopRedStride = 64 * 64;  // nf_pad <= 64 for fast-class windows (<= 4 free keyframes: 40 slots + rhs)

// One device-wide barrier of the G workgroups of a window.  A barrier that is not met within `timeout` ticks of the 100 MHz
// constant clock ABORTS the launch: every workgroup of the window returns, the pinned flag tells the host, and the host
// restores the batch's initial state and solves it through the launch sequence instead (limo_ba_batch_solve) - a timeout is
// never an error the caller sees.  (That clock keeps running while a wave is preempted - a GPU shared with another process,
// a debugger or a profiler serialising dispatches - so a healthy launch CAN time out; the default is 50 ms, KBA_COOP_TIMEOUT_MS
// sets it.)  `abort_word` is shared by the window's main barrier and the barrier of its Schur workgroups: whoever gives up
// first releases everybody at their next poll.
__device__ __forceinline__ bool coop_sync(int32_t* bar, int32_t* abort_word, int G, int& gen, int32_t* abort_host, long long timeout) {
    if (G == 1) {
        __syncthreads();
        return true;
    }
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();  // the workgroup's writes (ordered before lane 0 by the barrier above) leave this XCD's L2
        const int arrived = __hip_atomic_fetch_add(&bar[0], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == G - 1) {
            __hip_atomic_store(&bar[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&bar[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const long long t0 = wall_clock64();
            // (relaxed polls: an acquire load invalidates this XCD's L2 on EVERY poll - under the workgroup that is working;
            // the fence behind the loop does it once)
            while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(2);
                if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || (long long)wall_clock64() - t0 > timeout) {
                    __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *(volatile int32_t*)abort_host = 1;
                    __threadfence_system();
                    ok = false;
                    break;
                }
            }
        }
        __threadfence();  // ... and what the others wrote is fetched again
    }
    ++gen;
    return __syncthreads_and(ok) != 0;
}

__global__ __launch_bounds__(kBlock) void k_solve_coop(BatchView bv, SolveConsts c, CoopParams a) {
    // Workgroup -> (window, member).  xcd_map: the G workgroups of a window are placed on ONE XCD (block b runs on XCD b % 8 -
    // observed, not promised: it buys speed, never correctness): what one of them writes with plain stores stays in the L2 the
    // others read from, so the first touches behind every barrier are L2 hits instead of round trips over the fabric
    // (MI355X_MICROARCH.md, handoff-payload: same-XCD 1.7x).  Window w lives on XCD w % 8; the grid is 8 G ceil(n_win / 8) and the
    // blocks that map to no window leave at once.
    const int G = a.G;
    int w, g;
    if (a.xcd_map) {
        const int x = blockIdx.x & 7, s = blockIdx.x >> 3;
        w = x + 8 * (s / G);
        g = s % G;
        if (w >= bv.n_win) return;
    } else {
        w = blockIdx.x / G;
        g = blockIdx.x % G;
    }
    const int tid = threadIdx.x, wave = tid >> 6;
    const WinDesc& wd = bv.win[w];
    WinState& st = bv.st[w];
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int flag;
    __shared__ double red[6 * (kBlock / 64)];
    int32_t* bar = a.bar + 8 * w;  // [0..3]: all G workgroups, [4..7]: the workgroups 1 .. G - 1 among themselves
    int gen = 0, gen_sub = 0;
#ifdef KBA_COOP_TICKS  // debug build: where the time of workgroups 0, 1 and G - 1 of window 0 goes (100 MHz ticks)
    long long tk[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tl = wall_clock64();
#define KBA_CTICK(i)                         \
    do {                                    \
        const long long n_ = wall_clock64(); \
        tk[i] += n_ - tl;                   \
        tl = n_;                            \
    } while (0)
#else
#define KBA_CTICK(i)
#endif
#define KBA_GSYNC()                                        \
    do {                                                   \
        if (!coop_sync(bar, bar + 2, G, gen, a.abort_host, a.timeout_ticks)) return; \
    } while (0)
    const int lb0 = wd.lblk0, lb1 = wd.lblk0 + wd.n_lblk;
    const int n_pg = (wd.n_sblk_plain + c.schur_span - 1) / c.schur_span;
    const int n_gg = (wd.n_sblk - wd.n_sblk_plain + c.schur_span_gp - 1) / c.schur_span_gp;
    // Schur phase: the groups of blocks of the window dealt over the waves of the workgroups from `first_worker` on
    auto schur_phase = [&](int first_worker) __attribute__((always_inline)) {
        if (g < first_worker) return;
        const int n_workers = (G - first_worker) * (kBlock / 64);
        double* mine = smem + (size_t)wave * a.schur_lds;
        for (int t = (g - first_worker) * (kBlock / 64) + wave; t < n_pg + n_gg; t += n_workers) {
            if (t < n_pg) {
                const int sb = wd.sblk0 + t * c.schur_span;
                if (a.vp == 1)
                    schur_lean_group<1, false, true>(bv, sb, c.schur_span, c.schur_span_gp, mine);
                else
                    schur_lean_group<2, false, true>(bv, sb, c.schur_span, c.schur_span_gp, mine);
            } else {
                const int sb = wd.sblk0 + wd.n_sblk_plain + (t - n_pg) * c.schur_span_gp;
                if (a.vg == 1)
                    schur_lean_group<1, true, true>(bv, sb, c.schur_span, c.schur_span_gp, mine);
                else if (a.vg == 2)
                    schur_lean_group<2, true, true>(bv, sb, c.schur_span, c.schur_span_gp, mine);
                else
                    schur_lean_group<3, true, true>(bv, sb, c.schur_span, c.schur_span_gp, mine);
            }
            schur_wave_sync<true>();  // the wave's LDS region is reused by its next group
        }
    };
    // The window's LM state is written by lane 0 of workgroup 0 only, and only between two barriers of which the first
    // comes after every other reader's last use of the old value: all workgroups take the same branches.
    const int n_solves = 2 * c.num_trim_rounds + 1;
    for (int si = 0; si < n_solves; ++si) {
        const bool final_solve = si == n_solves - 1, retry = !final_solve && (si & 1);
        const int max_iter = final_solve ? c.max_iters : (retry ? 3 * c.trim_iters : c.trim_iters);
        KBA_GSYNC();
        if (g == 0) {
            if (tid == 0) {  // k_solve_init
                bool sel = true;
                if (!final_solve) sel = wd.do_trim != 0;
                if (retry) sel = sel && (st.solve_initial_cost - st.solve_final_cost <= 0.0);
                lm_solve_init(st, sel, max_iter, c);
            }
            __syncthreads();
            if (st.active && st.need_lin && tid < wd.n_view) view_consts_item(bv, wd.view0 + tid);  // k_view_consts
        }
        KBA_GSYNC();
        const long long t0 = a.cap_ticks > 0 ? (long long)wall_clock64() : 0ll;
        for (;;) {
            // Sum of the window's Schur slabs, entry by entry, by the workgroups first .. G - 1 (n_sum of them): cam_solve's
            // first phase - S minus the sum of the slabs - is a chain of dependent memory round trips for ONE workgroup (a
            // third of the kernel); here every workgroup sums a share of the entries over all slabs, in the order cam_solve
            // adds them ((q mod 4) partial sums, then (a0 + a1) + (a2 + a3)), and workgroup 0 reads ONE slab:
            // s - ((R + 0) + (0 + 0)) has the bits of s - R.
            double* outr = a.red + (int64_t)w * kCoopRedStride;
            auto slab_sum = [&](int first) __attribute__((always_inline)) {
                if (g < first) return;
                const int nfp = wd.nf_pad, slab = nfp * nfp, n_slab = n_pg + n_gg, n_sum = G - first;
                const double* sp = bv.S_part + wd.spart_off;
                for (int e = (g - first) * kBlock + tid; e < slab; e += n_sum * kBlock) {
                    double acc[4] = {0.0, 0.0, 0.0, 0.0};
                    const int n4 = n_slab & ~3;
                    // (the plain groups' slabs are zero and unwritten outside the pose slots + rhs, rows and columns <= nfq of the
                    // slab: those entries start behind them, on a multiple of four - cam_solve's rule, cam_solve's bits)
                    const int q_first = (e / nfp > wd.nfq || e % nfp > wd.nfq) ? (n_pg & ~3) : 0;
                    for (int q0 = q_first; q0 < n_slab; q0 += 16) {  // 16 loads in flight, then added in slab order
                        double v[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j) v[j] = q0 + j < n_slab ? sp[(int64_t)(q0 + j) * slab + e] : 0.0;
#pragma unroll
                        for (int j = 0; j < 16; ++j)
                            if (q0 + j < n_slab) acc[q0 + j < n4 ? (j & 3) : 0] += v[j];
                    }
                    outr[e] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
                }
            };
            bool slabs_summed = false;
            if (st.active && st.need_lin) {
                const bool scale_first = st.compute_scale != 0 || G == 1;
                // ---- k_lin_lm
                for (int b = lb0 + g; b < lb1; b += G) {
                    lin_lm_block<false, KBA_COOP_VLDS != 0, KBA_COOP_VLDS != 0>(bv, c, b);  // (view constants, sums and tail inputs in LDS: lin_lm_lds_bytes(., true, true))
                    __syncthreads();
                }
                KBA_CTICK(0);
                KBA_GSYNC();
                KBA_CTICK(1);
                // ---- k_cam_assemble (workgroup 0) and the Schur complement (everybody else, or everybody afterwards)
                if (g == 0) {
                    cam_assemble(bv, c, w, tid, kBlock, smem);
                    __syncthreads();
                    if (tid == 0) lm_decide_lin(st, bv.red[w], bv.reg_cost[2 * w + 1], c);
                }
                KBA_CTICK(2);
                if (scale_first) {
                    KBA_GSYNC();
                    KBA_CTICK(3);
                    if (st.active) schur_phase(0);  // (behind the barrier: lm_decide_lin has finished writing the state)
                } else {
                    // ... and they also sum the slabs while workgroup 0 is still assembling (the camera system takes longer
                    // than the Schur complement): a barrier among the workgroups 1 .. G - 1 only
                    schur_phase(1);
                    KBA_CTICK(4);
                    if (g >= 1) {
                        if (!coop_sync(bar + 4, bar + 2, G - 1, gen_sub, a.abort_host, a.timeout_ticks)) return;
                        slab_sum(1);
                    }
                    slabs_summed = true;
                    KBA_CTICK(14);
                }
                KBA_CTICK(4);
            } else {
                // ---- k_lm_damp (after a rejected step), then the Schur complement
                for (int b = lb0 + g; b < lb1; b += G) lm_damp_block(bv, c, b);
                KBA_CTICK(13);
                KBA_GSYNC();
                KBA_CTICK(1);
                if (st.active) schur_phase(0);  // (nobody writes the LM state in this phase; a window that is not selected for this solve idles)
                KBA_CTICK(4);
            }
            KBA_GSYNC();
            KBA_CTICK(5);
            if (a.cap_ticks > 0) {
                if (g == 0 && tid == 0 && st.active && (long long)wall_clock64() - t0 >= a.cap_ticks) lm_terminate(st, LIMO_NO_CONVERGENCE);  // k_expire
                KBA_GSYNC();
            }
            if (!st.active) break;
            // ---- k_cam_solve
            if (G > 1) {
                if (!slabs_summed) {
                    slab_sum(0);
                    KBA_CTICK(14);
                    KBA_GSYNC();
                    KBA_CTICK(15);
                }
                if (g == 0) {
                    BatchView bvr = bv;
                    bvr.S_red = outr - wd.sred_off;
                    SolveConsts cr = c;
                    cr.schur_nslab = 1;
                    cam_solve(bvr, cr, w, tid, kBlock, smem, &flag);
                }
            } else {
                cam_solve(bv, c, w, tid, kBlock, smem, &flag);
            }
            KBA_CTICK(6);
            KBA_GSYNC();
            KBA_CTICK(7);
            // ---- k_backsub
            for (int b = lb0 + g; b < lb1; b += G) {
                backsub_block(bv, c, b);
                __syncthreads();
            }
            KBA_CTICK(8);
            KBA_GSYNC();
            KBA_CTICK(9);
            // ---- k_step_decide, the keyframe part of k_accept, k_view_consts of the next linearisation
            if (g == 0) {
                reduce_step(bv, w, tid, kBlock, red, 64);
                __syncthreads();
                if (tid == 0) lm_decide_step(st, bv.red[w], c);
                __syncthreads();
                if (st.accept && tid < wd.n_kf) {
                    const int64_t i = wd.kf0 + tid;
                    for (int q = 0; q < 7; ++q) bv.pose[7 * i + q] = bv.pose_c[7 * i + q];
                    for (int q = 0; q < 3; ++q) bv.pdir[3 * i + q] = bv.pdir_c[3 * i + q];
                    bv.pdist[i] = bv.pdist_c[i];
                }
                __syncthreads();
                if (st.active && st.need_lin && tid < wd.n_view) view_consts_item(bv, wd.view0 + tid);
            }
            KBA_CTICK(10);
            KBA_GSYNC();
            KBA_CTICK(11);
            // (the landmark part of k_accept is inside the relinearisation since round 6: lin_lm_block reads the candidate landmarks of
            // an accepted step and moves them into place - the workgroup that linearises a block is the one that used to copy it)
        }
        if (retry && wd.do_trim) {  // trim(): k_trim_residual, k_trim_max, k_trim_select
            KBA_GSYNC();
            for (int b = wd.blk0 + g; b < wd.blk0 + wd.n_blk; b += G)
                for (int q = 0; q < kObsPerLane; ++q) trim_residual_lane(bv, b, tid + q * kBlock, a.plane_rep, a.plane_dep);
            KBA_GSYNC();
            for (int l = wd.lm0 + g * kBlock + tid; l < wd.lm0 + wd.n_lm; l += G * kBlock) trim_max_lane(bv, l, a.plane_rep, a.plane_dep);
            KBA_GSYNC();
            if (g == 0) trim_select_win(bv, c, w);
        }
    }
#ifdef KBA_COOP_TICKS
    if (w == 0 && tid == 0 && (g == 0 || g == 1 || g == G - 1))
        printf("[coop ticks g=%d of %d] lin %lld | gb1 %lld | assemble %lld | gb-scale %lld | schur %lld | gb2 %lld | solve %lld | gb3 %lld | backsub %lld | gb4 %lld | decide %lld | gb5 %lld | damp %lld | slab sum %lld | gb-slab %lld (x10 ns)\n",
               g, G, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5], tk[6], tk[7], tk[8], tk[9], tk[10], tk[11], tk[13], tk[14], tk[15]);
#endif
#ifdef KBA_PROFILE_TICKS  // (debug build: the phase stamps cam_assemble / cam_solve took in their LAST call, shader clocks)
    if (w == 0 && g == 0 && tid == 0)
        for (int l = 0; l < 2; ++l) {
            printf("[coop ticks cam_assemble %s lane] observation blocks %lld, ground-plane rows %lld, regulariser rows %lld, their sums %lld, mask + store %lld, reductions %lld\n",
                   l ? "last" : "first", kba_ticks[l][1] - kba_ticks[l][0], kba_ticks[l][2] - kba_ticks[l][1], kba_ticks[l][3] - kba_ticks[l][2],
                   kba_ticks[l][4] - kba_ticks[l][3], kba_ticks[l][5] - kba_ticks[l][4], kba_ticks[l][6] - kba_ticks[l][5]);
            printf("[coop ticks cam_solve %s lane] slab sum %lld, cholesky %lld, back-substitution %lld, step %lld (of it: model cost change %lld), reduction %lld\n", l ? "last" : "first",
                   kba_ticks[l][9] - kba_ticks[l][8], kba_ticks[l][10] - kba_ticks[l][9], kba_ticks[l][11] - kba_ticks[l][10],
                   kba_ticks[l][12] - kba_ticks[l][11], kba_ticks[l][14] - kba_ticks[l][11], kba_ticks[l][13] - kba_ticks[l][12]);
        }
#endif
#undef KBA_CTICK
#undef KBA_GSYNC
}

// ------------------------------------------------------------------------------------------ landmark sharding
// Exchange step of the landmark-sharded solve when the shards live on ONE GPU ("virtual shards"): the
// same sum an RCCL all-reduce forms, in shard order.  Every entry has exactly one owner (the other shards hold
// zero), so the result is exact for any order.
struct ShardPtrs {
    const void* p[8];
};
template <typename T>
__global__ void k_sum_shards(T* dst, ShardPtrs src, int n_shards, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T acc = static_cast<const T*>(src.p[0])[i];
    for (int r = 1; r < n_shards; ++r) acc += static_cast<const T*>(src.p[r])[i];
    dst[i] = acc;
}

// A shard's contribution to [S | rhs] before the exchange: the entries the camera solve reads (upper triangle + rhs),
// summed over the shard's partial slabs - 33 KB on the wire for a C4 window instead of 9 MB of slabs.
__global__ __launch_bounds__(kBlock) void k_slab_reduce(BatchView bv, const int32_t* wl, int n_shards) {
    const int w = wl_at(bv, wl, blockIdx.x);
    if (w < 0) return;
    if (!bv.st[w].active) return;
    const int n = schur_need_count(bv.win[w].nf);
    for (int e = blockIdx.y * kBlock + threadIdx.x; e < n; e += gridDim.y * kBlock) slab_reduce_entry(bv, w, n_shards, e);
}

// The rest of a shard's block (kba_items.hpp:shard_reduce_*): phase 0 after the linearisation, 1 after a rejected step's
// damping, 2 after the back-substitution.  One workgroup per listed window.
__global__ __launch_bounds__(kBlock) void k_shard_reduce(BatchView bv, const int32_t* wl, int shard, int phase) {
    const int w = wl_at(bv, wl, blockIdx.x);
    if (w < 0) return;
    const WinState& st = bv.st[w];
    if (!st.active) return;
    if (phase == 0) {
        if (st.need_lin) shard_reduce_lin(bv, w, shard, threadIdx.x, kBlock);
    } else if (threadIdx.x == 0) {
        if (phase == 1) {
            if (st.redamp) shard_reduce_damp(bv, w, shard);
        } else {
            shard_reduce_step(bv, w, shard);
        }
    }
}

// After the all-gather of an exchange: the doubles [off, off + count) of shard `shard`'s block go to their places among the
// P contributions of the consumer view (kba_items.hpp:unpack_entry).  `win` = the batch's original window descriptors.
__global__ void k_unpack(ExchangeLayout L, const WinDesc* win, const double* range, double* arena, int shard, int64_t off, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) unpack_entry(L, win, range, (size_t)off, arena, shard, (size_t)(off + i));
}

// out = landmark positions of the landmarks this shard owns, zero elsewhere (input of the final all-reduce that
// gives every rank every landmark).
__global__ void k_lm_owned(BatchView bv, double* out, int rank, int n_shards, int world) {
    const int gl = blockIdx.x * blockDim.x + threadIdx.x;
    if (gl >= bv.TL) return;
    const bool mine = (bv.lm_id[gl] % n_shards) % world == rank;  // shard s lives on rank s mod world
    for (int i = 0; i < 3; ++i) out[3 * (int64_t)gl + i] = mine ? bv.lm[3 * (int64_t)gl + i] : 0.0;
}

// ------------------------------------------------------------------------------------------ evaluate (Problem::Evaluate)
// k_evaluate: the MATERIALISED Jacobian pass SURVEY 8d grades - per observation the residual, J_pose (tangent space) and J_point
// written out (ReprojectionErrorWithQuaternions + LandmarkDepthError through AutoDiffCostFunction in the reference,
// cost_functors_ceres.hpp:53-222, bundle_adjuster_keyframes.cpp:584-620).  A pure streaming-store kernel, so it is laid out for the
// store path (scripts/micro/store_stream.hip: thirty planes of 8-byte stores run at 5.6 TB/s WHEN every wave's 512 bytes of a plane
// are aligned; profiles/r06_experiment_evaluate_store_path.txt has the variants that were not):
//   * it writes the rows that EXIST: rows u, v of every observation (r 2 + J_pose 12 + J_point 6 doubles = 160 B) into planes over
//     the observations, and the depth row (1 + 6 + 3 doubles = 80 B) of the observations that HAVE a depth into compact planes over
//     the depth observations (BatchView::obs_r / obs_Jp / obs_Jl) - the algorithmic unit, 212 B + 84 B per depth observation, is
//     what moves (the round-2 kernel wrote 248 B for every observation);
//   * the unit of work is a WAVE on an ALIGNED range of 64 observations clipped to one observation block (EvalChunk, from the
//     pack; a range that straddles two views is two chunks): a wave's store to a plane is four whole 128-byte lines - no LDS, no
//     barrier; the rank of a depth observation among the batch's = the chunk's first rank + one ballot;
//   * workgroups are dealt to the XCDs in contiguous runs of chunks (block b runs on XCD b % 8): the lines two neighbouring waves
//     share - a compact depth plane's run ends anywhere - meet in ONE L2 and leave it whole;
//   * the view's constants (k_view_consts_all, the launch before) are wave-uniform: scalar loads, scalar registers; the arithmetic
//     is the solve's (eval_obs_head / _rows: rcp_nr / rsqrt_nr, closed-form rotation block), not round 2's IEEE chain rule;
//   * everything LOADED is consumed before the first store is issued (one counter for loads and stores on gfx950: a load waited
//     for behind a store waits for the store's acknowledge);
//   * the cost leaves the kernel as ONE double per wave (chunk_cost, fixed summation order), the validity flags as bytes.
#ifndef KBA_EVAL_NT
#define KBA_EVAL_NT 0  // 1: non-temporal stores for the planes (A/B builds)
#endif
__device__ __forceinline__ void store_plane(double* p, double v) {
#if KBA_EVAL_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
// per-view constants of EVERY view of the batch at the current poses (evaluate-only batches have no LM state to ask)
__global__ void k_view_consts_all(BatchView bv) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < bv.TV) view_consts_item(bv, v);
}
__host__ __device__ inline int evaluate_grid(int n_echunk) { return 8 * ((n_echunk + 8 * (kBlock / 64) - 1) / (8 * (kBlock / 64))); }
#ifndef KBA_EVAL_STAGE
#define KBA_EVAL_STAGE 1  // 1: the depth rows of a workgroup leave through LDS as whole 128-byte lines; 0: every lane stores its own (A/B builds)
#endif
__global__ __launch_bounds__(kBlock) void k_evaluate(BatchView bv, SolveConsts c, int apply_loss, double* chunk_cost, uint8_t* obs_valid) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // workgroup -> run of four chunks: XCD x (= blockIdx % 8, observed placement) takes the x-th eighth of the chunks
    const int per_xcd = gridDim.x >> 3;
    const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int ci = __builtin_amdgcn_readfirstlane(wg * (kBlock / 64) + wave);  // (wave-uniform: scalar loads below)
    const bool have = ci < bv.n_echunk;
#if KBA_EVAL_STAGE
    // The depth rows of the workgroup's (up to) 256 observations have consecutive ranks [R0, R1) - the chunks are consecutive and
    // ranks follow the packed order - so they are collected in LDS and written as whole lines: a lane per plane entry, 64 consecutive
    // entries per wave from a multiple of 16 on.  (Each lane storing its own row at its rank: a run of ~29 x 8 B at any alignment per
    // instruction = three partial-line writes, 1.65x the requests per byte; profiles/r06_experiment_evaluate_store_path.txt.)
    __shared__ double stage[10][kBlock];
    __shared__ int wave_end[kBlock / 64];
#else
    if (!have) return;
#endif
    const EvalChunk ch = bv.echunk[have ? ci : 0];
    cdouble* vl = (cdouble*)(bv.view_lin + (int64_t)kViewLin * ch.view);
    const int64_t o = (int64_t)ch.base + lane;
    const bool in = have && o < ch.o1;
    // (lanes past the block's end read the inert padding of the view's segment - valid entries - and store nothing)
    const float u = bv.obs_u[o], v = bv.obs_v[o], d = bv.obs_d[o];
    const int gl = bv.obs_lm[o];
    const double p[3] = {bv.lm[3 * (int64_t)gl], bv.lm[3 * (int64_t)gl + 1], bv.lm[3 * (int64_t)gl + 2]};
    const double w = bv.lm_weight[gl];
    const bool dep = in && d > 0.0f;
    const unsigned long long m = __ballot(dep);
    const int rank = ch.dep0 + __popcll(m & ((1ull << lane) - 1ull));  // packed order
    const int64_t SO = bv.SO, SD = bv.SD;
    EvalHead h;
    double Jp[18], Jl[9];
    eval_obs_head(vl, c, p, w, u, v, d, apply_loss != 0, h);
    __builtin_amdgcn_sched_barrier(0);  // (loads above, stores below)
    eval_obs_rows(vl, p, h, Jp, Jl);
    if (in) {
#pragma unroll
        for (int k = 0; k < 2; ++k) store_plane(bv.obs_r + k * SO + o, h.r[k]);
#pragma unroll
        for (int k = 0; k < 12; ++k) store_plane(bv.obs_Jp + k * SO + o, Jp[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) store_plane(bv.obs_Jl + k * SO + o, Jl[k]);
        obs_valid[o] = h.ok ? 1 : 0;
    }
#if KBA_EVAL_STAGE
    const int R0 = bv.echunk[min(wg * (kBlock / 64), bv.n_echunk - 1)].dep0;  // first rank of the workgroup (its first chunk exists whenever one does)
    if (dep) {
        const int lr = rank - R0;  // < kBlock: at most one depth row per observation of the workgroup
        stage[0][lr] = h.r[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) stage[1 + k][lr] = Jp[12 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) stage[7 + k][lr] = Jl[6 + k];
    }
    if (lane == 0) wave_end[wave] = have ? ch.dep0 + (int)__popcll(m) : 0;
    __syncthreads();
    const int R1 = max(max(wave_end[0], wave_end[1]), max(wave_end[2], wave_end[3]));
    for (int e = (R0 & ~15) + (int)threadIdx.x; e < R1; e += kBlock) {
        if (e < R0) continue;
        const int lr = e - R0;
        store_plane(bv.obs_r + 2 * SO + e, stage[0][lr]);
#pragma unroll
        for (int k = 0; k < 6; ++k) store_plane(bv.obs_Jp + 12 * SO + k * SD + e, stage[1 + k][lr]);
#pragma unroll
        for (int k = 0; k < 3; ++k) store_plane(bv.obs_Jl + 6 * SO + k * SD + e, stage[7 + k][lr]);
    }
    if (!have) return;
#else
    if (dep) {
        store_plane(bv.obs_r + 2 * SO + rank, h.r[2]);
#pragma unroll
        for (int k = 0; k < 6; ++k) store_plane(bv.obs_Jp + 12 * SO + k * SD + rank, Jp[12 + k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) store_plane(bv.obs_Jl + 6 * SO + k * SD + rank, Jl[6 + k]);
    }
#endif
    const double ws = wave_sum(in ? h.cost : 0.0);  // lanes in a fixed order: deterministic
    if (lane == 0) chunk_cost[ci] = ws;
}

// ------------------------------------------------------------------------------------------ evaluate rows (limo_ba_evaluate_rows)
// The rows of window `w` that are not reprojection / depth blocks, through the device functions the solve kernels call:
// gp_lane (k_lin_lm's ground-plane row: gp_r / gp_F / gp_E planes, gp_cost) and reg_row_eval (k_cam_assemble's regulariser
// rows).  One workgroup; only behind limo_ba_evaluate_rows.
__global__ __launch_bounds__(kBlock) void k_eval_rows(BatchView bv, int w, RegRow* rows, int32_t* fixed) {
    const WinDesc& wd = bv.win[w];
    for (int g = wd.gp0 + threadIdx.x; g < wd.gp0 + wd.n_gp; g += kBlock) gp_lane(bv, g, false, bv.gp_cost);
    const int nrows = reg_row_count(wd);
    for (int i = threadIdx.x; i < nrows; i += kBlock) {
        int all_const;
        reg_row_eval(wd, bv.cmask, bv.pose, bv.pdir, bv.pdist, i, true, rows[i], all_const);
        fixed[i] = all_const;
    }
}

}  // namespace kba
