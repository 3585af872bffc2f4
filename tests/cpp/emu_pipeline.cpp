// TEST INFRASTRUCTURE — serial CPU emulation of the batched HIP pipeline.
//
// Runs the SAME per-lane / per-workgroup statements as the gfx950 kernels (limo_amd/csrc/kba_items.hpp,
// kba_lm.hpp) and the SAME host packing and phase orchestration (kba_pack.cpp), with plain loops in place of
// lanes and workgroup reductions.  It exists so that the host logic and the kernel arithmetic can be unit-tested
// against the oracle in the CPU-only test tier (`pytest -m "not gpu"`).  It is NOT part of the product: it is
// built only into tests/cpp/_build/libkba_emu.so and nothing under limo_amd/ links or loads it.
#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../limo_amd/csrc/kba_buffers.hpp"
#include "../../limo_amd/csrc/kba_items.hpp"
#include "../../limo_amd/csrc/kba_rows.hpp"

using namespace kba;

namespace {

// Exchange callback of a landmark-sharded run whose shards are separate processes (tests: torch.distributed / gloo):
// kind 0: recv[count] = element-wise sum over all ranks of send[count] (trimming maxima, final landmarks);
// kind 2: ALL-GATHER - recv[world][count] = every rank's send[count], in rank order (the per-iteration exchange of the solve).
typedef void (*emu_allreduce_fn)(const void* send, void* recv, int64_t count, int kind, void* user);

struct EmuBatch : Executor {
    PackedBatch P;
    BatchView bv;  // consumer view (window-level items); == the only view when not sharded
    SolveConsts c;
    std::vector<void*> allocs;
    std::vector<double> plane_rep, plane_dep;
    // landmark sharding (mirrors limo_ba_batch): producer views of the local shards, partial arrays private to each
    int shard_P = 1;
    std::vector<int> local_shards;
    std::vector<BatchView> pv;
    ExchangeLayout xl;
    std::vector<WinDesc> win_c;    // the consumer's window descriptors ("P workgroups, P rows")
    double* arena_c = nullptr;     // the P contributions, side by side
    std::vector<double*> blocks;   // block of local shard i
    std::vector<double*> trims;    // trimming arenas: [0] consumer, [1 + i] local shard i
    int world = 1, rank = 0;
    bool first_lin = false, assemble_pending = false;
    long n_exchanges = 0;
    long long exchange_doubles = 0;  // doubles this rank SENT in the per-iteration exchanges
    emu_allreduce_fn cb = nullptr;
    void* cb_user = nullptr;

    ~EmuBatch() override {
        for (void* p : allocs) std::free(p);
    }
    void alloc() {
        std::memset(&bv, 0, sizeof(bv));
        for_each_buffer(P, bv, [&](void** slot, size_t bytes, const void* init) {
            void* p = std::calloc(1, bytes ? bytes : 1);
            if (init) std::memcpy(p, init, bytes);
            *slot = p;
            allocs.push_back(p);
        });
        for (int w = 0; w < bv.n_win; ++w) {
            bv.st[w].term = -1;
        }
        plane_rep.assign(bv.SO, -1.0);
        plane_dep.assign(bv.SO, -1.0);
        shard_P = P.n_shards;
        pv.assign(1, bv);
        if (shard_P > 1) {
            if (local_shards.empty())
                for (int r = 0; r < shard_P; ++r) local_shards.push_back(r);
            xl = exchange_layout(P);
            auto zeros = [&](size_t n) {
                double* q = static_cast<double*>(std::calloc(n ? n : 1, sizeof(double)));
                allocs.push_back(q);
                return q;
            };
            pv.assign(local_shards.size(), bv);  // producers keep the batch's own per-workgroup arrays (every entry has ONE owner)
            for (size_t i = 0; i < pv.size(); ++i) blocks.push_back(zeros(xl.b_total));
            trims.assign(1 + pv.size(), nullptr);
            for (double*& t : trims) t = zeros(xl.trim_count);
            for (size_t i = 0; i < pv.size(); ++i) {
                exchange_bind_producer(xl, pv[i], blocks[i], trims[1 + i]);
                pv[i].S_part = zeros(xl.spart_count);  // private per shard
            }
            win_c = exchange_consumer_windows(P);
            arena_c = zeros(xl.c_total);
            exchange_bind_consumer(xl, bv, arena_c, trims[0], win_c.data());
            c.schur_nslab = shard_P;
            c.schur_packed = 1;
        }
    }
    bool owns(size_t i, int owner) const { return shard_P == 1 || owner == local_shards[i]; }
    bool owns_lm(size_t i, int gl) const { return shard_P == 1 || bv.lm_id[gl] % shard_P == local_shards[i]; }
    bool owned_here(int gl) const {
        for (size_t i = 0; i < pv.size(); ++i)
            if (owns_lm(i, gl)) return true;
        return false;
    }
    // Per-iteration exchange (points 1 = A1, 2 = A2, 3 = A, 4 = B of kba_buffers.hpp): all-gather of the shards' blocks - one
    // call per local slot (normally one shard per rank: ONE call) - then unpack_entry puts every contribution at its shard's
    // place in the consumer view.  Shard s lives on rank s mod world, in local slot s / world.
    void exchange(int point) {
        if (shard_P == 1) return;
        size_t off, count;
        xl.range(point, off, count);
        std::vector<double> recv;
        for (size_t i = 0; i < pv.size(); ++i) {
            ++n_exchanges;
            exchange_doubles += (long long)count;
            if (cb) {
                recv.assign((size_t)world * count, 0.0);
                cb(blocks[i] + off, recv.data(), (int64_t)count, 2, cb_user);
                for (int r = 0; r < world; ++r)
                    for (size_t e = 0; e < count; ++e) unpack_entry(xl, P.win.data(), recv.data() + (size_t)r * count, off, arena_c, (int)i * world + r, off + e);
            } else {
                for (size_t e = 0; e < count; ++e) unpack_entry(xl, P.win.data(), blocks[i] + off, off, arena_c, local_shards[i], off + e);
            }
        }
    }
    // Trimming round: per-landmark residual maxima, one owner per entry and zero elsewhere - an exact sum.
    void exchange_trim() {
        if (shard_P == 1) return;
        for (size_t k = 0; k < xl.trim_count; ++k) {
            double a = trims[1][k];
            for (size_t i = 1; i < pv.size(); ++i) a += trims[1 + i][k];
            trims[0][k] = a;
        }
        ++n_exchanges;
        if (cb) cb(trims[0], trims[0], (int64_t)xl.trim_count, 0, cb_user);
    }

    void solve_init(int max_iter, int select) override {
        for (int w = 0; w < bv.n_win; ++w) {
            WinState& s = bv.st[w];
            bool sel = true;
            if (select >= 1) sel = bv.win[w].do_trim != 0;
            if (select == 2) sel = sel && (s.solve_initial_cost - s.solve_final_cost <= 0.0);
            lm_solve_init(s, sel, max_iter, c);
        }
        first_lin = true;  // the next linearisation defines the Jacobi scaling (WinState::compute_scale)
        assemble_pending = false;
    }

    // camera assembly + the LM decision at the linearisation point (k_cam_assemble)
    void assemble() {
        std::vector<double> H((size_t)cam_assemble_scratch(kMaxNc, 1, kMaxViews));
        for (int w = 0; w < bv.n_win; ++w) {
            if (!bv.st[w].active || !bv.st[w].need_lin) continue;
            cam_assemble(bv, c, w, 0, 1, H.data());
            lm_decide_lin(bv.st[w], bv.red[w], bv.reg_cost[2 * w + 1], c);
        }
    }

    void linearize() override {
        for (int view = 0; view < bv.TV; ++view) {  // per-view constants of the current poses (k_view_consts)
            const WinState& s = bv.st[bv.view_win[view]];
            if (s.active && s.need_lin) view_consts_item(bv, view);
        }
        for (size_t si = 0; si < pv.size(); ++si) {
            BatchView& v = pv[si];
            // ground-plane rows
            for (int w = 0; w < v.n_win; ++w) {
                if (!v.st[w].active || !v.st[w].need_lin) continue;
                const WinDesc& wd = v.win[w];
                for (int g = wd.gp0; g < wd.gp0 + wd.n_gp; ++g)
                    if (owns_lm(si, v.gp_lm[g])) gp_lane(v, g, false, v.gp_cost);
            }
            // landmark-major linearisation (k_lin_lm): observations of every landmark, planes, landmark blocks; the
            // camera-side sums per (landmark workgroup, view, wave)
            std::vector<LinLane> cam(kMaxViews);
            for (int b = 0; b < v.n_lblk; ++b) {
                if (!owns(si, shard_P > 1 ? P.lblk_owner[b] : 0)) continue;
                const int w = v.lblk_win[b];
                if (!v.st[w].active || !v.st[w].need_lin) continue;
                const WinDesc& wd = v.win[w];
                double* out = v.lv_part + wd.lvpart_off + (int64_t)(b - wd.lblk0) * wd.n_view * kLinPartial;
                std::vector<double> slices((size_t)wd.n_view * kLinWaves * kLinPartial, 0.0);  // [view][wave][28]
                double gmax = 0.0, xn2 = 0.0;
                int fail = 0, damp_fail = 0;
                for (int t = 0; t < v.lblk_n[b]; ++t) {
                    double part[8];
                    fail |= lin_lm_lane(v, c, w, v.lblk_lm0[b] + t, v.st[w].first != 0, cam.data(), part);
                    gmax = std::fmax(gmax, part[0]);
                    xn2 += part[1];
                    damp_fail |= part[5] != 0.0;
                    for (int j = 0; j < wd.n_view; ++j) {
                        double* o = slices.data() + ((size_t)j * kLinWaves + t / 64) * kLinPartial;
                        for (int i = 0; i < kLinPartial; ++i) o[i] += cam[j].e[i];
                    }
                }
                for (int e = 0; e < wd.n_view * kLinPartial; ++e) {
                    const double* q = slices.data() + (size_t)(e / kLinPartial) * kLinWaves * kLinPartial + e % kLinPartial;
                    out[e] = (q[0] + q[kLinPartial]) + (q[2 * kLinPartial] + q[3 * kLinPartial]);
                }
                v.lblk_part[(int64_t)b * 8 + 0] = gmax;
                v.lblk_part[(int64_t)b * 8 + 1] = xn2;
                v.lblk_part[(int64_t)b * 8 + 5] = damp_fail ? 1.0 : 0.0;
                v.lblk_linfail[b] = fail;
            }
        }
        if (shard_P > 1) {
            for (size_t si = 0; si < pv.size(); ++si)
                for (int w = 0; w < bv.n_win; ++w)
                    if (pv[si].st[w].active && pv[si].st[w].need_lin) shard_reduce_lin(pv[si], w, local_shards[si], 0, 1);
            // Sharded: the camera assembly only has to come before the Schur complement when it defines the Jacobi scale (first
            // linearisation of a solve).  Every other iteration it waits for the slabs and ONE exchange serves both.
            if (!first_lin) {
                assemble_pending = true;
                return;
            }
            first_lin = false;
            exchange(1);
        }
        assemble();
    }

    int active_count() override {
        int n = 0;
        for (int w = 0; w < bv.n_win; ++w) n += bv.st[w].active;
        return n;
    }

    void expire(int) override {
        if (assemble_pending) {  // (as limo_hip.hip: the deferred camera assembly of the last linearisation)
            assemble_pending = false;
            exchange(1);
            assemble();
        }
        for (int w = 0; w < bv.n_win; ++w)
            if (bv.st[w].active) lm_terminate(bv.st[w], LIMO_NO_CONVERGENCE);
    }

    void step() override {
        std::vector<double> z;
        for (size_t si = 0; si < pv.size(); ++si) {
            BatchView& v = pv[si];
            // landmark damping
            for (int b = 0; b < v.n_lblk; ++b) {
                if (!owns(si, shard_P > 1 ? P.lblk_owner[b] : 0)) continue;
                const int w = v.lblk_win[b];
                if (!v.st[w].active || !v.st[w].redamp) continue;
                int fail = 0;
                for (int t = 0; t < v.lblk_n[b]; ++t) fail |= lm_damp_lane(v, c, w, v.lblk_lm0[b] + t);
                v.lblk_part[(int64_t)b * 8 + 5] = fail ? 1.0 : 0.0;
            }
            // Schur slabs: upper triangle of Z^T Z per Schur workgroup (rhs = column nfq, see kba_items.hpp)
            for (int sb = 0; sb < v.n_sblk; ++sb) {
                if (!owns(si, shard_P > 1 ? P.sblk_owner[sb] : 0)) continue;
                const int w = v.sblk_win[sb];
                if (!v.st[w].active) continue;
                const WinDesc& wd = v.win[w];
                const int nfp = wd.nf_pad, nfq = wd.nfq;
                const int slab = nfp * nfp;
                std::vector<int> vkl(wd.n_view);
                for (int j = 0; j < wd.n_view; ++j) vkl[j] = v.view_kf[wd.view0 + j] - wd.kf0;
                double* out = v.S_part + wd.spart_off + (int64_t)(sb - wd.sblk0) * slab;
                for (int i = 0; i < slab; ++i) out[i] = 0.0;
                for (int li = 0; li < v.sblk_n[sb]; ++li) {
                    const int gl = v.sblk_lm0[sb] + li;
                    if (v.lm_state[gl] != 1) continue;
                    double lmk[9], Y[3 * kCamSlots];
                    schur_load_lm(v, gl, lmk);
                    z.assign((size_t)3 * nfp, 0.0);
                    for (int kl = 0; kl < wd.n_kf; ++kl) {
                        schur_pair_block(v, wd, gl, kl, lmk, v.scale_c + wd.cam0, vkl.data(), gl >= wd.lm_gp0, Y);
                        for (int a = 0; a < kCamSlots; ++a) {
                            const int ci = v.cslot[wd.cam0 + kl * kCamSlots + a];
                            if (ci < 0) continue;
                            for (int cc = 0; cc < 3; ++cc) z[(size_t)cc * nfp + schur_col(ci, nfq)] = Y[a * 3 + cc];
                        }
                    }
                    {
                        const double g3[3] = {v.lm_g[gl], v.lm_g[v.SL + gl], v.lm_g[2 * v.SL + gl]};
                        double t3[3];
                        lm_t_of(lmk, g3, t3);
                        for (int cc = 0; cc < 3; ++cc) z[(size_t)cc * nfp + nfq] = t3[cc];
                    }
                    for (int cc = 0; cc < 3; ++cc) {
                        const double* zr = z.data() + (size_t)cc * nfp;
                        for (int a = 0; a <= wd.nf; ++a) {
                            if (zr[a] == 0.0) continue;
                            for (int bcol = a; bcol <= wd.nf; ++bcol) out[a * nfp + bcol] += zr[a] * zr[bcol];
                        }
                    }
                }
            }
        }
        if (shard_P > 1) {
            for (size_t si = 0; si < pv.size(); ++si)
                for (int w = 0; w < bv.n_win; ++w) {
                    if (!pv[si].st[w].active) continue;
                    if (pv[si].st[w].redamp) shard_reduce_damp(pv[si], w, local_shards[si]);
                    const int n = schur_need_count(pv[si].win[w].nf);
                    for (int e = 0; e < n; ++e) slab_reduce_entry(pv[si], w, shard_P, e);
                }
            if (assemble_pending) {  // ONE exchange: camera-side sums + ground-plane blocks + scalars + [S | rhs]
                assemble_pending = false;
                exchange(3);
                assemble();
            } else {
                exchange(2);
            }
        }
        // camera solve
        std::vector<double> S((size_t)cam_solve_scratch(kMaxNc, 1));
        for (int w = 0; w < bv.n_win; ++w) {
            if (!bv.st[w].active) continue;
            int flag = 0;
            cam_solve(bv, c, w, 0, 1, S.data(), &flag);
        }
        for (size_t si = 0; si < pv.size(); ++si) {
            BatchView& v = pv[si];
            // back-substitution
            for (int b = 0; b < v.n_lblk; ++b) {
                if (!owns(si, shard_P > 1 ? P.lblk_owner[b] : 0)) continue;
                const int w = v.lblk_win[b];
                if (!v.st[w].active) continue;
                double mcc = 0.0, s2 = 0.0, c2 = 0.0, cost = 0.0, fail = 0.0;
                for (int t = 0; t < v.lblk_n[b]; ++t) {
                    double part[8];
                    backsub_lane(v, c, w, v.lblk_lm0[b] + t, part);
                    mcc += part[2];
                    s2 += part[3];
                    c2 += part[4];
                    cost += part[6];
                    if (part[7] != 0.0) fail = 1.0;
                }
                v.lblk_part[(int64_t)b * 8 + 2] = mcc;
                v.lblk_part[(int64_t)b * 8 + 3] = s2;
                v.lblk_part[(int64_t)b * 8 + 4] = c2;
                v.lblk_part[(int64_t)b * 8 + 6] = cost;  // candidate cost of the observations of these landmarks
                v.lblk_part[(int64_t)b * 8 + 7] = fail;
            }
            for (int w = 0; w < v.n_win; ++w) {
                if (!v.st[w].active) continue;
                const WinDesc& wd = v.win[w];
                for (int g = wd.gp0; g < wd.gp0 + wd.n_gp; ++g)
                    if (owns_lm(si, v.gp_lm[g])) gp_lane(v, g, true, v.gp_cost_c);
            }
        }
        if (shard_P > 1)
            for (size_t si = 0; si < pv.size(); ++si)
                for (int w = 0; w < bv.n_win; ++w)
                    if (pv[si].st[w].active) shard_reduce_step(pv[si], w, local_shards[si]);
        exchange(4);
        for (int w = 0; w < bv.n_win; ++w) {
            if (!bv.st[w].active) continue;
            double red1[6];
            reduce_step(bv, w, 0, 1, red1);
            const double x_cost_before = bv.st[w].x_cost, radius_before = bv.st[w].radius;
            lm_decide_step(bv.st[w], bv.red[w], c);
            if (std::getenv("EMU_VERBOSE"))
                std::fprintf(stderr, "[emu] w%d it %d  x_cost %.12e  cand %.12e  mcc %.6e  step %.3e  radius %.3e  -> accept %d active %d term %d\n", w,
                             bv.st[w].iter, x_cost_before, bv.red[w].cand_cost, bv.red[w].mcc, std::sqrt(bv.red[w].step2), radius_before,
                             bv.st[w].accept, bv.st[w].active, bv.st[w].term);
        }
        // accept: candidate -> current
        for (int w = 0; w < bv.n_win; ++w) {
            if (!bv.st[w].accept) continue;
            const WinDesc& wd = bv.win[w];
            std::memcpy(bv.pose + 7 * (size_t)wd.kf0, bv.pose_c + 7 * (size_t)wd.kf0, sizeof(double) * 7 * wd.n_kf);
            std::memcpy(bv.pdir + 3 * (size_t)wd.kf0, bv.pdir_c + 3 * (size_t)wd.kf0, sizeof(double) * 3 * wd.n_kf);
            std::memcpy(bv.pdist + wd.kf0, bv.pdist_c + wd.kf0, sizeof(double) * wd.n_kf);
            std::memcpy(bv.lm + 3 * (size_t)wd.lm0, bv.lm_c + 3 * (size_t)wd.lm0, sizeof(double) * 3 * wd.n_lm);
        }
    }

    void trim() override {
        for (size_t si = 0; si < pv.size(); ++si) {
            BatchView& v = pv[si];
            for (int b = 0; b < v.n_blk; ++b) {
                if (!owns(si, shard_P > 1 ? P.blk_owner[b] : 0)) continue;
                const int w = v.view_win[v.blk_view[b]];
                if (!v.win[w].do_trim) continue;
                for (int t = 0; t < kObsBlock; ++t) trim_residual_lane(v, b, t, plane_rep.data(), plane_dep.data());
            }
            for (int w = 0; w < v.n_win; ++w) {
                const WinDesc& wd = v.win[w];
                if (!wd.do_trim) continue;
                for (int l = 0; l < wd.n_lm; ++l)
                    if (owns_lm(si, wd.lm0 + l)) trim_max_lane(v, wd.lm0 + l, plane_rep.data(), plane_dep.data());
            }
        }
        exchange_trim();
        for (int w = 0; w < bv.n_win; ++w) {
            const WinDesc& wd = bv.win[w];
            if (!wd.do_trim) continue;
            std::vector<uint8_t> out(wd.n_lm, 0);
            for (int l = 0; l < wd.n_lm; ++l) {
                out[l] = trim_is_outlier(bv.trim_dep + wd.lm0, bv.lm_id + wd.lm0, wd.n_lm, l, c.depth_quantile, c.min_groups) ||
                         trim_is_outlier(bv.trim_rep + wd.lm0, bv.lm_id + wd.lm0, wd.n_lm, l, c.reprojection_quantile, c.min_groups);
            }
            for (int l = 0; l < wd.n_lm; ++l)
                if (out[l] && bv.lm_state[wd.lm0 + l]) {
                    bv.lm_state[wd.lm0 + l] = 0;
                    bv.st[w].n_trimmed++;
                }
        }
    }

    // ---- streaming solve (mirrors k_sched + the list-driven launches of limo_hip.hip:solve_streaming)
    // Trimming of ONE window (the lock-step trim() above handles every do_trim window at once); unsharded only.
    void trim_window(int w) {
        const WinDesc& wd = bv.win[w];
        if (!wd.do_trim) return;
        for (int b = wd.blk0; b < wd.blk0 + wd.n_blk; ++b)
            for (int t = 0; t < kObsBlock; ++t) trim_residual_lane(bv, b, t, plane_rep.data(), plane_dep.data());
        for (int l = 0; l < wd.n_lm; ++l) trim_max_lane(bv, wd.lm0 + l, plane_rep.data(), plane_dep.data());
        std::vector<uint8_t> out(wd.n_lm, 0);
        for (int l = 0; l < wd.n_lm; ++l)
            out[l] = trim_is_outlier(bv.trim_dep + wd.lm0, bv.lm_id + wd.lm0, wd.n_lm, l, c.depth_quantile, c.min_groups) ||
                     trim_is_outlier(bv.trim_rep + wd.lm0, bv.lm_id + wd.lm0, wd.n_lm, l, c.reprojection_quantile, c.min_groups);
        for (int l = 0; l < wd.n_lm; ++l)
            if (out[l] && bv.lm_state[wd.lm0 + l]) {
                bv.lm_state[wd.lm0 + l] = 0;
                bv.st[w].n_trimmed++;
            }
    }
    // Windows move through n_slots slots; per round: scheduler, trimming of the windows whose trimming solve ended,
    // one LM iteration of every window in a slot.  Returns the number of rounds.
    int run_streaming(int n_slots) {
        std::vector<int> slot(n_slots, -1);
        int cursor = 0, done = 0, rounds = 0;
        while (done < bv.n_win) {
            for (int s = 0; s < n_slots; ++s) {
                for (int tries = 0; tries < 2; ++tries) {
                    if (slot[s] < 0) {
                        if (cursor >= bv.n_win) break;
                        slot[s] = cursor++;
                        bv.st[slot[s]].phase = PH_IDLE;
                    }
                    const int r = sched_advance(bv.st[slot[s]], bv.win[slot[s]], c);
                    if (r == 2) {
                        ++done;
                        slot[s] = -1;
                        continue;
                    }
                    break;
                }
            }
            linearize();  // every kernel item looks at the window's own state: windows outside the slots are idle
            step();
            // the windows whose trimming solve ended are trimmed during the round (side stream on the GPU) and iterate
            // again from the next one
            for (int s = 0; s < n_slots; ++s)
                if (slot[s] >= 0 && bv.st[slot[s]].phase == PH_TRIM) {
                    trim_window(slot[s]);
                    sched_after_trim(bv.st[slot[s]], c);
                }
            if (++rounds > 100000) break;
        }
        return rounds;
    }

    // rank mode: make every rank hold every landmark (sum of "owned, else zero")
    void gather_landmarks() {
        if (shard_P == 1 || !cb) return;
        std::vector<double> send((size_t)3 * bv.TL, 0.0), recv((size_t)3 * bv.TL, 0.0);
        for (int gl = 0; gl < bv.TL; ++gl)
            if (owned_here(gl))
                for (int i = 0; i < 3; ++i) send[3 * (size_t)gl + i] = bv.lm[3 * (size_t)gl + i];
        cb(send.data(), recv.data(), (int64_t)send.size(), 0, cb_user);
        std::memcpy(bv.lm, recv.data(), sizeof(double) * recv.size());
    }
};

void fill_report(const EmuBatch& B, int w, limo_ba_report* r) {
    const WinState& s = B.bv.st[w];
    const WinDesc& d = B.P.win[w];  // (the consumer view of a sharded solve carries modified descriptors)
    std::memset(r, 0, sizeof(*r));
    r->termination = s.term;
    r->num_solves = s.acc_solves;
    r->iterations_total = s.acc_iters;
    r->iterations_final = s.last_iters;
    r->successful_steps = s.acc_success;
    r->n_depth_blocks = d.n_depth;
    r->n_repr_blocks = d.n_repr;
    r->n_gp_blocks = d.n_gp;
    r->n_trimmed_landmarks = s.n_trimmed;
    r->num_linearizations = s.acc_lin;
    r->initial_cost = s.first_initial_cost;
    r->final_cost = s.solve_final_cost;
}

// per window: indices (caller's order) of the landmarks the trimming rounds removed - what limo_ba_batch_trimmed reports
static thread_local std::vector<std::vector<int32_t>> g_last_trimmed;
void record_trimmed(const EmuBatch& B) {
    g_last_trimmed.assign(B.bv.n_win, {});
    for (int w = 0; w < B.bv.n_win; ++w) {
        const WinDesc& d = B.bv.win[w];
        for (int l = 0; l < d.n_lm; ++l)
            if (B.P.lm_state[d.lm0 + l] != 0 && B.bv.lm_state[d.lm0 + l] == 0) g_last_trimmed[w].push_back(B.bv.lm_id[d.lm0 + l]);
    }
}

void write_back(const EmuBatch& B, limo_ba_window* windows) {
    for (int w = 0; w < B.bv.n_win; ++w) {
        const WinDesc& d = B.bv.win[w];
        std::memcpy(windows[w].kf_pose, B.bv.pose + 7 * (size_t)d.kf0, sizeof(double) * 7 * d.n_kf);
        std::memcpy(windows[w].kf_plane_dir, B.bv.pdir + 3 * (size_t)d.kf0, sizeof(double) * 3 * d.n_kf);
        std::memcpy(windows[w].kf_plane_dist, B.bv.pdist + d.kf0, sizeof(double) * d.n_kf);
        for (int l = 0; l < d.n_lm; ++l)
            for (int i = 0; i < 3; ++i) windows[w].lm_pos[3 * (size_t)B.bv.lm_id[d.lm0 + l] + i] = B.bv.lm[3 * (size_t)(d.lm0 + l) + i];
    }
}

}  // namespace

extern "C" {

int emu_ba_solve_batch(int32_t n, limo_ba_window* windows, const limo_ba_options* o, limo_ba_report* reports,
                       int pose_only, const limo_speed_prior* prior) {
    EmuBatch B;
    std::string err;
    PackOptions po;
    po.pose_only = pose_only != 0;
    po.prior = prior;
    int rc = pack_windows(n, windows, *o, po, B.P, err);
    if (rc != LIMO_OK) return rc;
    B.c = make_consts(*o);
    B.alloc();
    run_schedule(B, *o);
    write_back(B, windows);
    record_trimmed(B);
    if (reports)
        for (int w = 0; w < n; ++w) fill_report(B, w, reports + w);
    return LIMO_OK;
}

// The same batch through the streaming schedule (n_slots windows in flight); the per-window results must not depend on it.
int emu_ba_solve_batch_streaming(int32_t n, limo_ba_window* windows, const limo_ba_options* o, limo_ba_report* reports, int n_slots) {
    EmuBatch B;
    std::string err;
    int rc = pack_windows(n, windows, *o, PackOptions(), B.P, err);
    if (rc != LIMO_OK) return rc;
    B.c = make_consts(*o);
    B.alloc();
    B.run_streaming(n_slots < 1 ? 1 : n_slots);
    write_back(B, windows);
    record_trimmed(B);
    if (reports)
        for (int w = 0; w < n; ++w) fill_report(B, w, reports + w);
    return LIMO_OK;
}

// Landmarks (caller's indices) of window w removed by trimming in the last emu_ba_solve_batch of this thread.
int emu_last_trimmed(int w, int32_t* out, int cap) {
    if (w < 0 || w >= (int)g_last_trimmed.size()) return -1;
    int n = 0;
    for (int32_t id : g_last_trimmed[w]) {
        if (n < cap && out) out[n] = id;
        ++n;
    }
    return n;
}

// Landmark-sharded solve of one window.  cb == NULL: n_shards virtual shards in this process (the
// exchange is a plain sum); else this process is shard `rank` of n_shards and cb is the all-reduce.
int emu_ba_solve_sharded(limo_ba_window* window, const limo_ba_options* o, int n_shards, int rank, int world,
                         emu_allreduce_fn cb, void* user, limo_ba_report* report) {
    EmuBatch B;
    std::string err;
    PackOptions po;
    po.shards = n_shards;
    int rc = pack_windows(1, window, *o, po, B.P, err);
    if (rc != LIMO_OK) return rc;
    B.c = make_consts(*o);
    if (cb) {  // shard s lives on rank s mod world
        for (int r = 0; r < n_shards; ++r)
            if (r % world == rank) B.local_shards.push_back(r);
        B.cb = cb;
        B.cb_user = user;
        B.world = world;
        B.rank = rank;
    }
    B.alloc();
    run_schedule(B, *o);
    B.gather_landmarks();
    write_back(B, window);
    if (report) fill_report(B, 0, report);
    return LIMO_OK;
}

int emu_ba_evaluate(const limo_ba_window* window, const limo_ba_options* o, int apply_loss, double* cost,
                    double* residuals, double* jac_pose, double* jac_lm, uint8_t* valid) {
    EmuBatch B;
    std::string err;
    PackOptions po;
    po.evaluate_only = true;
    int rc = pack_windows(1, window, *o, po, B.P, err);
    if (rc != LIMO_OK) return rc;
    B.c = make_consts(*o);
    B.alloc();
    double total = 0.0;
    for (int b = 0; b < B.bv.n_blk; ++b) {
        const int view = B.bv.blk_view[b];
        const double* cam = B.bv.view_cam + 16 * (int64_t)view;
        for (int t = 0; t < B.bv.blk_n[b]; ++t) {
            const int64_t o_ = B.bv.blk_obs0[b] + t;
            const int gl = B.bv.obs_lm[o_];
            // (the statements of k_evaluate: the view's constants, then eval_obs - with the IEEE operations on the host)
            double vl[40];
            view_consts_compute(cam, B.bv.pose + 7 * (int64_t)B.bv.view_kf[view], vl);
            EvalOut oo;
            const bool ok = eval_obs((const double*)vl, B.c, B.bv.lm + 3 * (int64_t)gl, B.bv.lm_weight[gl], B.bv.obs_u[o_], B.bv.obs_v[o_], B.bv.obs_d[o_],
                                     apply_loss != 0, oo);
            const int src = B.P.obs_src[o_];
            total += oo.cost;
            if (valid) valid[src] = ok ? 1 : 0;
            if (residuals) std::memcpy(residuals + 3 * (size_t)src, oo.r, sizeof(oo.r));
            if (jac_pose) std::memcpy(jac_pose + 18 * (size_t)src, oo.Jp, sizeof(oo.Jp));
            if (jac_lm) std::memcpy(jac_lm + 9 * (size_t)src, oo.Jl, sizeof(oo.Jl));
        }
    }
    if (cost) *cost = total;
    return LIMO_OK;
}

// limo_ba_evaluate_rows on the emulated pipeline: the same lane functions (gp_lane, reg_row_eval) in host loops, the same
// row assembly (kba_rows.hpp).
int emu_ba_evaluate_rows(const limo_ba_window* window, const limo_speed_prior* prior, int pose_only, const limo_ba_options* o, int32_t cap,
                         limo_ba_row* rows, int32_t* n_rows) {
    EmuBatch B;
    std::string err;
    PackOptions po;
    po.pose_only = pose_only != 0;
    po.prior = pose_only ? prior : nullptr;
    int rc = pack_windows(1, window, *o, po, B.P, err);
    if (rc != LIMO_OK) return rc;
    B.c = make_consts(*o);
    B.alloc();
    const WinDesc& wd = B.bv.win[0];
    for (int g = wd.gp0; g < wd.gp0 + wd.n_gp; ++g) gp_lane(B.bv, g, false, B.bv.gp_cost);
    const int n_reg = reg_row_count(wd);
    std::vector<RegRow> regs(std::max(1, n_reg));
    std::vector<int32_t> fixed(std::max(1, n_reg));
    for (int i = 0; i < n_reg; ++i) {
        int ac;
        reg_row_eval(wd, B.bv.cmask, B.bv.pose, B.bv.pdir, B.bv.pdist, i, true, regs[i], ac);
        fixed[i] = ac;
    }
    *n_rows = rows_from_linearisation(B.P, 0, B.bv.gp_r, B.bv.gp_F, B.bv.gp_E, B.bv.gp_cost, regs.data(), fixed.data(), cap, rows);
    return LIMO_OK;
}

// pack_windows alone (host-side flattening of a batch: what limo_ba_batch_create spends before the upload), milliseconds
double emu_pack_ms(int32_t n, const limo_ba_window* windows, const limo_ba_options* o) {
    PackedBatch P;
    std::string err;
    PackOptions po;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = pack_windows(n, windows, *o, po, P, err);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc == LIMO_OK ? ms : -1.0;
}

// The two statements of the rotation-tangent Jacobian M(q, p) side by side (kba_math.hpp): the chain-rule form
// rot_tangent_jac (what Problem::Evaluate's restatement uses) and the closed form -2 [Rh(q) p]_x the solve's kernels use.
void emu_rot_tangent_forms(const double* q, const double* p, double* m_chain, double* m_closed) {
    double R[9];
    kba::rot_tangent_jac(q, p, m_chain);
    kba::quat_R(q, R);
    kba::rot_tangent_from_R(R, kba::quat_norm2_minus_1(q), p, m_closed);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// Optional: export the product's C-ABI names on top of the emulation, so that C++ code written against
// include/limo_hip.h (the kba shim, tests/cpp/test_kba_shim.cpp) can be exercised in the CPU-only test tier.
// Built only into tests/cpp/_build/libkba_emu_abi.so.  NEVER linked into the product.
#ifdef KBA_EMU_EXPORT_ABI
struct limo_ctx {
    std::string err;
};
extern "C" {
int limo_abi_version(void) { return LIMO_ABI_VERSION; }
int limo_ctx_create(int, limo_ctx** out) {
    *out = new limo_ctx();
    return LIMO_OK;
}
void limo_ctx_destroy(limo_ctx* c) { delete c; }
void* limo_host_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
void limo_host_free(void* p) { std::free(p); }
int limo_ctx_set_stream(limo_ctx*, void*) { return LIMO_OK; }
const char* limo_last_error(const limo_ctx* c) { return c ? c->err.c_str() : ""; }
void limo_ba_default_options(limo_ba_options* o) {
    std::memset(o, 0, sizeof(*o));
    o->depth_thres = 0.16;
    o->reprojection_thres = 1.6;
    o->depth_quantile = 0.95;
    o->reprojection_quantile = 0.95;
    o->num_trim_rounds = 1;
    o->trim_solver_iterations = 2;
    o->min_landmarks_for_trimming = 100;
    o->minimum_number_residual_groups = 30;
    o->max_num_iterations = 100;
    o->max_solver_time_sec = -1.0;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->min_relative_decrease = 1e-3;
    o->max_num_consecutive_invalid_steps = 5;
    o->jacobi_scaling = 1;
}
int limo_ba_solve(limo_ctx*, limo_ba_window* w, const limo_ba_options* o, limo_ba_report* r) {
    return emu_ba_solve_batch(1, w, o, r, 0, nullptr);
}
int limo_ba_adjust_pose_only(limo_ctx*, limo_ba_window* w, const limo_speed_prior* p, const limo_ba_options* o,
                             limo_ba_report* r) {
    return emu_ba_solve_batch(1, w, o, r, 1, p);
}
#ifdef KBA_EMU_DEPTH
// depth assignment of the emulated backend = the CPU oracle's (oracle/depth_oracle.cpp, liboracle.so): test
// infrastructure standing in for test infrastructure, so that limo_stream can run its drive without a GPU
void oracle_depth_default_params(limo_depth_params* out);
int oracle_depth_estimate(const float* cloud_xyzi, size_t n_pts, const double* T_cam_lidar, double f, double cx, double cy, int32_t img_w,
                          int32_t img_h, const float* feat_uv, size_t n_feat, const uint8_t* feat_is_ground, const limo_depth_params* params,
                          float* depth_out);
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
#endif
}
#endif
