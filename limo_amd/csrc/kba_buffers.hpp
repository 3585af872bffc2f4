// kba_buffers.hpp — the list of buffers behind a BatchView, written once so the HIP library (hipMalloc /
// hipMemcpy) and the test emulator (malloc / memcpy) allocate exactly the same layout.
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>

#include "kba_pack.hpp"

namespace kba {

// f(void** slot, size_t bytes, const void* init /* may be null: zero-fill */)
template <typename F>
void for_each_buffer(const PackedBatch& P, BatchView& bv, F f) {
    bv.n_win = P.n_win;
    bv.TK = P.TK;
    bv.TL = P.TL;
    bv.TO = P.TO;
    bv.TV = P.TV;
    bv.TG = P.TG;
    bv.n_blk = P.n_blk;
    bv.n_lblk = P.n_lblk;
    bv.n_sblk = P.n_sblk;
    bv.Vmax = P.Vmax;
    bv.SO = P.SO;
    bv.SL = P.SL;
    bv.SG = P.SG;
    const size_t D = sizeof(double), I = sizeof(int32_t);
    const size_t TK = (size_t)(P.TK > 0 ? P.TK : 1), TL = (size_t)(P.TL > 0 ? P.TL : 1), TO = (size_t)(P.TO > 0 ? P.TO : 1),
                 TV = (size_t)(P.TV > 0 ? P.TV : 1), TG = (size_t)(P.TG > 0 ? P.TG : 1);
    const size_t NB = (size_t)(P.n_blk > 0 ? P.n_blk : 1), NL = (size_t)(P.n_lblk > 0 ? P.n_lblk : 1),
                 NS = (size_t)(P.n_sblk > 0 ? P.n_sblk : 1), NW = (size_t)P.n_win;
#define KBA_BUF(field, bytes, init) f((void**)&bv.field, (size_t)(bytes), (const void*)(init))
    KBA_BUF(win, NW * sizeof(WinDesc), P.win.data());
    KBA_BUF(st, NW * sizeof(WinState), nullptr);
    KBA_BUF(red, NW * sizeof(WinRed), nullptr);
    KBA_BUF(pose, TK * 7 * D, P.pose.data());
    KBA_BUF(pdir, TK * 3 * D, P.pdir.data());
    KBA_BUF(pdist, TK * D, P.pdist.data());
    KBA_BUF(lm, TL * 3 * D, P.lm.data());
    KBA_BUF(pose_c, TK * 7 * D, P.pose.data());
    KBA_BUF(pdir_c, TK * 3 * D, P.pdir.data());
    KBA_BUF(pdist_c, TK * D, P.pdist.data());
    KBA_BUF(lm_c, TL * 3 * D, P.lm.data());
    KBA_BUF(kf_win, TK * I, P.kf_win.data());
    KBA_BUF(kf_blk0, TK * I, P.kf_blk0.data());
    KBA_BUF(kf_nblk, TK * I, P.kf_nblk.data());
    KBA_BUF(kf_gp0, TK * I, P.kf_gp0.data());
    KBA_BUF(kf_ngp, TK * I, P.kf_ngp.data());
    KBA_BUF(cmask, TK * kCamSlots, P.cmask.data());
    KBA_BUF(cpresent, TK * kCamSlots, P.cpresent.data());
    KBA_BUF(cslot, TK * kCamSlots * I, P.cslot.data());
    KBA_BUF(lm_win, TL * I, P.lm_win.data());
    KBA_BUF(lm_id, TL * I, P.lm_id.data());
    KBA_BUF(lm_weight, TL * D, P.lm_weight.data());
    KBA_BUF(lm_state, TL, P.lm_state.data());
    KBA_BUF(lm_gp, TL * I, P.lm_gp.data());
    KBA_BUF(lm_slot, P.lm_slot.size() * I, P.lm_slot.data());
    KBA_BUF(view_kf, TV * I, P.view_kf.data());
    KBA_BUF(view_win, TV * I, P.view_win.data());
    KBA_BUF(view_cam, TV * 16 * D, P.view_cam.data());
    KBA_BUF(view_lin, TV * kViewLin * D, nullptr);
    KBA_BUF(view_lin_c, TV * kViewLin * D, nullptr);
    KBA_BUF(kf_dR, TK * 9 * D, nullptr);
    KBA_BUF(blk_view, NB * I, P.blk_view.data());
    KBA_BUF(blk_obs0, NB * I, P.blk_obs0.data());
    KBA_BUF(blk_n, NB * I, P.blk_n.data());
    KBA_BUF(obs_lm, TO * I, P.obs_lm.data());
    KBA_BUF(obs_u, TO * sizeof(float), P.obs_u.data());
    KBA_BUF(obs_v, TO * sizeof(float), P.obs_v.data());
    KBA_BUF(obs_d, TO * sizeof(float), P.obs_d.data());
    KBA_BUF(lblk_win, NL * I, P.lblk_win.data());
    KBA_BUF(lblk_lm0, NL * I, P.lblk_lm0.data());
    KBA_BUF(lblk_n, NL * I, P.lblk_n.data());
    KBA_BUF(sblk_win, NS * I, P.sblk_win.data());
    KBA_BUF(sblk_lm0, NS * I, P.sblk_lm0.data());
    KBA_BUF(sblk_n, NS * I, P.sblk_n.data());
    KBA_BUF(gp_lm, TG * I, P.gp_lm.data());
    KBA_BUF(gp_kf, TG * I, P.gp_kf.data());
    KBA_BUF(gp_w, TG * D, P.gp_w.data());
    KBA_BUF(gp_r, (size_t)P.SG * D, nullptr);
    KBA_BUF(gp_F, (size_t)P.SG * 10 * D, nullptr);
    KBA_BUF(gp_E, (size_t)P.SG * 3 * D, nullptr);
    KBA_BUF(gp_cost, TG * D, nullptr);
    KBA_BUF(gp_cost_c, TG * D, nullptr);
    KBA_BUF(obs_r, (size_t)(P.evaluate_only ? P.SO * 3 : 1) * D, nullptr);  // residual planes: limo_ba_evaluate only
    KBA_BUF(obs_c, (size_t)(P.evaluate_only ? 1 : P.SO * 4) * D, nullptr);
    KBA_BUF(obs_Jp, (size_t)(P.evaluate_only ? P.SO * 18 : 1) * D, nullptr);
    KBA_BUF(obs_Jl, (size_t)(P.evaluate_only ? P.SO * 9 : 1) * D, nullptr);
    KBA_BUF(lv_part, (size_t)(P.lvpart_total > 0 ? P.lvpart_total : 1) * D, nullptr);
    KBA_BUF(lblk_linfail, NL * D, nullptr);
    KBA_BUF(lm_V, (size_t)P.SL * 6 * D, nullptr);
    KBA_BUF(lm_g, (size_t)P.SL * 3 * D, nullptr);
    KBA_BUF(lm_scale, (size_t)P.SL * 3 * D, nullptr);
    KBA_BUF(lm_Li, (size_t)P.SL * 6 * D, nullptr);
    KBA_BUF(lm_t, (size_t)P.SL * 3 * D, nullptr);
    KBA_BUF(lblk_part, NL * 8 * D, nullptr);
    KBA_BUF(Hcc, (size_t)(P.hcc_total > 0 ? P.hcc_total : 1) * D, nullptr);
    KBA_BUF(gc, TK * kCamSlots * D, nullptr);
    KBA_BUF(scale_c, TK * kCamSlots * D, nullptr);
    KBA_BUF(yc, TK * kCamSlots * D, nullptr);
    KBA_BUF(delta_c, TK * kCamSlots * D, nullptr);
    KBA_BUF(S_part, (size_t)(P.spart_total > 0 ? P.spart_total : 1) * D, nullptr);
    KBA_BUF(S_red, (size_t)(P.sred_total > 0 ? P.sred_total : 1) * D, nullptr);
    KBA_BUF(cam_scratch, (size_t)(P.camscr_total > 0 ? P.camscr_total : 1) * D, nullptr);
    KBA_BUF(reg_cost, NW * 2 * D, nullptr);
    KBA_BUF(trim_rep, TL * D, nullptr);
    KBA_BUF(trim_dep, TL * D, nullptr);
    KBA_BUF(n_active, 8 * I, nullptr);
#undef KBA_BUF
}

// Per-workgroup partial arrays that cross from the landmark-side kernels (owned by ONE shard of a landmark-sharded
// solve) to the window-level kernels (replicated on every shard): the exchange set of SURVEY §8e.
//
// A sharded solve keeps them in ONE contiguous arena of doubles per view (the consumer view and every local shard), laid
// out so that each exchange point is ONE contiguous range: one k_sum_shards launch + one all-reduce per point, three per
// LM iteration (before k_cam_assemble / k_cam_solve / k_step_decide) and one per trimming round:
//   [ lv_part | lblk_linfail | gp_r | gp_F | gp_cost | gp_cost_c | lblk_part | S_red ]   [ trim_rep | trim_dep ]
//     point 1 ------------------------------------------------------------>|
//     point 4                                          |<-------------------|
//     point 2                                                      |<-------------->|          point 8 (own range)
// (gp_cost_c rides along at point 1: its values there are the previous iteration's, nobody reads them before point 4
// rewrites them).  Every entry has exactly one owner and is zero elsewhere, so the sums are exact in any order.
// S_part (the per-workgroup Schur slabs) is private to a shard and never exchanged: k_slab_reduce folds it into S_red.
struct ExchangeLayout {
    struct Slot {
        size_t member;  // offset of the pointer inside BatchView
        size_t off, count;  // doubles inside the arena
    };
    std::vector<Slot> slots;
    size_t total = 0;                    // doubles per arena
    size_t off[4] = {0, 0, 0, 0};        // range of exchange point 1, 2, 4, 8 (index = log2 of the point)
    size_t count[4] = {0, 0, 0, 0};
    size_t spart_count = 1;              // doubles of the private S_part
};
inline ExchangeLayout exchange_layout(const PackedBatch& P) {
    const size_t NL = (size_t)std::max(1, P.n_lblk), TG = (size_t)std::max(1, P.TG), TL = (size_t)std::max(1, P.TL);
    ExchangeLayout L;
    auto add = [&](size_t member, size_t count) {
        const size_t o = L.total;
        L.slots.push_back({member, o, count});
        L.total += (count + 31) / 32 * 32;  // 256-byte steps; the padding stays zero in every view
        return o;
    };
    const size_t o_lv = add(offsetof(BatchView, lv_part), (size_t)std::max<int64_t>(1, P.lvpart_total));
    add(offsetof(BatchView, lblk_linfail), NL);
    add(offsetof(BatchView, gp_r), (size_t)P.SG);
    add(offsetof(BatchView, gp_F), (size_t)P.SG * 10);
    add(offsetof(BatchView, gp_cost), TG);
    const size_t o_gcc = add(offsetof(BatchView, gp_cost_c), TG);
    const size_t o_lp = add(offsetof(BatchView, lblk_part), NL * 8);
    const size_t e_lp = L.total;
    add(offsetof(BatchView, S_red), (size_t)std::max<int64_t>(1, P.sred_total));
    const size_t e_sred = L.total;
    const size_t o_trim = add(offsetof(BatchView, trim_rep), TL);
    add(offsetof(BatchView, trim_dep), TL);
    L.off[0] = o_lv;
    L.count[0] = e_lp - o_lv;
    L.off[1] = o_lp;
    L.count[1] = e_sred - o_lp;
    L.off[2] = o_gcc;
    L.count[2] = e_lp - o_gcc;
    L.off[3] = o_trim;
    L.count[3] = L.total - o_trim;
    L.spart_count = (size_t)std::max<int64_t>(1, P.spart_total);
    return L;
}
inline int exchange_index(int point) { return point == 1 ? 0 : point == 2 ? 1 : point == 4 ? 2 : 3; }
// point `view` at the slices of `arena`
inline void exchange_bind(const ExchangeLayout& L, BatchView& view, double* arena) {
    for (const ExchangeLayout::Slot& sl : L.slots) *reinterpret_cast<double**>(reinterpret_cast<char*>(&view) + sl.member) = arena + sl.off;
}

}  // namespace kba
