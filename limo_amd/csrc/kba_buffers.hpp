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
    bv.SD = P.SD;
    bv.echunk = nullptr;
    bv.n_echunk = P.n_echunk;
    bv.pad_e = 0;
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
    // evaluate-only batches (limo_ba_evaluate): rows u, v over the observations + the depth row over the depth observations
    KBA_BUF(obs_r, (size_t)(P.evaluate_only ? P.SO * 2 + P.SD : 1) * D, nullptr);
    KBA_BUF(obs_c, (size_t)(P.evaluate_only ? 1 : P.SO * 2) * D, nullptr);
    KBA_BUF(obs_Jp, (size_t)(P.evaluate_only ? P.SO * 12 + P.SD * 6 : 1) * D, nullptr);
    KBA_BUF(obs_Jl, (size_t)(P.evaluate_only ? P.SO * 6 + P.SD * 3 : 1) * D, nullptr);
    if (P.evaluate_only) KBA_BUF(echunk, (size_t)std::max(1, P.n_echunk) * sizeof(EvalChunk), P.echunk.empty() ? nullptr : P.echunk.data());
    KBA_BUF(lv_part, (size_t)(P.lvpart_total > 0 ? P.lvpart_total : 1) * D, nullptr);
    KBA_BUF(lblk_linfail, NL * D, nullptr);
    KBA_BUF(lm_V, (size_t)P.SL * 6 * D, nullptr);
    KBA_BUF(lm_g, (size_t)P.SL * 3 * D, nullptr);
    KBA_BUF(lm_scale, (size_t)P.SL * 3 * D, nullptr);
    KBA_BUF(lm_Li, (size_t)P.SL * 6 * D, nullptr);
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
    if (P.n_shards > 1) {  // landmark-sharded batches: who owns a landmark workgroup / a ground-plane row (shard_reduce_*)
        KBA_BUF(lblk_owner, NL * I, P.lblk_owner.data());
        KBA_BUF(gp_owner, TG * I, P.gp_owner.data());
    }
#undef KBA_BUF
}

// Landmark-sharded solve: what crosses from the landmark-side kernels (owned by ONE shard) to the window-level
// kernels (replicated on every shard).
//
// A shard folds its per-workgroup partial arrays into ONE contiguous block of doubles (kba_items.hpp:shard_reduce_lin / _step,
// slab_reduce_entry) - camera-side sums per view, ground-plane rows as F^T F | F^T r per keyframe, the landmark-side scalars,
// and the entries of [S | rhs] the camera solve reads (upper triangle + rhs):
//     block = [ x_lv | x_lf | x_gp | x_gc | x_gcc | x_lb | S ]
//               point A1 (first linearisation of a solve: the camera assembly defines the Jacobi scale the Schur
//               |<------------------------------------>|    complement needs)            point A2 |<->|
//               point A  = the whole block: every other iteration, ONE exchange before camera assembly + camera solve
//               point B                    |<--------->|    before the step decision (9 doubles per window)
// An exchange is an ALL-GATHER of the blocks (one per shard, 41 KB at a 10-keyframe / 8000-landmark window) followed by
// unpack_entry: the consumer view holds the P contributions side by side, in shard order, behind the ordinary members
// (lv_part, lblk_linfail, gp_cost, gp_cost_c, lblk_part, S_red) and gp_red, with a WinDesc that says "P workgroups, P rows" -
// cam_assemble / cam_solve / reduce_step add them in shard order.  Nothing is summed on the wire, so the result does not
// depend on how the P shards are spread over ranks (P virtual shards on one GPU = P ranks, bit for bit).
// The trimming step exchanges the per-landmark residual maxima (one owner per entry, zero elsewhere: an exact sum) once per
// trimming round.  S_part (the per-workgroup Schur slabs) is private to a shard and never exchanged.
// (struct ExchangeLayout: kba_layout.hpp - the kernels that unpack an exchange read it too)
inline ExchangeLayout exchange_layout(const PackedBatch& P) {
    ExchangeLayout L;
    auto pad = [](size_t n) { return (n + 31) / 32 * 32; };  // 256-byte steps
    L.P = P.n_shards;
    L.n_win = P.n_win;
    L.TK = P.TK;
    L.n_lv = pad((size_t)std::max<int64_t>(1, P.xlv_total));
    L.n_w = pad((size_t)std::max(1, P.n_win));
    L.n_gp = pad((size_t)std::max(1, P.TK) * kGpRed);
    L.n_lb = pad((size_t)std::max(1, P.n_win) * 8);
    L.n_S = pad((size_t)std::max<int64_t>(1, P.sred_total / std::max(1, P.n_shards)));
    size_t o = 0;
    L.b_lv = o, o += L.n_lv;
    L.b_lf = o, o += L.n_w;
    L.b_gp = o, o += L.n_gp;
    L.b_gc = o, o += L.n_w;
    L.b_gcc = o, o += L.n_w;
    L.b_lb = o, o += L.n_lb;
    L.b_S = o, o += L.n_S;
    L.b_total = o;
    const size_t Pn = (size_t)L.P;
    o = 0;
    L.c_lv = o, o += Pn * L.n_lv;
    L.c_lf = o, o += Pn * L.n_w;
    L.c_gp = o, o += Pn * L.n_gp;
    L.c_gc = o, o += Pn * L.n_w;
    L.c_gcc = o, o += Pn * L.n_w;
    L.c_lb = o, o += Pn * L.n_lb;
    L.c_S = o, o += Pn * L.n_S;
    L.c_total = o;
    L.trim_count = 2 * (size_t)std::max(1, P.TL);
    L.spart_count = (size_t)std::max<int64_t>(1, P.spart_total);
    return L;
}
// producer view of one shard: x_* and S_red point into the shard's block, trim_rep / trim_dep into its trimming arena
inline void exchange_bind_producer(const ExchangeLayout& L, BatchView& v, double* block, double* trim) {
    v.x_lv = block + L.b_lv;
    v.x_lf = block + L.b_lf;
    v.x_gp = block + L.b_gp;
    v.x_gc = block + L.b_gc;
    v.x_gcc = block + L.b_gcc;
    v.x_lb = block + L.b_lb;
    v.S_red = block + L.b_S;
    v.trim_rep = trim;
    v.trim_dep = trim + L.trim_count / 2;
}
// consumer view: the P contributions behind the ordinary members; `win_c` = the windows' descriptors with the consumer's
// lblk0 / n_lblk / lvpart_off / gp0 / n_gp (exchange_consumer_windows)
inline void exchange_bind_consumer(const ExchangeLayout& L, BatchView& v, double* arena, double* trim, const WinDesc* win_c) {
    v.lv_part = arena + L.c_lv;
    v.lblk_linfail = arena + L.c_lf;
    v.gp_red = arena + L.c_gp;
    v.gp_red_P = L.P;
    v.gp_cost = arena + L.c_gc;
    v.gp_cost_c = arena + L.c_gcc;
    v.lblk_part = arena + L.c_lb;
    v.S_red = arena + L.c_S;
    v.trim_rep = trim;
    v.trim_dep = trim + L.trim_count / 2;
    v.win = win_c;
}
inline std::vector<WinDesc> exchange_consumer_windows(const PackedBatch& P) {
    std::vector<WinDesc> out(P.win.begin(), P.win.end());
    for (int w = 0; w < P.n_win; ++w) {
        WinDesc& d = out[w];
        d.lblk0 = w * P.n_shards;
        d.n_lblk = P.n_shards;
        d.lvpart_off = (int64_t)P.n_shards * d.xlv_off;
        d.gp0 = w * P.n_shards;
        d.n_gp = d.n_gp > 0 ? P.n_shards : 0;
    }
    return out;
}

}  // namespace kba
