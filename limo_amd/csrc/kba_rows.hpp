// kba_rows.hpp — host side of limo_ba_evaluate_rows: the rows struct of include/limo_hip.h from what the device functions
// gp_lane (planes gp_r / gp_F / gp_E, gp_cost) and reg_row_eval (RegRow) produced for window `w` of a packed batch.
// Shared by the HIP library (limo_hip.hip) and the CPU-tier emulator (tests/cpp/emu_pipeline.cpp).
#pragma once
#include <algorithm>
#include <cstring>

#include "kba_items.hpp"
#include "kba_pack.hpp"

namespace kba {

// Returns the number of rows of the problem; at most cap are written.
inline int rows_from_linearisation(const PackedBatch& P, int w, const double* gp_r, const double* gp_F, const double* gp_E, const double* gp_cost,
                                   const RegRow* regs, const int32_t* fixed, int32_t cap, limo_ba_row* rows) {
    const WinDesc& wd = P.win[w];
    const int n_reg = reg_row_count(wd);
    int n = 0;
    auto next = [&]() -> limo_ba_row* {
        limo_ba_row* r = n < cap ? rows + n : nullptr;
        ++n;
        if (r) {
            std::memset(r, 0, sizeof(*r));
            r->kf[0] = r->kf[1] = r->lm = -1;
        }
        return r;
    };
    for (int g = wd.gp0; g < wd.gp0 + wd.n_gp; ++g) {
        limo_ba_row* r = next();
        if (!r) continue;
        r->kind = LIMO_ROW_GROUND_HEIGHT;
        r->kf[0] = P.gp_kf[g] - wd.kf0;
        r->lm = P.lm_id[P.gp_lm[g]];
        r->r = gp_r[g];
        r->cost = gp_cost[g];
        for (int i = 0; i < 10; ++i) r->jac_kf[0][i] = gp_F[(size_t)i * P.SG + g];
        for (int i = 0; i < 3; ++i) r->jac_lm[i] = gp_E[(size_t)i * P.SG + g];
    }
    for (int i = 0; i < n_reg; ++i) {
        limo_ba_row* r = next();
        if (!r) continue;
        int j = i;  // the row numbering of kba_items.hpp:reg_row_eval
        if (wd.has_scale_reg && j == 0) {
            r->kind = LIMO_ROW_SCALE;
        } else {
            j -= wd.has_scale_reg ? 1 : 0;
            const int npair = wd.has_gp_reg ? wd.n_kf - 1 : 0, nglob = wd.has_gp_reg ? 3 * wd.n_kf : 0;
            if (j < 5 * npair) {
                const int sub = j % 5;
                r->kind = sub < 3 ? LIMO_ROW_NORMAL_DIFF : sub == 3 ? LIMO_ROW_DIST_DIFF : LIMO_ROW_PLANE_MOTION;
                r->sub = sub < 3 ? sub : 0;
                r->kf[0] = j / 5;  // (a row whose Jacobian was not requested still names its pair)
                r->kf[1] = j / 5 + 1;
            } else if (j < 5 * npair + nglob) {
                r->kind = LIMO_ROW_GLOBAL_NORMAL;
                r->sub = (j - 5 * npair) % 3;
                r->kf[0] = (j - 5 * npair) / 3;
            } else {
                r->kind = LIMO_ROW_SPEED;
                r->sub = j - 5 * npair - nglob;
                r->kf[0] = 0;
            }
        }
        if (r->kind == LIMO_ROW_SCALE) {
            r->kf[0] = 0;
            r->kf[1] = 1;
        }
        const RegRow& q = regs[i];
        r->fixed = fixed[i];
        r->r = q.r;
        for (int k = 0; k < q.n; ++k) {
            const int kf = q.col[k] / kCamSlots, slot = q.col[k] % kCamSlots;
            r->jac_kf[kf == r->kf[0] ? 0 : 1][slot] += q.val[k];
        }
    }
    // block costs on the sub == 0 rows: 1/2 |r_block|^2 (ScaledLoss(Trivial, w): the weight is inside r)
    for (int i = 0; i < std::min(n, (int)cap); ++i) {
        limo_ba_row& r = rows[i];
        if (r.kind == LIMO_ROW_GROUND_HEIGHT || r.sub != 0) continue;
        const int nb = (r.kind == LIMO_ROW_NORMAL_DIFF || r.kind == LIMO_ROW_GLOBAL_NORMAL || r.kind == LIMO_ROW_SPEED) ? 3 : 1;
        double s2 = 0.0;
        for (int k = 0; k < nb && i + k < std::min(n, (int)cap); ++k) s2 += rows[i + k].r * rows[i + k].r;
        r.cost = 0.5 * s2;
    }
    return n;
}

}  // namespace kba
