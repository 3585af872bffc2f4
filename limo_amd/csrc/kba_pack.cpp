// kba_pack.cpp — see kba_pack.hpp.
#include "kba_pack.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <limits>
#include <mutex>
#include <numeric>
#include <thread>
#include <pthread.h>
#include <unistd.h>
#include <exception>

#include "kba_items.hpp"

namespace kba {

// ---- the arena of the big pack arrays (kba_pack.hpp:PackArena)
namespace {
thread_local PackArena* t_pack_arena = nullptr;
std::mutex g_arena_ranges_m;
std::vector<std::pair<const char*, size_t>> g_arena_ranges;
std::atomic<int> g_arena_range_count{0};  // (pack_arena_owns is on every deallocation path: no lock while no arena exists)
}  // namespace
void pack_arena_lend(PackArena* a) { t_pack_arena = a; }
void pack_arena_register(const void* base, size_t cap, bool add) {
    std::lock_guard<std::mutex> lk(g_arena_ranges_m);
    if (add) {
        g_arena_ranges.emplace_back(static_cast<const char*>(base), cap);
    } else {
        for (size_t i = 0; i < g_arena_ranges.size(); ++i)
            if (g_arena_ranges[i].first == base) {
                g_arena_ranges.erase(g_arena_ranges.begin() + i);
                break;
            }
    }
    g_arena_range_count.store((int)g_arena_ranges.size(), std::memory_order_release);
}
bool pack_arena_owns(const void* p) {
    if (g_arena_range_count.load(std::memory_order_acquire) == 0) return false;
    std::lock_guard<std::mutex> lk(g_arena_ranges_m);
    const char* q = static_cast<const char*>(p);
    for (const auto& r : g_arena_ranges)
        if (q >= r.first && q < r.first + r.second) return true;
    return false;
}
void* pack_arena_take(size_t bytes) {
    PackArena* a = t_pack_arena;
    if (!a) return nullptr;
    const size_t need = (bytes + 255) / 256 * 256;
    a->wanted += need;
    if (!a->base || a->used + need > a->cap) return nullptr;
    void* p = a->base + a->used;
    a->used += need;
    return p;
}

namespace {
inline int64_t pad64(int64_t n) {
    return (n + 63) / 64 * 64;
}

// Host threads of the pack, kept between calls: a 1024-window create runs three parallel passes, and spawning 64 threads for each of
// them cost more than the passes' own work on a busy 256-thread host.  One parallel region at a time (a second caller - another host
// thread packing for another context - finds the pool busy and spawns its own threads as before); the caller takes part in the work.
//   * The caller only waits for the workers that JOINED the region: a worker that wakes after the items have run out (a loaded host)
//     finds the region closed and goes back to sleep - nobody waits for threads the scheduler has not run yet.
//   * An exception thrown by f on a worker is caught there and rethrown by run() on the caller (the first one wins).
//   * The pool is never destroyed (its threads sleep on a condition variable until the process ends).  A forked child gets a FRESH pool
//     object (pthread_atfork): the parent's threads do not exist there and its mutexes may have been held at the fork.
class PackPool {
public:
    // runs f(0 .. n - 1) on up to nt threads (the caller included); false = busy, nothing done
    bool run(unsigned nt, int n, const std::function<void(int)>& f) {
        std::unique_lock<std::mutex> region(region_, std::try_to_lock);
        if (!region.owns_lock()) return false;
        {
            std::lock_guard<std::mutex> lk(m_);
            while (threads_.size() + 1 < nt) {
                threads_.emplace_back([this] { worker(); });
                threads_.back().detach();
            }
            fn_ = &f;
            n_ = n;
            next_.store(0);
            want_ = std::min<size_t>(threads_.size(), nt > 0 ? nt - 1 : 0);
            joined_ = 0;
            open_ = true;
            error_ = nullptr;
            ++gen_;
        }
        cv_.notify_all();
        std::exception_ptr mine;
        try {
            for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) f(i);
        } catch (...) {
            mine = std::current_exception();
            next_.store(n);  // (the workers stop taking items)
        }
        std::unique_lock<std::mutex> lk(m_);
        open_ = false;  // late wakers skip this region
        done_.wait(lk, [this] { return joined_ == 0; });
        fn_ = nullptr;
        std::exception_ptr e = mine ? mine : error_;
        error_ = nullptr;
        lk.unlock();
        if (e) std::rethrow_exception(e);
        return true;
    }

private:
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* f = nullptr;
            int n = 0;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (!open_ || want_ == 0) continue;  // closed already, or enough workers: not this region
                --want_;
                ++joined_;
                f = fn_;
                n = n_;
            }
            std::exception_ptr err;
            try {
                for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) (*f)(i);
            } catch (...) {
                err = std::current_exception();
                next_.store(n);
            }
            {
                std::lock_guard<std::mutex> lk(m_);
                if (err && !error_) error_ = err;
                if (--joined_ == 0) done_.notify_one();
            }
        }
    }
    std::mutex region_, m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, joined_ = 0;
    bool open_ = false;
    size_t want_ = 0;
    uint64_t gen_ = 0;
    std::exception_ptr error_;
};
std::atomic<PackPool*> g_pack_pool{nullptr};
void pack_pool_after_fork() { g_pack_pool.store(nullptr); }  // (the parent's object is abandoned in the child: its threads are not there)
PackPool& pack_pool() {
    PackPool* p = g_pack_pool.load(std::memory_order_acquire);
    if (!p) {
        static std::mutex make;
        static bool hooked = false;
        // (in a forked child `make` may have been copied in the locked state only if the fork raced the very first call; a process that
        // forks while it is creating its first batch is outside what this pool supports)
        std::lock_guard<std::mutex> lk(make);
        p = g_pack_pool.load(std::memory_order_acquire);
        if (!p) {
            p = new PackPool();  // (leaked on purpose: see the class comment)
            if (!hooked) {
                pthread_atfork(nullptr, nullptr, pack_pool_after_fork);
                hooked = true;
            }
            g_pack_pool.store(p, std::memory_order_release);
        }
    }
    return *p;
}
}  // namespace

SolveConsts make_consts(const limo_ba_options& o) {
    SolveConsts c;
    std::memset(&c, 0, sizeof(c));
    c.a_rep = o.reprojection_thres;
    c.a_dep = o.depth_thres;
    c.inv_a_rep2 = 1.0 / (c.a_rep * c.a_rep);
    c.inv_a_dep2 = 1.0 / (c.a_dep * c.a_dep);
    c.function_tolerance = o.function_tolerance;
    c.gradient_tolerance = o.gradient_tolerance;
    c.parameter_tolerance = o.parameter_tolerance;
    c.initial_radius = o.initial_trust_region_radius;
    c.max_radius = o.max_trust_region_radius;
    c.min_radius = o.min_trust_region_radius;
    c.min_lm_diagonal = o.min_lm_diagonal;
    c.max_lm_diagonal = o.max_lm_diagonal;
    c.min_relative_decrease = o.min_relative_decrease;
    c.max_invalid = o.max_num_consecutive_invalid_steps;
    c.jacobi_scaling = o.jacobi_scaling;
    c.depth_quantile = o.depth_quantile;
    c.reprojection_quantile = o.reprojection_quantile;
    c.min_groups = o.minimum_number_residual_groups;
    c.schur_span = 1;
    c.schur_span_gp = 1;
    c.num_trim_rounds = o.num_trim_rounds;
    c.trim_iters = o.trim_solver_iterations;
    c.max_iters = o.max_num_iterations;
    if (const char* e = std::getenv("KBA_DEBUG_STAGE")) c.pad = std::atoi(e);  // profiling aid only
    return c;
}

int pack_windows(int32_t n, const limo_ba_window* windows, const limo_ba_options& opts, const PackOptions& po,
                 PackedBatch& P, std::string& err) {
    if (n <= 0 || !windows) {
        err = "no windows";
        return LIMO_ERR_INVALID;
    }
    const auto t_p0 = std::chrono::steady_clock::now();
    P = PackedBatch();
    P.n_win = n;
    P.n_shards = std::max(1, po.shards);
    P.evaluate_only = po.evaluate_only;
    P.win.resize(n);
    // ---- pass 1: sizes, views
    struct View {
        int kf, cam;
    };
    std::vector<std::vector<View>> views(n);
    std::vector<std::vector<int>> obs_view(n);
    std::vector<std::vector<int>> view_pad_off(n);  // evaluate-only batches: offset of every view's 64-aligned segment inside its window
    // host threads over the windows (used by both passes)
    auto for_windows = [&](auto&& f) {
        unsigned nt = std::thread::hardware_concurrency();
        // (1024 C2 windows on the 256-thread host of an MI355X box: fill 15.5 ms with 16 threads, 6.5 ms with 64, no gain beyond)
        nt = std::max(1u, std::min({nt, 64u, (unsigned)((n + 7) / 8)}));
        if (const char* e = std::getenv("KBA_PACK_THREADS")) nt = std::max(1, std::atoi(e));
        if (nt == 1) {
            for (int w = 0; w < n; ++w) f(w);
            return;
        }
        static const bool use_pool = !(std::getenv("KBA_PACK_POOL") && std::atoi(std::getenv("KBA_PACK_POOL")) == 0);
        if (use_pool && pack_pool().run(nt, n, std::function<void(int)>(f))) return;
        std::atomic<int> next{0};
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&] {
                for (int w = next.fetch_add(1); w < n; w = next.fetch_add(1)) f(w);
            });
        for (auto& th : pool) th.join();
    };
    std::vector<int> rc1(n, LIMO_OK);
    std::vector<const char*> err1(n, nullptr);
    for_windows([&](int w) {  // validation + the views (keyframe, camera) that carry observations
        const limo_ba_window& W = windows[w];
        auto fail = [&](int code, const char* msg) {
            rc1[w] = code;
            err1[w] = msg;
        };
        if (W.n_kf < 0 || W.n_lm < 0 || W.n_obs < 0 || W.n_cam < 0) return fail(LIMO_ERR_INVALID, "negative size");
        // The reference's NotEnoughKeyframesException counts ALL pushed keyframes (keyframes_.size() < 3,
        // bundle_adjuster_keyframes.cpp:630) - that check lives in the C++ shim.  The window only holds the ACTIVE
        // ones, and the reference solves happily with two (deactivateKeyframes(min_conn, 3, max) can leave exactly the
        // two newest, mono_lidar.cpp:249) or one of them: the scale / ground-plane regularisers simply need > 1
        // (:771, :891).  Only an empty window has nothing to solve.
        if (!po.pose_only && !po.evaluate_only && W.n_kf < 1)
            return fail(LIMO_ERR_NOT_ENOUGH_KF, "window without active keyframes");
        if (po.pose_only && W.n_kf != 1) return fail(LIMO_ERR_INVALID, "pose-only window must hold exactly one keyframe");
        if (W.n_kf > kMaxKf) return fail(LIMO_ERR_INVALID, "window has more keyframes than kMaxKf (20)");
        if ((W.n_kf && (!W.kf_pose || !W.kf_plane_dir || !W.kf_plane_dist || !W.kf_fixation)) ||
            (W.n_lm && (!W.lm_pos || !W.lm_weight || !W.lm_is_ground)) || (W.n_cam && !W.cam) ||
            (W.n_obs && (!W.obs_kf || !W.obs_lm || !W.obs_cam || !W.obs_u || !W.obs_v || !W.obs_d)))
            return fail(LIMO_ERR_INVALID, "null pointer in window");
        std::vector<int> vid((size_t)W.n_kf * std::max(1, W.n_cam), -1);
        for (int i = 0; i < W.n_obs; ++i) {
            const int k = W.obs_kf[i], l = W.obs_lm[i], c = W.obs_cam[i];
            if (k < 0 || k >= W.n_kf || l < 0 || l >= W.n_lm || c < 0 || c >= W.n_cam)
                return fail(LIMO_ERR_INVALID, "observation index out of range");
            vid[(size_t)k * W.n_cam + c] = 0;
        }
        int nv = 0;
        for (int k = 0; k < W.n_kf; ++k)
            for (int c = 0; c < W.n_cam; ++c)
                if (vid[(size_t)k * W.n_cam + c] == 0) {
                    vid[(size_t)k * W.n_cam + c] = nv++;
                    views[w].push_back({k, c});
                }
        if (nv > kMaxViews) return fail(LIMO_ERR_INVALID, "window has more than kMaxViews (64) (keyframe, camera) views with observations");
        obs_view[w].resize(W.n_obs);
        for (int i = 0; i < W.n_obs; ++i) obs_view[w][i] = vid[(size_t)W.obs_kf[i] * W.n_cam + W.obs_cam[i]];
    });
    for (int w = 0; w < n; ++w) {
        if (rc1[w] != LIMO_OK) {
            err = err1[w];
            return rc1[w];
        }
        const limo_ba_window& W = windows[w];
        const int nv = (int)views[w].size();
        P.Vmax = std::max(P.Vmax, nv);
        WinDesc& d = P.win[w];
        std::memset(&d, 0, sizeof(d));
        d.kf0 = P.TK;
        d.n_kf = W.n_kf;
        d.lm0 = P.TL;
        d.n_lm = W.n_lm;
        d.view0 = P.TV;
        d.n_view = nv;
        d.obs0 = P.TO;
        d.n_obs = W.n_obs;
        if (po.evaluate_only) {
            // evaluate-only batches: every view's observations start on a multiple of 64 (k_evaluate: a wave takes an ALIGNED range
            // of 64 observations of ONE view; landmark order inside a view as everywhere: the landmark gathers of a wave stay
            // inside ~1.5 KB).  The padding entries are inert observations (src -1, no depth) that no wave stores anything for.
            // (Tried and dropped: the view's depth observations first, so that a wave's depth rows are 64 aligned entries - the
            // depth planes then cost what their bytes say, but every wave's gather spans twice the landmarks and the pass as a
            // whole got 4 % slower: profiles/r06_experiment_evaluate_store_path.txt.)
            std::vector<int>& vo = view_pad_off[w];
            vo.assign(nv + 1, 0);
            for (int i = 0; i < W.n_obs; ++i) vo[obs_view[w][i] + 1]++;
            for (int v = 0; v < nv; ++v) vo[v + 1] = vo[v] + (int)pad64(vo[v + 1]);
            d.n_obs = vo[nv];
        }
        d.nc = W.n_kf * kCamSlots;
        d.nc_pad = (d.nc + 15) / 16 * 16;
        d.cam0 = d.kf0 * kCamSlots;
        d.pose_only = po.pose_only ? 1 : 0;
        P.TK += W.n_kf;
        P.TL += W.n_lm;
        P.TV += nv;
        P.TO += d.n_obs;
    }
    const auto t_p0b = std::chrono::steady_clock::now();
    P.SO = pad64(std::max(1, P.TO)) + kObsBlock;  // + a dump area: lanes past the end of a partial block store there (k_linearize)
    P.SL = pad64(std::max(1, P.TL));
    P.pose.resize((size_t)P.TK * 7);
    P.pdir.resize((size_t)P.TK * 3);
    P.pdist.resize(P.TK);
    P.kf_win.resize(P.TK);
    P.kf_blk0.assign(std::max(1, P.TK), 0);
    P.kf_nblk.assign(std::max(1, P.TK), 0);
    P.kf_gp0.assign(std::max(1, P.TK), 0);
    P.kf_ngp.assign(std::max(1, P.TK), 0);
    P.cmask.assign((size_t)P.TK * kCamSlots, 0);
    P.cpresent.assign((size_t)P.TK * kCamSlots, 0);
    P.cslot.assign((size_t)std::max(1, P.TK) * kCamSlots, -1);
    P.lm.resize((size_t)P.TL * 3);
    P.lm_win.resize(P.TL);
    P.lm_id.resize(P.TL);
    P.lm_gp.resize(P.TL);  // (-1 / 0 per window in pass 2: first touch by the thread that fills the window)
    P.lm_weight.resize(P.TL);
    P.lm_state.resize(P.TL);
    P.lm_slot.resize((size_t)std::max(1, P.Vmax) * P.SL);  // filled with -1 per window in pass 2, padding here
    for (int v = 0; v < std::max(1, P.Vmax); ++v)
        for (int64_t l = P.TL; l < P.SL; ++l) P.lm_slot[(size_t)v * P.SL + l] = -1;
    P.view_kf.resize(P.TV);
    P.view_win.resize(P.TV);
    P.view_cam.assign((size_t)P.TV * 16, 0.0);
    P.obs_lm.resize(P.TO);
    P.obs_u.resize(P.TO);
    P.obs_v.resize(P.TO);
    P.obs_d.resize(P.TO);
    P.obs_src.resize(P.TO);

    // ---- pass 2: fill.  Windows are independent: every window writes its own ranges of the fixed-size arrays and
    //      collects its variable-size lists (workgroup tables, ground-plane rows) locally with LOCAL indices; a serial
    //      merge concatenates them.  Host threads share the windows (the packing of a 1024-window batch is otherwise
    //      the longest step of limo_ba_batch_create).
    struct Local {
        std::vector<int32_t> blk_view, blk_obs0, blk_n, blk_owner, lblk_win, lblk_lm0, lblk_n, lblk_owner, sblk_win, sblk_lm0,
            sblk_n, sblk_owner, gp_lm, gp_kf, gp_owner;
        std::vector<double> gp_w;
        std::string err;
        int rc = LIMO_OK;
    };
    const auto t_p1 = std::chrono::steady_clock::now();
    std::vector<Local> locals(n);
    auto pack_one = [&](int w) {
        Local& L = locals[w];
        const limo_ba_window& W = windows[w];
        WinDesc& d = P.win[w];
        for (int v = 0; v < std::max(1, P.Vmax); ++v)
            std::fill_n(P.lm_slot.data() + (size_t)v * P.SL + d.lm0, W.n_lm, -1);
        std::memcpy(P.pose.data() + (size_t)d.kf0 * 7, W.kf_pose, sizeof(double) * 7 * W.n_kf);
        std::memcpy(P.pdir.data() + (size_t)d.kf0 * 3, W.kf_plane_dir, sizeof(double) * 3 * W.n_kf);
        std::memcpy(P.pdist.data() + d.kf0, W.kf_plane_dist, sizeof(double) * W.n_kf);
        for (int k = 0; k < W.n_kf; ++k) P.kf_win[d.kf0 + k] = w;
        // ---- ground-plane residuals, addGroundPlaneResiduals(10.), bundle_adjuster_keyframes.cpp:517-562
        //      (decided first: landmarks carrying a ground-plane row are packed LAST in their window, so a Schur
        //      tile of plain landmarks only touches the pose slots of the reduced camera system)
        struct GpTmp {
            int kf, lm;
            double w;
        };
        std::vector<GpTmp> gp_tmp;
        std::vector<uint8_t> has_gp(W.n_lm, 0);
        if (!po.pose_only && !po.evaluate_only) {
            const double weight = 10.;
            double Rk[kMaxKf][9];  // (one rotation matrix per keyframe, not per (landmark, keyframe) pair)
            for (int k = 0; k < W.n_kf; ++k) quat_R(W.kf_pose + 7 * k, Rk[k]);
            for (int l = 0; l < W.n_lm; ++l) {
                if (!W.lm_is_ground[l]) continue;
                double min_dist = std::numeric_limits<double>::max();
                int kf_id = -1;
                for (int k = 0; k < W.n_kf; ++k) {
                    if (W.kf_plane_dist[k] < -10.) continue;
                    const double* R = Rk[k];
                    double y[3];
                    mat3_vec(R, W.lm_pos + 3 * l, y);
                    y[0] += W.kf_pose[7 * k + 4];
                    y[1] += W.kf_pose[7 * k + 5];
                    y[2] += W.kf_pose[7 * k + 6];
                    const double dist = std::sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
                    if (dist < min_dist) {
                        min_dist = dist;
                        kf_id = k;
                    }
                }
                if (kf_id < 0) continue;
                const double max_valid_dist = 25.;
                if (min_dist < max_valid_dist) {
                    gp_tmp.push_back({kf_id, l, weight * (1. - min_dist / max_valid_dist)});
                    has_gp[l] = 1;
                }
            }
        }
        // packed landmark order: plain landmarks, then ground-plane landmarks; inside each class by owning shard
        // (landmark id mod n_shards), then input order.  seg[] = start of every run of one (class, shard):
        // with n_shards > 1 no workgroup straddles a run, so every workgroup has exactly one owner.
        const int NS = std::max(1, po.shards);
        std::vector<int> perm(W.n_lm), seg;
        {
            int nxt = 0;
            for (int pass = 0; pass < 2; ++pass)
                for (int r = 0; r < NS; ++r) {
                    const int before = nxt;
                    for (int l = 0; l < W.n_lm; ++l)
                        if (has_gp[l] == pass && l % NS == r) perm[l] = nxt++;
                    if (NS > 1 && nxt > before) seg.push_back(before);
                }
            if (seg.empty()) seg.push_back(0);
            seg.push_back(W.n_lm);
            d.lm_gp0 = d.lm0 + W.n_lm - (int)gp_tmp.size();
        }
        std::vector<int> seg_of(W.n_lm, 0);  // run index of a packed landmark
        for (size_t si = 0; si + 1 < seg.size(); ++si)
            for (int l = seg[si]; l < seg[si + 1]; ++l) seg_of[l] = (int)si;
        for (int l = 0; l < W.n_lm; ++l) {
            const int g = d.lm0 + perm[l];
            P.lm_win[g] = w;
            P.lm_id[g] = l;
            P.lm_weight[g] = W.lm_weight[l];
            for (int i = 0; i < 3; ++i) P.lm[(size_t)g * 3 + i] = W.lm_pos[3 * l + i];
        }
        for (int v = 0; v < d.n_view; ++v) {
            const int gv = d.view0 + v;
            P.view_kf[gv] = d.kf0 + views[w][v].kf;
            P.view_win[gv] = w;
            const double* cam = W.cam + 10 * views[w][v].cam;
            double* vc = P.view_cam.data() + (size_t)gv * 16;
            vc[0] = cam[0];
            vc[1] = cam[1];
            vc[2] = cam[2];
            quat_R(cam + 3, vc + 4);
            vc[13] = cam[7];
            vc[14] = cam[8];
            vc[15] = cam[9];
        }
        // observations sorted by (view, landmark).  A (view, landmark) pair occurs at most once in a valid window, so the order is
        // a placement: a dense table [view][packed landmark] -> observation, read out row by row (a comparison sort of the ~9000
        // observations of a C2 window through two indirections was the pack's hot spot: 0.5 ms per window and host thread).
        // Duplicates (rejected below with the same message as before) keep their input order behind the first one.
        std::vector<int> order(W.n_obs);
        {
            std::vector<int> first((size_t)std::max(1, d.n_view) * std::max(1, W.n_lm), -1);
            bool dup = false;
            for (int i = 0; i < W.n_obs; ++i) {
                int& cell = first[(size_t)obs_view[w][i] * W.n_lm + perm[W.obs_lm[i]]];
                if (cell < 0)
                    cell = i;
                else
                    dup = true;
            }
            if (!dup) {
                int nxt = 0;
                for (size_t c = 0; c < first.size(); ++c)
                    if (first[c] >= 0) order[nxt++] = first[c];
            } else {  // (an invalid window: the plain stable sort puts the duplicates side by side for the check below)
                std::iota(order.begin(), order.end(), 0);
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
                    if (obs_view[w][a] != obs_view[w][b]) return obs_view[w][a] < obs_view[w][b];
                    return perm[W.obs_lm[a]] < perm[W.obs_lm[b]];
                });
            }
        }
        d.blk0 = (int)L.blk_view.size();
        std::vector<int> lm_nobs(W.n_lm, 0);
        int depth_blocks = 0;
        int pos = 0;
        for (int v = 0; v < d.n_view; ++v) {
            const int start = pos;
            while (pos < W.n_obs && obs_view[w][order[pos]] == v) ++pos;
            auto packed_at = [&](int i) { return d.obs0 + (po.evaluate_only ? view_pad_off[w][v] + (i - start) : i); };
            for (int b0 = start; b0 < pos;) {
                int b1 = std::min(pos, b0 + kObsBlock);
                const int s0 = seg_of[perm[W.obs_lm[order[b0]]]];
                for (int i = b0 + 1; i < b1; ++i)
                    if (seg_of[perm[W.obs_lm[order[i]]]] != s0) {
                        b1 = i;
                        break;
                    }
                const int gkf = d.kf0 + views[w][v].kf;
                if (P.kf_nblk[gkf] == 0) P.kf_blk0[gkf] = (int)L.blk_view.size();
                P.kf_nblk[gkf]++;
                L.blk_view.push_back(d.view0 + v);
                L.blk_obs0.push_back(packed_at(b0));
                L.blk_n.push_back(b1 - b0);
                L.blk_owner.push_back(W.obs_lm[order[b0]] % NS);
                b0 = b1;
            }
            for (int i = start; i < pos; ++i) {
                const int src = order[i];
                const int o = packed_at(i);
                const int l = perm[W.obs_lm[src]];
                P.obs_lm[o] = d.lm0 + l;
                P.obs_u[o] = W.obs_u[src];
                P.obs_v[o] = W.obs_v[src];
                P.obs_d[o] = W.obs_d[src];
                P.obs_src[o] = src;
                int32_t& slot = P.lm_slot[(size_t)v * P.SL + d.lm0 + l];
                if (slot != -1) {
                    L.err = "duplicate (keyframe, landmark, camera) observation";
                    L.rc = LIMO_ERR_INVALID;
                    return;
                }
                slot = o;
                lm_nobs[l]++;
                if (W.obs_d[src] > 0.0f) depth_blocks++;
            }
            if (po.evaluate_only) {  // the padding behind the view's observations: inert entries
                for (int o = d.obs0 + view_pad_off[w][v] + (pos - start); o < d.obs0 + view_pad_off[w][v + 1]; ++o) {
                    P.obs_lm[o] = d.lm0;
                    P.obs_u[o] = P.obs_v[o] = 0.0f;
                    P.obs_d[o] = -1.0f;
                    P.obs_src[o] = -1;
                }
            }
        }
        d.n_blk = (int)L.blk_view.size() - d.blk0;
        d.n_depth = depth_blocks;
        d.n_repr = W.n_obs;
        for (int l = 0; l < W.n_lm; ++l) P.lm_state[d.lm0 + l] = lm_nobs[l] > 0 ? (po.pose_only ? 2 : 1) : 0;
        for (int l = 0; l < W.n_lm; ++l) P.lm_gp[d.lm0 + l] = -1;

        // ---- ground-plane rows
        d.gp0 = (int)L.gp_lm.size();
        {
            for (const GpTmp& g : gp_tmp)
                if (P.lm_state[d.lm0 + perm[g.lm]] == 0) P.lm_state[d.lm0 + perm[g.lm]] = 1;  // constrained by its gp block only
            // rows sorted by keyframe (stable in landmark order) so each keyframe owns a contiguous range
            std::stable_sort(gp_tmp.begin(), gp_tmp.end(), [](const GpTmp& a, const GpTmp& b) { return a.kf < b.kf; });
            for (const GpTmp& g : gp_tmp) {
                const int gi = (int)L.gp_lm.size();
                if (P.kf_ngp[d.kf0 + g.kf] == 0) P.kf_gp0[d.kf0 + g.kf] = gi;
                P.kf_ngp[d.kf0 + g.kf]++;
                P.lm_gp[d.lm0 + perm[g.lm]] = gi;
                L.gp_lm.push_back(d.lm0 + perm[g.lm]);
                L.gp_kf.push_back(d.kf0 + g.kf);
                L.gp_w.push_back(g.w);
                L.gp_owner.push_back(g.lm % NS);
            }
        }
        d.n_gp = (int)L.gp_lm.size() - d.gp0;

        // ---- which parameter blocks exist / are free (B9)
        uint8_t* present = P.cpresent.data() + (size_t)d.cam0;
        uint8_t* freem = P.cmask.data() + (size_t)d.cam0;
        auto set_block = [&](uint8_t* m, int k, int s0, int cnt) {
            for (int i = 0; i < cnt; ++i) m[k * kCamSlots + s0 + i] = 1;
        };
        if (po.evaluate_only) {
            for (int k = 0; k < W.n_kf; ++k) {
                set_block(present, k, 0, 6);
                set_block(freem, k, 0, 6);
            }
        } else if (po.pose_only) {
            set_block(present, 0, 0, 6);
            set_block(freem, 0, 0, 6);  // the new keyframe is not in active_keyframe_ids_, so never constant
            if (po.prior && po.prior->speed_weight > 0.0) {
                d.speed_w = po.prior->speed_weight;
                d.speed_dt = po.prior->dt_cur;
                for (int i = 0; i < 3; ++i) d.speed_vel[i] = po.prior->vel_prev[i];
                quat_R(po.prior->pose_before, d.speed_Rb);
                for (int i = 0; i < 3; ++i) d.speed_tb[i] = po.prior->pose_before[4 + i];
            }
        } else {
            // scale regularisation, :704-716 / :890-904
            const int n_depth = d.n_depth, n_gp = d.n_gp;
            double scale_w = -1.0;
            if (n_depth > 10 || n_gp > 10) {
                if (n_gp < 30) scale_w = 1000. / (static_cast<double>(n_depth + static_cast<double>(n_gp)));
            } else {
                scale_w = 1000.;
            }
            if (scale_w > 0.0 && W.n_kf > 1) {
                d.has_scale_reg = 1;
                d.scale_w = scale_w;
                double dd[3];
                rel_translation(W.kf_pose + 7, W.kf_pose, dd, nullptr, nullptr, false);
                d.scale_s0 = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
            }
            d.has_gp_reg = (n_gp > 0 && W.n_kf > 1) ? 1 : 0;  // :717-719, :771
            std::vector<int> kf_obs(W.n_kf, 0), kf_gp(W.n_kf, 0);
            for (int i = 0; i < W.n_obs; ++i) kf_obs[W.obs_kf[i]]++;
            for (int g = d.gp0; g < d.gp0 + d.n_gp; ++g) kf_gp[L.gp_kf[g] - d.kf0]++;
            for (int k = 0; k < W.n_kf; ++k) {
                const bool pose_in = kf_obs[k] > 0 || kf_gp[k] > 0 || d.has_gp_reg || (d.has_scale_reg && k < 2);
                const bool plane_in = kf_gp[k] > 0 || d.has_gp_reg;
                if (pose_in) set_block(present, k, 0, 6);
                if (plane_in) set_block(present, k, 6, 4);
                const bool fixed = W.kf_fixation[k] == LIMO_FIX_POSE;  // deactivatePoseParameters, :198-219
                if (pose_in && !fixed) set_block(freem, k, 0, 6);
                if (plane_in && !fixed) set_block(freem, k, 6, 3);
                if (plane_in && !fixed && !(n_depth < 10)) set_block(freem, k, 9, 1);  // :722-728
            }
        }
        // compact numbering of the free slots: every pose slot first, then the plane slots (kba_layout.hpp, WinDesc::nfq)
        d.nf = 0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = 0; i < d.nc; ++i)
                if (freem[i] && ((i % kCamSlots) >= 6) == (pass == 1)) P.cslot[(size_t)d.cam0 + i] = d.nf++;
            if (pass == 0) d.nfq = d.nf;
        }
        d.nf_pad = (d.nf + 1 + 15) / 16 * 16;  // + the rhs column
        d.do_trim = (W.n_lm > opts.min_landmarks_for_trimming) ? 1 : 0;  // :741 / :865

        // ---- workgroup tables
        d.lblk0 = (int)L.lblk_win.size();
        for (size_t si = 0; si + 1 < seg.size(); ++si)
            for (int l0 = seg[si]; l0 < seg[si + 1]; l0 += kBlock) {
                L.lblk_win.push_back(w);
                L.lblk_lm0.push_back(d.lm0 + l0);
                L.lblk_n.push_back(std::min(kBlock, seg[si + 1] - l0));
                L.lblk_owner.push_back(P.lm_id[d.lm0 + l0] % NS);
            }
        d.n_lblk = (int)L.lblk_win.size() - d.lblk0;
        d.sblk0 = (int)L.sblk_win.size();
        d.n_sblk_plain = 0;
        if (!po.pose_only && !po.evaluate_only && d.nf > 0) {
            const int per = kSchurLmPerBlock;
            // no Schur block straddles a shard run or the plain / ground-plane boundary
            const int n_plain = W.n_lm - (int)gp_tmp.size();
            std::vector<int> cuts(seg);
            cuts.push_back(n_plain);
            std::sort(cuts.begin(), cuts.end());
            cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
            for (size_t si = 0; si + 1 < cuts.size(); ++si)
                for (int l0 = cuts[si]; l0 < cuts[si + 1]; l0 += per) {
                    L.sblk_win.push_back(w);
                    L.sblk_lm0.push_back(d.lm0 + l0);
                    L.sblk_n.push_back(std::min(per, cuts[si + 1] - l0));
                    L.sblk_owner.push_back(P.lm_id[d.lm0 + l0] % NS);
                    if (l0 < n_plain) d.n_sblk_plain++;
                }
        }
        d.n_sblk = (int)L.sblk_win.size() - d.sblk0;
        // fast Schur variant (kba_kernels.hip): <= 4 keyframes with a free slot, one view per keyframe
        d.schur_fast = 1;
        d.n_fk = 0;
        for (int q = 0; q < 4; ++q) d.fk[q] = d.fk_view[q] = -1;
        for (int k = 0; k < W.n_kf; ++k) {
            int nv = 0, view = -1;
            for (int v = 0; v < d.n_view; ++v)
                if (views[w][v].kf == k) {
                    ++nv;
                    view = d.view0 + v;
                }
            if (nv > 1) d.schur_fast = 0;
            if (freem[k * kCamSlots] || freem[k * kCamSlots + 6]) {
                if (d.n_fk < 4) {
                    d.fk[d.n_fk] = k;
                    d.fk_view[d.n_fk] = view;
                }
                d.n_fk++;
            }
        }
        if (d.n_fk > 4) d.schur_fast = 0;
        d.n_view_fixed0 = 0;  // leading views whose keyframe has no free pose block
        while (d.n_view_fixed0 < d.n_view && !freem[views[w][d.n_view_fixed0].kf * kCamSlots]) d.n_view_fixed0++;
        if (po.evaluate_only) d.n_view_fixed0 = 0;
        };
    for_windows(pack_one);
    const auto t_p2 = std::chrono::steady_clock::now();
    // ---- merge: local lists -> global lists, local indices -> global indices.  A serial pass fixes every window's place in the
    //      global lists (running sums) and the sizes; the copies and the index shifts of the windows then run on the host threads
    //      (2000 landmark entries per window to shift: this was 4.5 ms of a 1024-window create as one loop).
    struct Base {
        int blk, lblk, sblk, gp;
    };
    std::vector<Base> base(n);
    {
        Base run{0, 0, 0, 0};
        for (int w = 0; w < n; ++w) {
            Local& L = locals[w];
            if (L.rc != LIMO_OK) {
                err = L.err;
                return L.rc;
            }
            base[w] = run;
            run.blk += (int)L.blk_view.size();
            run.lblk += (int)L.lblk_win.size();
            run.sblk += (int)L.sblk_win.size();
            run.gp += (int)L.gp_lm.size();
            WinDesc& d = P.win[w];
            d.blk0 += base[w].blk;
            d.lblk0 += base[w].lblk;
            d.sblk0 += base[w].sblk;
            d.gp0 += base[w].gp;
            d.hcc_off = P.hcc_total;
            P.hcc_total += (int64_t)d.nc * d.nc;
            d.spart_off = P.spart_total;
            P.spart_total += (int64_t)d.n_sblk * ((int64_t)d.nf_pad * d.nf_pad);
            d.sred_off = P.sred_total;
            if (P.n_shards > 1) P.sred_total += (int64_t)P.n_shards * schur_need_pad(d.nf);  // packed: upper triangle + rhs only
            d.xlv_off = P.xlv_total;
            P.xlv_total += (int64_t)d.n_view * kLinPartial;
            d.lvpart_off = P.lvpart_total;
            P.lvpart_total += (int64_t)d.n_lblk * d.n_view * kLinPartial;
            {   // camera system too large for LDS: scratch in global memory (kba_items.hpp:kCamLdsCapBytes)
                const int need = std::max(cam_assemble_scratch(d.nc, kBlock, d.n_view), cam_solve_scratch(d.nc, kBlock));
                d.cam_scr_off = -1;
                if (need * (int)sizeof(double) > kCamLdsCapBytes) {
                    d.cam_scr_off = P.camscr_total;
                    P.camscr_total += need;
                }
            }
        }
        for (auto* v : {&P.blk_view, &P.blk_obs0, &P.blk_n, &P.blk_owner}) v->resize(run.blk);
        for (auto* v : {&P.lblk_win, &P.lblk_lm0, &P.lblk_n, &P.lblk_owner}) v->resize(run.lblk);
        for (auto* v : {&P.sblk_win, &P.sblk_lm0, &P.sblk_n, &P.sblk_owner}) v->resize(run.sblk);
        for (auto* v : {&P.gp_lm, &P.gp_kf, &P.gp_owner}) v->resize(run.gp);
        P.gp_w.resize(run.gp);
    }
    for_windows([&](int w) {
        Local& L = locals[w];
        const WinDesc& d = P.win[w];
        const Base& bs = base[w];
        for (int k = d.kf0; k < d.kf0 + d.n_kf; ++k) {
            if (P.kf_nblk[k]) P.kf_blk0[k] += bs.blk;
            if (P.kf_ngp[k]) P.kf_gp0[k] += bs.gp;
        }
        for (int l = d.lm0; l < d.lm0 + d.n_lm; ++l)
            if (P.lm_gp[l] >= 0) P.lm_gp[l] += bs.gp;
        auto put = [](auto& dst, const auto& src, int at) { std::copy(src.begin(), src.end(), dst.begin() + at); };
        put(P.blk_view, L.blk_view, bs.blk);
        put(P.blk_obs0, L.blk_obs0, bs.blk);
        put(P.blk_n, L.blk_n, bs.blk);
        put(P.blk_owner, L.blk_owner, bs.blk);
        put(P.lblk_win, L.lblk_win, bs.lblk);
        put(P.lblk_lm0, L.lblk_lm0, bs.lblk);
        put(P.lblk_n, L.lblk_n, bs.lblk);
        put(P.lblk_owner, L.lblk_owner, bs.lblk);
        put(P.sblk_win, L.sblk_win, bs.sblk);
        put(P.sblk_lm0, L.sblk_lm0, bs.sblk);
        put(P.sblk_n, L.sblk_n, bs.sblk);
        put(P.sblk_owner, L.sblk_owner, bs.sblk);
        put(P.gp_lm, L.gp_lm, bs.gp);
        put(P.gp_kf, L.gp_kf, bs.gp);
        put(P.gp_w, L.gp_w, bs.gp);
        put(P.gp_owner, L.gp_owner, bs.gp);
        L = Local();
    });
    if (std::getenv("KBA_PACK_TRACE"))
        std::fprintf(stderr, "[kba] pack: views + sizes %.1f ms, arrays %.1f ms, fill %.1f ms, merge %.1f ms\n", std::chrono::duration<double, std::milli>(t_p0b - t_p0).count(),
                     std::chrono::duration<double, std::milli>(t_p1 - t_p0b).count(), std::chrono::duration<double, std::milli>(t_p2 - t_p1).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_p2).count());
    P.TG = (int)P.gp_lm.size();
    P.SG = pad64(std::max(1, P.TG));
    P.n_blk = (int)P.blk_view.size();
    P.n_lblk = (int)P.lblk_win.size();
    P.n_sblk = (int)P.sblk_win.size();
    if (P.evaluate_only) {
        // The materialised pass (k_evaluate) writes the rows that EXIST: the depth rows go into COMPACT planes over the depth
        // observations only, indexed by the observation's rank among them in packed order (obs_rank).  Its work items are
        // wave-sized: an aligned range of 64 observations of one observation block (= one view), each knowing the rank of its
        // first depth observation.
        P.echunk.clear();
        P.obs_rank.assign((size_t)std::max(1, P.TO), -1);
        std::vector<int32_t> before((size_t)P.TO + 1, 0);  // depth observations in front of observation o
        for (int o = 0; o < P.TO; ++o) {
            const bool dep = P.obs_d[o] > 0.0f;
            if (dep) P.obs_rank[o] = before[o];
            before[o + 1] = before[o] + (dep ? 1 : 0);
        }
        P.TD = before[P.TO];
        P.SD = pad64(std::max(1, P.TD)) + 64;
        for (int b = 0; b < P.n_blk; ++b) {
            const int o0 = P.blk_obs0[b], o1 = o0 + P.blk_n[b];
            for (int a = o0; a < o1; a += 64) {  // (o0 is a multiple of 64: views are aligned, blocks are 1024 apart inside one)
                EvalChunk ch;
                ch.base = a;
                ch.view = P.blk_view[b];
                ch.o0 = o0;
                ch.o1 = o1;
                ch.dep0 = before[a];
                ch.pad = 0;
                P.echunk.push_back(ch);
            }
        }
        P.n_echunk = (int)P.echunk.size();
    }
    return LIMO_OK;
}

void run_schedule(Executor& ex, const limo_ba_options& o) {
    using Clock = std::chrono::steady_clock;
    auto run_solve = [&](int max_iter, int select) {
        ex.solve_init(max_iter, select);
        const auto t0 = Clock::now();
        for (;;) {
            ex.linearize();
            if (ex.active_count() == 0) break;
            if (o.max_solver_time_sec > 0.0 &&
                std::chrono::duration<double>(Clock::now() - t0).count() >= o.max_solver_time_sec) {
                ex.expire(0);
                break;
            }
            ex.step();
        }
    };
    for (int r = 0; r < o.num_trim_rounds; ++r) {
        run_solve(o.trim_solver_iterations, 1);
        run_solve(3 * o.trim_solver_iterations, 2);
        ex.trim();
    }
    run_solve(o.max_num_iterations, 0);
}

}  // namespace kba
