// limo_hip.hip — the C-ABI of include/limo_hip.h on top of the gfx950 kernels (kba_kernels.hip).
// Host side: packing (kba_pack.cpp), device allocation, launch sequencing, result download.
// There is NO CPU fallback here: without a HIP device limo_ctx_create fails with LIMO_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "kba_kernels.hip"

#include "kba_buffers.hpp"
#include "kba_rows.hpp"

using namespace kba;

#include "limo_ctx.hpp"

#define HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                     \
            return LIMO_ERR_RUNTIME;                                                             \
        }                                                                                        \
    } while (0)

namespace {

inline int cdiv(int64_t a, int64_t b) {
    return (int)((a + b - 1) / b);
}

struct EventPair {
    hipEvent_t a, b;
    int kernel;  // LIMO_KERNEL_*
};

}  // namespace

struct limo_ba_batch : Executor {
    limo_ctx* ctx = nullptr;
    PackedBatch P;
    bool holds_pack_arena = false;  // P's big arrays live in ctx->pack_arena
    BatchView bv;
    SolveConsts c;
    limo_ba_options opts;
    std::vector<std::pair<void*, size_t>> allocs;  // device blocks of this batch (returned to the context's pool)
    double *d_plane_rep = nullptr, *d_plane_dep = nullptr;
    double *d_pose0 = nullptr, *d_pdir0 = nullptr, *d_pdist0 = nullptr, *d_lm0 = nullptr;
    uint8_t* d_lm_state0 = nullptr;
    std::vector<WinState> st0;  // reset image of the LM state
    int32_t* h_active = nullptr;  // pinned, 4 slots: written by k_cam_assemble, read by the host one iteration later
    int32_t* d_h_active = nullptr;  // the same words as the device sees them
    size_t h_flags_bytes = 0;
    hipEvent_t act_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int it_no = 0;
    // worklists of the windows that still iterate (nullptr = every workgroup): rebuilt on the host whenever the
    // active set has halved, so late LM iterations only launch the workgroups that have work
    int32_t *d_wl_blk = nullptr, *d_wl_lblk = nullptr, *d_wl_sblk = nullptr, *d_wl_win = nullptr, *d_flags = nullptr;
    int32_t* d_wl_sblk_part = nullptr;                          // Schur worklist of a re-batched active set
    struct FullSblk {  // Schur worklist of every window for one (span, span_gp), built on first use
        int32_t* d = nullptr;
        int n = 0, n_plain = 0, n_fgp = 0;
    };
    std::map<std::pair<int, int>, FullSblk> wl_sblk_full;
    // Schur worklists are ordered [plain groups of fast windows | ground-plane groups of fast windows | generic windows]
    int n_wl_sblk_plain = 0, n_wl_sblk_fgp = 0;
    const void* schur_fn_plain = nullptr;
    int plain_lds_bytes = 0;
    const void* schur_fn_leangp = nullptr;
    int leangp_lds_bytes = 0;
    int schur_vp = 1, schur_vg = 1;  // the same choice as template arguments of the one-launch solve (k_solve_coop)
    bool schur_pair_ok = false;      // the batch's variants are <2, false> / <3, true>: k_schur_lean_pair<2, 3> exists for them
    static constexpr int kSchurPairBound = 384;  // windows in flight in a slot group up to which a round launches the pair kernel (192: -0.7 %, 768: -9 % at 1024 windows)
    std::vector<uint8_t> win_fast;  // per window: k_schur<.., true> applies (<= 4 keyframes with free slots, one view each)
    int avg_sblk = 0;
    int32_t* h_flags = nullptr;  // pinned
    std::vector<int32_t> h_wl;
    bool use_wl = false;
    int n_wl_blk = 0, n_wl_lblk = 0, n_wl_sblk = 0, n_wl_win = 0, listed = 0;
    int max_nc = 0, asm_bytes = 0, solve_bytes = 0, trim_bytes = 0;
    const void* schur_fn_gen = nullptr;  // k_schur_wide<NPW> for the windows outside the fast class (chosen per window: results
                                         // do not depend on what else is in the batch)
    int wide_lds_bytes = 0;
    int rc = LIMO_OK;
    // ---- streaming solve (device-side scheduler k_sched): windows move through n_slots slots, a finished window is
    // replaced by the next pending one, so every launch round works on a full set (kba_kernels.hip:k_sched)
    int n_slots = 0;
    int32_t* d_sched_ctl = nullptr;   // [0] cursor over the batch's windows, [1] windows finished (shared by the groups)
    // ---- landmark sharding.  shard_P == 1: everything below is inert (pv = {bv}).
    // Every shard holds the same global layout and owns the observation / landmark / Schur workgroups of its
    // landmarks (rank lists); the per-workgroup partial arrays it produces live in its own "producer view" pv[i] and
    // are summed into the consumer view bv before each window-level kernel: by RCCL all-reduce when the shards are
    // ranks (one local shard), by k_sum_shards when all shards are virtual on this GPU.
    int shard_P = 1, shard_rank = 0;
    bool shard_virtual = true;
    std::vector<int> local_shards;
    std::vector<BatchView> pv;
    ExchangeLayout xl;
    double* arena_c = nullptr;          // consumer view: the P contributions side by side (kba_buffers.hpp:exchange_layout)
    std::vector<double*> blocks;        // block of local shard i (what it contributes per LM iteration)
    std::vector<double*> trims;         // trimming arenas: [0] consumer, [1 + i] local shard i
    double* d_gather = nullptr;         // [world][block] staging of the all-gather
    WinDesc* d_win_orig = nullptr;      // the batch's own window descriptors (bv.win of the consumer view is modified)
    bool first_lin = false, assemble_pending = false;
    int64_t n_exchanges = 0, exchange_bytes = 0;  // all-reduce calls / bytes of this batch so far (limo_ba_batch_exchange_stats)
    struct RankLists {
        int32_t *full_blk = nullptr, *full_lblk = nullptr, *full_sblk = nullptr;  // every window listed
        int32_t *act_blk = nullptr, *act_lblk = nullptr, *act_sblk = nullptr;     // re-batched active set
        int n_full_blk = 0, n_full_lblk = 0, n_full_sblk = 0, n_full_sblk_plain = 0, n_full_sblk_fgp = 0, n_sblk_plain = 0, n_sblk_fgp = 0;
        const int32_t *blk = nullptr, *lblk = nullptr, *sblk = nullptr;           // lists in use
        int n_blk = 0, n_lblk = 0, n_sblk = 0;
    };
    std::vector<RankLists> rl;
    double* d_lm_tmp = nullptr;
    // kernel timing (linearize) via HIP events on the batch's stream
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    double k_ms_acc[2] = {0.0, 0.0};      // indexed by LIMO_KERNEL_LINEARIZE / LIMO_KERNEL_SCHUR
    int64_t k_launches[2] = {0, 0};
    hipEvent_t ev_total_a = nullptr, ev_total_b = nullptr;
    double total_ms_acc = 0.0;
    double last_solve_sec = 0.0;

    ~limo_ba_batch() override {
        // (the arrays of P that live in the context's pinned pack arena are not freed one by one - kba_pack.hpp:PackArena -, the arena
        // is free for the next batch from here on; whatever of this batch's upload is still in flight only feeds device blocks that
        // go back to the pool below and are rewritten, in stream order, by their next user)
        if (holds_pack_arena) ctx->pack_arena_busy = false;
        for (auto& a : allocs) ctx->pool_free(a.first, a.second);
        if (h_active) ctx->host_free(h_active, 64);
        if (h_flags) ctx->host_free(h_flags, h_flags_bytes);
        stream_teardown();
        for (auto& e : ev_pool) {
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        for (auto& e : act_ev)
            if (e) (void)hipEventDestroy(e);
        if (ev_total_a) (void)hipEventDestroy(ev_total_a);
        if (ev_total_b) (void)hipEventDestroy(ev_total_b);
    }

    int dmalloc(void** p, size_t bytes) {
        bytes = bytes ? bytes : 8;
        HIP_TRY(ctx, ctx->pool_alloc(p, bytes));
        allocs.push_back({*p, bytes});
        // debugging aid: KBA_POISON=1 fills every block with NaN bytes first, so that a kernel reading a word nobody
        // wrote shows up as a wrong result instead of depending on what the block held before
        static const bool poison = std::getenv("KBA_POISON") != nullptr;
        if (poison) {
            HIP_TRY(ctx, hipMemsetAsync(*p, 0xFF, bytes, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        return LIMO_OK;
    }

    // Schur worklist over `windows` (all of them when null): first block of every group of `span` blocks of one class,
    // ordered [plain groups of fast windows | ground-plane groups of fast windows | groups of generic windows];
    // owner >= 0 keeps the blocks of that shard only.
    void build_sblk_list(const std::vector<int32_t>* windows, int span, int span_gp, int owner, std::vector<int32_t>& v, int& n_plain, int& n_fgp) const {
        v.clear();
        n_plain = n_fgp = 0;
        const int nw = windows ? (int)windows->size() : P.n_win;
        for (int cls = 0; cls < 3; ++cls) {
            for (int q = 0; q < nw; ++q) {
                const int w = windows ? (*windows)[q] : q;
                const WinDesc& d = P.win[w];
                if ((cls < 2) != (win_fast[w] != 0)) continue;
                auto groups = [&](int i0, int i1, int sp) {
                    for (int i = i0; i < i1; i += sp)
                        if (owner < 0 || P.sblk_owner[d.sblk0 + i] == owner) v.push_back(d.sblk0 + i);
                };
                if (cls != 1) groups(0, d.n_sblk_plain, span);
                if (cls != 0) groups(d.n_sblk_plain, d.n_sblk, span_gp);
            }
            if (cls == 0) n_plain = (int)v.size();
            if (cls == 1) n_fgp = (int)v.size() - n_plain;
        }
    }

    int upload() {
        std::memset(&bv, 0, sizeof(bv));
        win_fast.assign(P.n_win, 1);
        for (int w = 0; w < P.n_win; ++w) win_fast[w] = P.win[w].schur_fast ? 1 : 0;  // decided at pack time (kba_pack.cpp)
        // Every buffer of the batch view lives in ONE device block: [initialised buffers | zero-filled buffers].
        // Small batches (a single window) stage the initialised part in pinned host memory and upload it with one
        // copy; large ones copy buffer by buffer (no second host copy of hundreds of MB).  One memset for the rest.
        int status = LIMO_OK;
        struct Ent {
            void** slot;
            size_t bytes, off;
            const void* init;
        };
        std::vector<Ent> ents;
        for_each_buffer(P, bv, [&](void** slot, size_t bytes, const void* init) { ents.push_back({slot, bytes ? bytes : 8, 0, init}); });
        // the LM state starts from its reset image (reset_state restores exactly this), uploaded with everything else: a
        // fresh batch needs no reset pass (six copies and a stream drain per single-window call)
        st0.assign(P.n_win, WinState());
        std::memset(st0.data(), 0, sizeof(WinState) * st0.size());
        for (auto& q : st0) q.term = -1;
        for (Ent& e : ents)
            if (e.slot == (void**)&bv.st) e.init = st0.data();
        // the pristine copies limo_ba_batch_reset restores from
        ents.push_back({(void**)&d_pose0, sizeof(double) * 7 * std::max(1, P.TK), 0, P.pose.empty() ? nullptr : P.pose.data()});
        ents.push_back({(void**)&d_pdir0, sizeof(double) * 3 * std::max(1, P.TK), 0, P.pdir.empty() ? nullptr : P.pdir.data()});
        ents.push_back({(void**)&d_pdist0, sizeof(double) * std::max(1, P.TK), 0, P.pdist.empty() ? nullptr : P.pdist.data()});
        ents.push_back({(void**)&d_lm0, sizeof(double) * 3 * std::max(1, P.TL), 0, P.lm.empty() ? nullptr : P.lm.data()});
        ents.push_back({(void**)&d_lm_state0, (size_t)std::max(1, P.TL), 0, P.lm_state.empty() ? nullptr : P.lm_state.data()});
        auto align = [](size_t x) { return (x + 255) / 256 * 256; };
        size_t init_total = 0, zero_total = 0;
        for (Ent& e : ents)
            if (e.init) {
                e.off = init_total;
                init_total += align(e.bytes);
            }
        for (Ent& e : ents)
            if (!e.init) {
                e.off = init_total + zero_total;
                zero_total += align(e.bytes);
            }
        char* arena = nullptr;
        if (dmalloc((void**)&arena, init_total + zero_total)) return LIMO_ERR_RUNTIME;
        for (Ent& e : ents) *e.slot = arena + e.off;
        constexpr size_t kStageMax = 16u << 20;
        if (init_total <= kStageMax) {
            if (ctx->staging_cap < init_total) {
                if (ctx->staging) (void)hipHostFree(ctx->staging);
                ctx->staging = nullptr;
                ctx->staging_cap = 0;
                HIP_TRY(ctx, hipHostMalloc(&ctx->staging, std::max<size_t>(init_total, 1u << 20)));
                ctx->staging_cap = std::max<size_t>(init_total, 1u << 20);
            }
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // an earlier upload may still read the staging buffer
            for (const Ent& e : ents)
                if (e.init) std::memcpy(static_cast<char*>(ctx->staging) + e.off, e.init, e.bytes);
            if (init_total) HIP_TRY(ctx, hipMemcpyAsync(arena, ctx->staging, init_total, hipMemcpyHostToDevice, ctx->stream));
        } else {
            // (a host array that initialises several device buffers - the landmarks: current, candidate and pristine copy - crosses the
            // bus ONCE and is copied on the device for the others: 96 MB of 390 less for 1024 C2 windows, 12.2 -> ~9 ms)
            for (size_t i = 0; i < ents.size(); ++i) {
                const Ent& e = ents[i];
                if (!e.init) continue;
                const Ent* first = nullptr;
                for (size_t j = 0; j < i && !first; ++j)
                    if (ents[j].init == e.init && ents[j].bytes == e.bytes) first = &ents[j];
                if (first)
                    HIP_TRY(ctx, hipMemcpyAsync(arena + e.off, arena + first->off, e.bytes, hipMemcpyDeviceToDevice, ctx->stream));
                else
                    HIP_TRY(ctx, hipMemcpyAsync(arena + e.off, e.init, e.bytes, hipMemcpyHostToDevice, ctx->stream));
            }
        }
        if (zero_total) HIP_TRY(ctx, hipMemsetAsync(arena + init_total, 0, zero_total, ctx->stream));
        if (status != LIMO_OK) return status;
        pv.assign(1, bv);
        if (shard_P > 1) {
            xl = exchange_layout(P);
            d_win_orig = const_cast<WinDesc*>(bv.win);
            pv.assign(local_shards.size(), bv);  // producers keep the batch's own per-workgroup arrays (every entry has ONE owner)
            auto zeros = [&](double** out, size_t n) -> int {
                if (dmalloc((void**)out, sizeof(double) * std::max<size_t>(1, n))) return LIMO_ERR_RUNTIME;
                HIP_TRY(ctx, hipMemsetAsync(*out, 0, sizeof(double) * std::max<size_t>(1, n), ctx->stream));
                return LIMO_OK;
            };
            if (zeros(&arena_c, xl.c_total)) return LIMO_ERR_RUNTIME;
            trims.assign(1 + local_shards.size(), nullptr);
            for (double*& t : trims)
                if (zeros(&t, xl.trim_count)) return LIMO_ERR_RUNTIME;
            blocks.assign(local_shards.size(), nullptr);
            if (!shard_virtual && zeros(&d_gather, (size_t)ctx->comm_world * xl.b_total)) return LIMO_ERR_RUNTIME;
            {   // consumer view: "P workgroups, P rows" per window
                std::vector<WinDesc> wc = exchange_consumer_windows(P);
                WinDesc* d_wc = nullptr;
                if (dmalloc((void**)&d_wc, sizeof(WinDesc) * wc.size())) return LIMO_ERR_RUNTIME;
                HIP_TRY(ctx, hipMemcpy(d_wc, wc.data(), sizeof(WinDesc) * wc.size(), hipMemcpyHostToDevice));
                exchange_bind_consumer(xl, bv, arena_c, trims[0], d_wc);
            }
            rl.assign(local_shards.size(), RankLists());
            for (size_t i = 0; i < local_shards.size(); ++i) {
                if (zeros(&blocks[i], xl.b_total)) return LIMO_ERR_RUNTIME;  // the shard's own block ...
                exchange_bind_producer(xl, pv[i], blocks[i], trims[1 + i]);
                if (zeros(&pv[i].S_part, xl.spart_count)) return LIMO_ERR_RUNTIME;  // ... and its private Schur slabs
                const int r = local_shards[i];
                auto make = [&](const std::vector<int32_t>& owner, int32_t** full, int32_t** act, int* n) -> int {
                    std::vector<int32_t> v;
                    for (size_t k = 0; k < owner.size(); ++k)
                        if (owner[k] == r) v.push_back((int32_t)k);
                    *n = (int)v.size();
                    if (dmalloc((void**)full, sizeof(int32_t) * std::max<size_t>(1, v.size()))) return LIMO_ERR_RUNTIME;
                    if (dmalloc((void**)act, sizeof(int32_t) * std::max<size_t>(1, v.size()))) return LIMO_ERR_RUNTIME;
                    if (!v.empty()) HIP_TRY(ctx, hipMemcpy(*full, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice));
                    return LIMO_OK;
                };
                if (make(P.blk_owner, &rl[i].full_blk, &rl[i].act_blk, &rl[i].n_full_blk)) return LIMO_ERR_RUNTIME;
                if (make(P.lblk_owner, &rl[i].full_lblk, &rl[i].act_lblk, &rl[i].n_full_lblk)) return LIMO_ERR_RUNTIME;
                {   // Schur blocks (span 1)
                    std::vector<int32_t> v;
                    build_sblk_list(nullptr, 1, 1, r, v, rl[i].n_full_sblk_plain, rl[i].n_full_sblk_fgp);
                    rl[i].n_full_sblk = (int)v.size();
                    if (dmalloc((void**)&rl[i].full_sblk, sizeof(int32_t) * std::max<size_t>(1, v.size()))) return LIMO_ERR_RUNTIME;
                    if (dmalloc((void**)&rl[i].act_sblk, sizeof(int32_t) * std::max<size_t>(1, v.size()))) return LIMO_ERR_RUNTIME;
                    if (!v.empty()) HIP_TRY(ctx, hipMemcpy(rl[i].full_sblk, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice));
                }
            }
            if (!shard_virtual && dmalloc((void**)&d_lm_tmp, sizeof(double) * 3 * std::max(1, P.TL))) return LIMO_ERR_RUNTIME;
        }
        if (dmalloc((void**)&d_plane_rep, sizeof(double) * P.SO)) return LIMO_ERR_RUNTIME;
        if (dmalloc((void**)&d_plane_dep, sizeof(double) * P.SO)) return LIMO_ERR_RUNTIME;
        HIP_TRY(ctx, ctx->host_alloc((void**)&h_active, 64));
        HIP_TRY(ctx, hipHostGetDevicePointer((void**)&d_h_active, h_active, 0));
        bv.n_active_host = nullptr;
        h_flags_bytes = sizeof(int32_t) * std::max(1, P.n_win);
        HIP_TRY(ctx, ctx->host_alloc((void**)&h_flags, h_flags_bytes));
        if (dmalloc((void**)&d_wl_blk, sizeof(int32_t) * std::max(1, P.n_blk))) return LIMO_ERR_RUNTIME;
        if (dmalloc((void**)&d_wl_lblk, sizeof(int32_t) * std::max(1, P.n_lblk))) return LIMO_ERR_RUNTIME;
        if (dmalloc((void**)&d_wl_sblk_part, sizeof(int32_t) * std::max(1, P.n_sblk))) return LIMO_ERR_RUNTIME;
        avg_sblk = P.n_win ? (P.n_sblk + P.n_win - 1) / P.n_win : 0;
        if (dmalloc((void**)&d_wl_win, sizeof(int32_t) * std::max(1, P.n_win))) return LIMO_ERR_RUNTIME;
        if (dmalloc((void**)&d_flags, sizeof(int32_t) * std::max(1, P.n_win))) return LIMO_ERR_RUNTIME;
        for (const WinDesc& d : P.win) max_nc = std::max(max_nc, (int)d.nc);
        {
            bool any_fast = false, any_gen = false;
            int t_gen = 1, nfp_gen = 16, nc_gen = kCamSlots, nv_gen = 1, max_nfq = 0, max_nf = 0;
            for (int w = 0; w < P.n_win; ++w) {
                const WinDesc& d = P.win[w];
                if (win_fast[w]) {
                    any_fast = true;
                    max_nfq = std::max(max_nfq, (int)d.nfq);
                    max_nf = std::max(max_nf, (int)d.nf);
                } else if (d.n_sblk > 0) {
                    any_gen = true;
                    t_gen = std::max(t_gen, d.nf_pad / 16);
                    nfp_gen = std::max(nfp_gen, (int)d.nf_pad);
                    nc_gen = std::max(nc_gen, (int)d.nc);
                    nv_gen = std::max(nv_gen, (int)d.n_view);
                }
            }
            if (any_fast) {
                schur_vp = (max_nfq + 16) / 16 <= 1 ? 1 : 2;
                schur_vg = std::min(3, (max_nf + 16) / 16);
                schur_fn_plain = (max_nfq + 16) / 16 <= 1 ? (const void*)k_schur_lean<1, false, 4> : (const void*)k_schur_lean<2, false, 3>;
                plain_lds_bytes = schur_lean_lds_bytes(max_nfq + 1);
                HIP_TRY(ctx, hipFuncSetAttribute(schur_fn_plain, hipFuncAttributeMaxDynamicSharedMemorySize, plain_lds_bytes));
                const int tg = (max_nf + 16) / 16;
                schur_fn_leangp = tg <= 1 ? (const void*)k_schur_lean<1, true, 3> : tg == 2 ? (const void*)k_schur_lean<2, true, 2> : (const void*)k_schur_lean<3, true, 2>;
                leangp_lds_bytes = schur_lean_lds_bytes(max_nf + 1);
                HIP_TRY(ctx, hipFuncSetAttribute(schur_fn_leangp, hipFuncAttributeMaxDynamicSharedMemorySize, leangp_lds_bytes));
                schur_pair_ok = schur_vp == 2 && schur_vg == 3 && !(std::getenv("KBA_NO_SCHUR_PAIR") && std::atoi(std::getenv("KBA_NO_SCHUR_PAIR")) != 0);
                if (schur_pair_ok)
                    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_schur_lean_pair<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, std::max(plain_lds_bytes, leangp_lds_bytes)));
            }
            if (any_gen) {
                const int tiles = t_gen * (t_gen + 1) / 2, npw = (tiles + kWideWaves - 1) / kWideWaves;
                schur_fn_gen = npw <= 1 ? (const void*)k_schur_wide<1> : npw <= 3 ? (const void*)k_schur_wide<3>
                             : npw <= 6 ? (const void*)k_schur_wide<6> : (const void*)k_schur_wide<12>;
                if (npw > 12) {
                    ctx->err = "window with too many free camera slots for k_schur_wide";
                    return LIMO_ERR_INVALID;
                }
                wide_lds_bytes = schur_wide_lds_bytes(nfp_gen, nc_gen, nv_gen);
                HIP_TRY(ctx, hipFuncSetAttribute(schur_fn_gen, hipFuncAttributeMaxDynamicSharedMemorySize, wide_lds_bytes));
            }
        }
        {   // LDS of the window-level kernels: the largest window that still works in LDS (the others: cam_scr_off)
            int nc_lds = kCamSlots, nf_lds = 1, nv_lds = 1;
            for (const WinDesc& d : P.win)
                if (d.cam_scr_off < 0) {
                    nc_lds = std::max(nc_lds, (int)d.nc);
                    nf_lds = std::max(nf_lds, (int)d.nf);
                    nv_lds = std::max(nv_lds, (int)d.n_view);
                }
            asm_bytes = cam_assemble_scratch(nc_lds, kBlock, nv_lds) * (int)sizeof(double);
            solve_bytes = cam_solve_scratch(nc_lds, kBlock, nf_lds) * (int)sizeof(double);  // the compact system: nf <= nc free slots
        }
        int max_lm = 1;
        for (const WinDesc& d : P.win) max_lm = std::max(max_lm, (int)d.n_lm);
        {
            int np2 = 1;
            while (np2 < max_lm) np2 <<= 1;
            trim_bytes = np2 <= kTrimMaxSort ? np2 * 12 + max_lm + 16 : 16;
        }
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_cam_assemble, hipFuncAttributeMaxDynamicSharedMemorySize, asm_bytes));
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_cam_solve, hipFuncAttributeMaxDynamicSharedMemorySize, solve_bytes));
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_trim_select, hipFuncAttributeMaxDynamicSharedMemorySize, trim_bytes));
        for (auto& e : act_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreate(&ev_total_a));
        HIP_TRY(ctx, hipEventCreate(&ev_total_b));
        return LIMO_OK;
    }

    bool pristine = true;  // the device state is the state of create / reset: what a recovered barrier timeout of the one-launch solve restores
    int reset_state() {
        pristine = true;
        HIP_TRY(ctx, hipMemcpyAsync(bv.st, st0.data(), sizeof(WinState) * st0.size(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(bv.pose, d_pose0, sizeof(double) * 7 * P.TK, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(bv.pdir, d_pdir0, sizeof(double) * 3 * P.TK, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(bv.pdist, d_pdist0, sizeof(double) * P.TK, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(bv.lm, d_lm0, sizeof(double) * 3 * P.TL, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(bv.lm_state, d_lm_state0, P.TL, hipMemcpyDeviceToDevice, ctx->stream));
        return LIMO_OK;
    }

    void note(hipError_t e, const char* what) {
        if (e != hipSuccess && rc == LIMO_OK) {
            rc = LIMO_ERR_RUNTIME;
            ctx->err = std::string(what) + ": " + hipGetErrorString(e);
        }
    }
#define LAUNCH_CHECK(what) note(hipGetLastError(), what)

    // ---- Executor
    // Schur granularity: a wave takes two plain Schur blocks (128 landmarks, 8 tiles) or one ground-plane block.  Fixed
    // (not a function of how many windows are in flight): the partial-slab layout of a window, and with it the order in
    // which its Schur complement is summed, is then the same alone and inside any batch - single-window and batched
    // solves give the same bits.  (Measured at 1024 C2 windows: span 2 is as fast as 4, span 1 costs 2 %.)
    // The spans also decide which slabs of S_part are "plain" (written in part only, kba_kernels.hip:schur_lean_group) - a slab must keep
    // its class for the life of the batch, so the A/B environment variables are read ONCE per batch, at its first solve.
    int span_env = 0, span_gp_env = 0;  // 0: not read yet
    void set_span(int) {
        if (span_env == 0) {
            span_env = 2;
            span_gp_env = 1;
            if (const char* e = std::getenv("KBA_SPAN_GP")) span_gp_env = std::max(1, std::atoi(e));
            if (const char* e = std::getenv("KBA_SPAN")) span_env = std::max(1, std::atoi(e));  // A/B timing aids
        }
        c.schur_span = span_env;
        c.schur_span_gp = span_gp_env;
        if (shard_P > 1) c.schur_span = c.schur_span_gp = 1;  // Schur blocks are cut at shard boundaries
        c.schur_nslab = shard_P > 1 ? shard_P : 0;
        c.schur_packed = shard_P > 1 ? 1 : 0;
    }

    // ---- sharding helpers
    const int32_t* list_blk(size_t i) const { return shard_P > 1 ? rl[i].blk : (use_wl ? d_wl_blk : nullptr); }
    const int32_t* list_lblk(size_t i) const { return shard_P > 1 ? rl[i].lblk : (use_wl ? d_wl_lblk : nullptr); }
    const int32_t* list_sblk(size_t i) const { return shard_P > 1 ? rl[i].sblk : d_wl_sblk; }
    int count_blk(size_t i) const { return shard_P > 1 ? rl[i].n_blk : n_wl_blk; }
    int count_lblk(size_t i) const { return shard_P > 1 ? rl[i].n_lblk : n_wl_lblk; }
    int count_sblk(size_t i) const { return shard_P > 1 ? rl[i].n_sblk : n_wl_sblk; }
    int count_sblk_plain(size_t i) const { return shard_P > 1 ? rl[i].n_sblk_plain : n_wl_sblk_plain; }
    int count_sblk_fgp(size_t i) const { return shard_P > 1 ? rl[i].n_sblk_fgp : n_wl_sblk_fgp; }
    int shard_of(size_t i) const { return shard_P > 1 ? local_shards[i] : 0; }

    // Per-iteration exchange (point 1 = A1, 2 = A2, 3 = A, 4 = B; kba_buffers.hpp:exchange_layout): ALL-GATHER of the shards'
    // blocks over the ranks - one call per local slot, normally one shard per rank: ONE ncclAllGather of 41 KB at a C4 window -
    // then k_unpack puts every contribution at its shard's place in the consumer view.  Shard s lives on rank s mod world, in
    // local slot s / world.  Virtual shards (no communicator): the unpack alone.
    void exchange(int point) {
        if (shard_P == 1) return;
        hipStream_t s = ctx->stream;
        size_t off, count;
        xl.range(point, off, count);
        for (size_t i = 0; i < pv.size(); ++i) {
            if (shard_virtual) {
                hipLaunchKernelGGL(k_unpack, dim3(cdiv((int64_t)count, 256)), dim3(256), 0, s, xl, (const WinDesc*)d_win_orig, (const double*)(blocks[i] + off), arena_c,
                                   local_shards[i], (int64_t)off, (int64_t)count);
                LAUNCH_CHECK("k_unpack");
            } else {
                if (ctx->xfn) {
                    host_exchange(blocks[i] + off, d_gather, count, 0);
                } else {
                    ncclResult_t r = ncclAllGather(blocks[i] + off, d_gather, count, ncclDouble, (ncclComm_t)ctx->comm, s);
                    if (r != ncclSuccess && rc == LIMO_OK) {
                        rc = LIMO_ERR_RUNTIME;
                        ctx->err = std::string("ncclAllGather: ") + ncclGetErrorString(r);
                    }
                }
                for (int q = 0; q < ctx->comm_world; ++q) {
                    hipLaunchKernelGGL(k_unpack, dim3(cdiv((int64_t)count, 256)), dim3(256), 0, s, xl, (const WinDesc*)d_win_orig,
                                       (const double*)(d_gather + (size_t)q * count), arena_c, (int)i * ctx->comm_world + q, (int64_t)off, (int64_t)count);
                    LAUNCH_CHECK("k_unpack");
                }
            }
            ++n_exchanges;
            exchange_bytes += (int64_t)count * (int64_t)sizeof(double);
        }
    }
    // The caller's transport (limo_ctx_comm_init_host): `count` doubles at `src` (device) go to the host, through the callback
    // (kind 0: all-gather into world * count doubles, rank order; kind 1: sum over the ranks into count doubles) and back to `dst`.
    void host_exchange(const double* src, double* dst, size_t count, int kind) {
        hipStream_t s = ctx->stream;
        const size_t n_out = kind == 0 ? (size_t)ctx->comm_world * count : count, need = count + n_out;
        if (ctx->xhost_cap < need) {
            if (ctx->xhost) (void)hipHostFree(ctx->xhost);
            ctx->xhost = nullptr;
            ctx->xhost_cap = 0;
            if (hipHostMalloc((void**)&ctx->xhost, sizeof(double) * need) != hipSuccess) {
                note(hipErrorOutOfMemory, "hipHostMalloc(exchange staging)");
                std::vector<double> z(need, 0.0);  // (the peers are still met: see below)
                ctx->xfn(z.data(), z.data() + count, (long long)count, kind, ctx->xuser);
                return;
            }
            ctx->xhost_cap = need;
        }
        // The transport is a COLLECTIVE: a rank that skipped it after a local error would leave its peers blocked in it for ever.  So
        // the call is made in every case - after an error with a zeroed contribution - and only the device copies are skipped; the
        // solve of this rank ends with LIMO_ERR_RUNTIME, the peers' solves end (with a result that misses this rank's share).
        if (rc == LIMO_OK) {
            note(hipMemcpyAsync(ctx->xhost, src, sizeof(double) * count, hipMemcpyDeviceToHost, s), "exchange: device -> host");
            note(hipStreamSynchronize(s), "exchange: sync");
        }
        if (rc != LIMO_OK) std::memset(ctx->xhost, 0, sizeof(double) * count);
        ctx->xfn(ctx->xhost, ctx->xhost + count, (long long)count, kind, ctx->xuser);
        if (rc != LIMO_OK) return;
        note(hipMemcpyAsync(dst, ctx->xhost + count, sizeof(double) * n_out, hipMemcpyHostToDevice, s), "exchange: host -> device");
        note(hipStreamSynchronize(s), "exchange: sync");  // (the staging buffer is reused by the next step)
    }
    // Trimming round: per-landmark residual maxima, one owner per entry and zero elsewhere - an exact sum in any order.
    void exchange_trim() {
        if (shard_P == 1) return;
        hipStream_t s = ctx->stream;
        const int64_t n = (int64_t)xl.trim_count;
        ShardPtrs sp;
        for (size_t i = 0; i < pv.size(); ++i) sp.p[i] = trims[1 + i];
        hipLaunchKernelGGL(k_sum_shards<double>, dim3(cdiv(n, 256)), dim3(256), 0, s, trims[0], sp, (int)pv.size(), n);
        LAUNCH_CHECK("k_sum_shards");
        if (!shard_virtual && ctx->xfn) {
            host_exchange(trims[0], trims[0], (size_t)n, 1);
        } else if (!shard_virtual) {
            ncclResult_t r = ncclAllReduce(trims[0], trims[0], (size_t)n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, s);
            if (r != ncclSuccess && rc == LIMO_OK) {
                rc = LIMO_ERR_RUNTIME;
                ctx->err = std::string("ncclAllReduce: ") + ncclGetErrorString(r);
            }
        }
        ++n_exchanges;
        exchange_bytes += n * (int64_t)sizeof(double);
    }

    void full_lists() {
        use_wl = false;
        n_wl_blk = P.n_blk;
        n_wl_lblk = P.n_lblk;
        n_wl_win = P.n_win;
        listed = P.n_win;
        set_span(P.n_win);
        FullSblk& fl = wl_sblk_full[{c.schur_span, c.schur_span_gp}];
        if (!fl.d) {
            std::vector<int32_t> v;
            build_sblk_list(nullptr, c.schur_span, c.schur_span_gp, -1, v, fl.n_plain, fl.n_fgp);
            fl.n = (int)v.size();
            if (dmalloc((void**)&fl.d, sizeof(int32_t) * std::max<size_t>(1, v.size())) == LIMO_OK && !v.empty())
                note(hipMemcpy(fl.d, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice), "upload Schur worklist");
        }
        d_wl_sblk = fl.d;
        n_wl_sblk = fl.n;
        n_wl_sblk_plain = fl.n_plain;
        n_wl_sblk_fgp = fl.n_fgp;
        for (RankLists& r : rl) {
            r.n_sblk_plain = r.n_full_sblk_plain;
            r.n_sblk_fgp = r.n_full_sblk_fgp;
            r.blk = r.full_blk;
            r.lblk = r.full_lblk;
            r.sblk = r.full_sblk;
            r.n_blk = r.n_full_blk;
            r.n_lblk = r.n_full_lblk;
            r.n_sblk = r.n_full_sblk;
        }
    }

    // Rebuild the worklists from the windows that are active right now (synchronises the stream).
    void rebatch() {
        hipStream_t s = ctx->stream;
        hipLaunchKernelGGL(k_export_active, dim3(cdiv(P.n_win, 256)), dim3(256), 0, s, bv, d_flags);
        note(hipMemcpyAsync(h_flags, d_flags, sizeof(int32_t) * P.n_win, hipMemcpyDeviceToHost, s), "memcpy flags");
        note(hipStreamSynchronize(s), "sync flags");
        if (rc != LIMO_OK) return;
        std::vector<int32_t> wb, wlb, wsb, ww;
        for (int w = 0; w < P.n_win; ++w) {
            if (!h_flags[w]) continue;
            const WinDesc& d = P.win[w];
            ww.push_back(w);
            for (int i = 0; i < d.n_blk; ++i) wb.push_back(d.blk0 + i);
            for (int i = 0; i < d.n_lblk; ++i) wlb.push_back(d.lblk0 + i);
        }
        set_span((int)ww.size());
        build_sblk_list(&ww, c.schur_span, c.schur_span_gp, -1, wsb, n_wl_sblk_plain, n_wl_sblk_fgp);
        auto up = [&](int32_t* dst, const std::vector<int32_t>& v) {
            if (!v.empty()) note(hipMemcpyAsync(dst, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice, s), "upload worklist");
        };
        up(d_wl_blk, wb);
        up(d_wl_lblk, wlb);
        d_wl_sblk = d_wl_sblk_part;
        up(d_wl_sblk, wsb);
        up(d_wl_win, ww);
        std::vector<std::vector<int32_t>> keep;  // host lists of the shards, alive until the sync below
        for (size_t i = 0; i < rl.size(); ++i) {
            const int r = local_shards[i];
            std::vector<int32_t> b1, b2, b3;
            for (int w : ww) {
                const WinDesc& d = P.win[w];
                for (int k = d.blk0; k < d.blk0 + d.n_blk; ++k)
                    if (P.blk_owner[k] == r) b1.push_back(k);
                for (int k = d.lblk0; k < d.lblk0 + d.n_lblk; ++k)
                    if (P.lblk_owner[k] == r) b2.push_back(k);
            }
            build_sblk_list(&ww, 1, 1, r, b3, rl[i].n_sblk_plain, rl[i].n_sblk_fgp);
            up(rl[i].act_blk, b1);
            up(rl[i].act_lblk, b2);
            up(rl[i].act_sblk, b3);
            rl[i].blk = rl[i].act_blk;
            rl[i].lblk = rl[i].act_lblk;
            rl[i].sblk = rl[i].act_sblk;
            rl[i].n_blk = (int)b1.size();
            rl[i].n_lblk = (int)b2.size();
            rl[i].n_sblk = (int)b3.size();
            keep.push_back(std::move(b1));
            keep.push_back(std::move(b2));
            keep.push_back(std::move(b3));
        }
        note(hipStreamSynchronize(s), "sync worklists");  // the host vectors go out of scope
        use_wl = true;
        n_wl_blk = (int)wb.size();
        n_wl_lblk = (int)wlb.size();
        n_wl_sblk = (int)wsb.size();
        n_wl_win = (int)ww.size();
        listed = n_wl_win;
    }

    // k_lin_lm<4, true> (the default since round 6): four waves per SIMD, the window's view constants, the landmark block's running sums and
    // the tail's inputs in LDS (kba_kernels.hip:lin_lm_block VLDS / ACCL) - scalar loads of the view constants return out of order, so
    // every use of one waited for all of them (round 5: 550 -> 513 us per round of 4096 slots through LDS).  With the camera-side sums in
    // their raw form (kba_items.hpp:lin_cam_half0: the 3 x 6 pose Jacobian is never formed) the kernel needs 129 registers instead of 149,
    // so the fourth wave costs nothing else: 505 -> 495 us per round, bench line +0.7 % (KBA_LIN_WAVES=3, read once, takes <3, true>).
    // Batches whose windows have so many views that the copy would cost occupancy (> 48 KB of LDS per workgroup: more than ~16 views)
    // and KBA_LIN_VLDS=0 (read once) take <., false>: scalar loads, sums in registers - same bits in all variants.
    // KBA_LIN_LDS_PAD=<bytes>: extra dynamic LDS per workgroup (occupancy experiments).
    int lin_lds_set[3] = {0, 0, 0};
    void launch_lin_lm(int grid, hipStream_t s, const BatchView& v, const int32_t* wl) {
        static const int lw = std::getenv("KBA_LIN_WAVES") ? std::atoi(std::getenv("KBA_LIN_WAVES")) : 4;
        static const int want_vlds = std::getenv("KBA_LIN_VLDS") ? std::atoi(std::getenv("KBA_LIN_VLDS")) : 1;
        static const int pad = std::getenv("KBA_LIN_LDS_PAD") ? std::atoi(std::getenv("KBA_LIN_LDS_PAD")) : 0;
        const bool four = lw >= 4;
        const bool vlds = want_vlds != 0 && lin_lm_lds_bytes(P.Vmax, true, true) <= 48 * 1024;
        const int which = four ? 2 : vlds ? 1 : 0;
        const void* fn = four ? (vlds ? (const void*)k_lin_lm<4, true> : (const void*)k_lin_lm<4, false>) : vlds ? (const void*)k_lin_lm<3, true> : (const void*)k_lin_lm<3, false>;
        const int lds = lin_lm_lds_bytes(P.Vmax, four, vlds) + pad;
        if (lds > 48 * 1024 && lin_lds_set[which] < lds) {  // (beyond the default dynamic-LDS limit: many views, or the padding)
            note(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds), "hipFuncSetAttribute(k_lin_lm)");
            lin_lds_set[which] = lds;
        }
        const BatchView* vp = &v;
        const SolveConsts* cp = &c;
        void* args[] = {(void*)vp, (void*)cp, (void*)&wl};
        note(hipLaunchKernel(fn, dim3(grid), dim3(kBlock), args, lds, s), "launch k_lin_lm");
    }

    void solve_init(int max_iter, int select) override {
        it_no = 0;
        first_lin = true;  // the next linearisation defines the Jacobi scaling (WinState::compute_scale)
        assemble_pending = false;  // (a solve that ended right behind a linearisation leaves nothing for the next one)
        full_lists();
        hipLaunchKernelGGL(k_solve_init, dim3(cdiv(P.n_win, 256)), dim3(256), 0, ctx->stream, bv, c, max_iter, select);
        LAUNCH_CHECK("k_solve_init");
    }

    // Event pair around one launch of a timed kernel (nullptr once the pool is exhausted).
    EventPair* timed(int kernel, hipStream_t on = nullptr) {
        if (ev_used >= 16384) return nullptr;
        if (ev_used == ev_pool.size()) {
            EventPair e;
            e.kernel = kernel;
            note(hipEventCreate(&e.a), "hipEventCreate");
            note(hipEventCreate(&e.b), "hipEventCreate");
            ev_pool.push_back(e);
        }
        EventPair* ep = &ev_pool[ev_used++];
        ep->kernel = kernel;
        note(hipEventRecord(ep->a, on ? on : ctx->stream), "hipEventRecord");
        return ep;
    }

    void linearize() override {
        hipStream_t s = ctx->stream;
        {
            if (P.TV) {
                hipLaunchKernelGGL(k_view_consts, dim3(cdiv(P.TV, 256)), dim3(256), 0, s, bv);
                LAUNCH_CHECK("k_view_consts");
            }
            // ground-plane rows first: the landmark pass adds them to the landmark blocks
            EventPair* ep = timed(LIMO_KERNEL_LINEARIZE);
            for (size_t i = 0; i < pv.size(); ++i)
                if (count_lblk(i)) {
                    launch_lin_lm(count_lblk(i), s, pv[i], list_lblk(i));
                    LAUNCH_CHECK("k_lin_lm");
                }
            if (ep) note(hipEventRecord(ep->b, s), "hipEventRecord");
        }
        if (shard_P > 1) {
            for (size_t i = 0; i < pv.size(); ++i)
                if (n_wl_win) {
                    hipLaunchKernelGGL(k_shard_reduce, dim3(n_wl_win), dim3(kBlock), 0, s, pv[i], use_wl ? (const int32_t*)d_wl_win : (const int32_t*)nullptr, shard_of(i), 0);
                    LAUNCH_CHECK("k_shard_reduce");
                }
            // Sharded: the camera assembly only has to come before the Schur complement when it defines the Jacobi scale (the
            // first linearisation of a solve).  Every other iteration it waits for the slabs and ONE exchange serves both (step()).
            if (!first_lin) {
                assemble_pending = true;
                return;
            }
            first_lin = false;
            exchange(1);
        }
        assemble(it_no);
    }

    // k_cam_assemble + the active-window counter of iteration `iter`: a ring of 4 slots so the host can read iteration i-1
    // while iteration i runs
    void assemble(int iter) {
        hipStream_t s = ctx->stream;
        const int slot = iter & 3;
        BatchView bvs = bv;
        bvs.n_active = bv.n_active + 2 * slot;
        bvs.n_active_host = d_h_active + slot;
        if (n_wl_win) hipLaunchKernelGGL(k_cam_assemble, dim3(n_wl_win), dim3(kBlock), asm_bytes, s, bvs, c, use_wl ? d_wl_win : nullptr);
        LAUNCH_CHECK("k_cam_assemble");
        note(hipEventRecord(act_ev[slot], s), "record n_active");
    }

    // Lagging by one iteration: the count of iteration i-1 is read while iteration i is already enqueued, so the
    // host never drains the stream inside a solve.  The price is at most one extra iteration of no-op launches.
    int active_count() override {
        if (rc != LIMO_OK) return 0;
        const int cur = it_no++;
        if (cur == 0) return P.n_win;  // iteration zero: nothing to wait for yet
        const int slot = (cur - 1) & 3;
        note(hipEventSynchronize(act_ev[slot]), "sync n_active");
        if (rc != LIMO_OK) return 0;
        const int a = h_active[slot];
        static const bool trace = std::getenv("KBA_TRACE_ACTIVE") != nullptr;  // profiling aid
        if (trace) std::fprintf(stderr, "[kba] iteration %d: active %d, listed %d, span %d\n", cur - 1, a, listed, c.schur_span);
        // re-batch when at most half of the listed windows still iterate (and the list is worth shrinking)
        if (a > 0 && listed >= 8 && 2 * a <= listed) rebatch();
        return a;
    }

    void expire(int) override {
        if (assemble_pending) {  // sharded: the camera assembly of the last linearisation was deferred behind the Schur slabs - the
                                 // time cap ends the solve before them: assemble now (exchange of the linearisation sums only), so
                                 // that the windows' cost is the cost of the point they stop at
            assemble_pending = false;
            exchange(1);
            assemble(it_no - 1);
        }
        hipLaunchKernelGGL(k_expire, dim3(cdiv(P.n_win, 256)), dim3(256), 0, ctx->stream, bv);
        LAUNCH_CHECK("k_expire");
    }

    void step() override {
        hipStream_t s = ctx->stream;
        for (size_t i = 0; i < pv.size(); ++i)
            if (count_lblk(i)) {
                hipLaunchKernelGGL(k_lm_damp, dim3(count_lblk(i)), dim3(kBlock), 0, s, pv[i], c, list_lblk(i));
                LAUNCH_CHECK("k_lm_damp");
            }
        {
            EventPair* ep = timed(LIMO_KERNEL_SCHUR);
            for (size_t i = 0; i < pv.size(); ++i) {
                int n_plain = count_sblk_plain(i), n_fgp = count_sblk_fgp(i);
                const int n_gen = count_sblk(i) - n_plain - n_fgp;
                int span = c.schur_span, span_gp = c.schur_span_gp;
                const int32_t* wlp = list_sblk(i);
                if (n_plain) {
                    void* args[] = {(void*)&pv[i], (void*)&wlp, (void*)&span, (void*)&span_gp};
                    note(hipLaunchKernel(schur_fn_plain, dim3(n_plain), dim3(64), args, plain_lds_bytes, s), "launch k_schur_lean");
                    LAUNCH_CHECK("k_schur_lean");
                    wlp += n_plain;
                }
                if (n_fgp) {
                    void* args[] = {(void*)&pv[i], (void*)&wlp, (void*)&span, (void*)&span_gp};
                    note(hipLaunchKernel(schur_fn_leangp, dim3(n_fgp), dim3(64), args, leangp_lds_bytes, s), "launch k_schur_lean (gp)");
                    LAUNCH_CHECK("k_schur_lean (gp)");
                    wlp += n_fgp;
                }
                if (n_gen) {
                    void* args[] = {(void*)&pv[i], (void*)&wlp, (void*)&span, (void*)&span_gp};
                    note(hipLaunchKernel(schur_fn_gen, dim3(n_gen), dim3(64 * kWideWaves), args, wide_lds_bytes, s), "launch k_schur_wide");
                    LAUNCH_CHECK("k_schur_wide");
                }
            }
            if (ep) note(hipEventRecord(ep->b, s), "hipEventRecord");
        }
        if (shard_P > 1) {
            if (n_wl_win)
                for (size_t i = 0; i < pv.size(); ++i) {
                    const int32_t* wlw = use_wl ? (const int32_t*)d_wl_win : (const int32_t*)nullptr;
                    hipLaunchKernelGGL(k_shard_reduce, dim3(n_wl_win), dim3(kBlock), 0, s, pv[i], wlw, shard_of(i), 1);
                    hipLaunchKernelGGL(k_slab_reduce, dim3(n_wl_win, 8), dim3(kBlock), 0, s, pv[i], wlw, shard_P);
                    LAUNCH_CHECK("k_slab_reduce");
                }
            if (assemble_pending) {  // ONE exchange: camera-side sums + ground-plane blocks + scalars + [S | rhs]
                assemble_pending = false;
                exchange(3);
                assemble(it_no - 1);  // (active_count() has moved it_no on to the next iteration)
            } else {
                exchange(2);
            }
        }
        if (n_wl_win) hipLaunchKernelGGL(k_cam_solve, dim3(n_wl_win), dim3(kBlock), solve_bytes, s, bv, c, use_wl ? d_wl_win : nullptr);
        LAUNCH_CHECK("k_cam_solve");
        for (size_t i = 0; i < pv.size(); ++i) {
            if (count_lblk(i)) {
                hipLaunchKernelGGL(k_backsub, dim3(count_lblk(i)), dim3(kBlock), 0, s, pv[i], c, list_lblk(i));
                LAUNCH_CHECK("k_backsub");
            }
        }
        if (shard_P > 1 && n_wl_win)
            for (size_t i = 0; i < pv.size(); ++i) {
                hipLaunchKernelGGL(k_shard_reduce, dim3(n_wl_win), dim3(kBlock), 0, s, pv[i], use_wl ? (const int32_t*)d_wl_win : (const int32_t*)nullptr, shard_of(i), 2);
                LAUNCH_CHECK("k_shard_reduce");
            }
        exchange(4);
        if (n_wl_win) hipLaunchKernelGGL(k_step_decide, dim3(n_wl_win), dim3(64), 0, s, bv, c, use_wl ? d_wl_win : nullptr);
        LAUNCH_CHECK("k_step_decide");
        hipLaunchKernelGGL(k_accept, dim3(cdiv((int64_t)P.TK + P.TL, 256)), dim3(256), 0, s, bv);
        LAUNCH_CHECK("k_accept");
    }

    void trim() override {
        hipStream_t s = ctx->stream;
        bool any = false;
        for (const WinDesc& d : P.win) any = any || d.do_trim;
        if (!any) return;
        for (size_t i = 0; i < pv.size(); ++i) {
            const int nb = shard_P > 1 ? rl[i].n_full_blk : P.n_blk;
            if (nb) {
                hipLaunchKernelGGL(k_trim_residual, dim3(nb), dim3(kBlock), 0, s, pv[i], d_plane_rep, d_plane_dep,
                                   shard_P > 1 ? (const int32_t*)rl[i].full_blk : (const int32_t*)nullptr);
                LAUNCH_CHECK("k_trim_residual");
            }
        }
        if (P.TL) {
            for (size_t i = 0; i < pv.size(); ++i) {
                hipLaunchKernelGGL(k_trim_max, dim3(cdiv(P.TL, 256)), dim3(256), 0, s, pv[i], (const double*)d_plane_rep,
                                   (const double*)d_plane_dep, shard_of(i), shard_P);
                LAUNCH_CHECK("k_trim_max");
            }
        }
        exchange_trim();
        hipLaunchKernelGGL(k_trim_select, dim3(P.n_win), dim3(kBlock), trim_bytes, s, bv, c);
        LAUNCH_CHECK("k_trim_select");
    }

    // ------------------------------------------------------------------------------------------ streaming solve
    // Slot groups: the slots are split into `n_groups` groups, each with its own stream, worklists and scheduler state;
    // they take their windows from one shared cursor.  The groups run the same round sequence out of phase, so the
    // MFMA-bound Schur kernels and the latency-bound window-level kernels of one group overlap with the HBM-bound
    // scans of the other.
    struct StreamGroup {
        hipStream_t stream = nullptr, trim_stream = nullptr;
        bool own_stream = false;
        hipEvent_t sched_ev = nullptr, trim_ev = nullptr, done_ev = nullptr, round_ev[4] = {nullptr, nullptr, nullptr, nullptr};
        int32_t *d_slot_win = nullptr, *d_lists = nullptr, *d_slot_cnt = nullptr;
        int32_t *h_done = nullptr, *d_h_done = nullptr;  // pinned ring of 4
        int n_slots = 0;
        int cap[SL_COUNT] = {0};
        int mx[SL_COUNT] = {0};  // entries of list k a single window can have
        BatchView sv;
    };
    std::vector<StreamGroup> groups;
    int stream_groups_built = 0;
    // The launch train of a round (round 6): the per-view constants inside k_sched_fill, accepted landmarks + re-damping in ONE launch
    // (k_after_step) - 12 launches per round instead of 14.  KBA_UNFUSED_TRAIN=1 (read per solve) keeps k_view_consts, k_lm_damp and
    // k_accept as launches of their own: the same device functions, the same bits (tests/test_gpu_reproducible.py).
    bool fused_train = true;
    hipEvent_t start_ev = nullptr;

    void stream_teardown() {
        for (StreamGroup& g : groups) {
            if (g.own_stream && g.stream) (void)hipStreamDestroy(g.stream);
            if (g.trim_stream) (void)hipStreamDestroy(g.trim_stream);
            for (hipEvent_t e : {g.sched_ev, g.trim_ev, g.done_ev, g.round_ev[0], g.round_ev[1], g.round_ev[2], g.round_ev[3]})
                if (e) (void)hipEventDestroy(e);
            if (g.h_done) ctx->host_free(g.h_done, 64);
        }
        groups.clear();
        if (start_ev) (void)hipEventDestroy(start_ev);
        start_ev = nullptr;
    }

    int stream_slots() const {
        int n = std::min((int)P.n_win, std::max(1024, std::min(kSchedMaxSlots, (int)P.n_win / 4)));
        if (const char* e = std::getenv("KBA_SLOTS")) n = std::max(1, std::min({std::atoi(e), (int)P.n_win, kSchedMaxSlots}));
        return n;
    }
    int stream_setup(int n_groups) {
        if (stream_groups_built == n_groups) return LIMO_OK;
        (void)hipStreamSynchronize(ctx->stream);
        stream_teardown();  // (device blocks of an earlier layout stay with the batch until it is destroyed)
        // windows in flight: a quarter of the batch (so that the ramp-down at the end of the batch is a small part of the
        // solve), at least 1024 (a round of fewer windows is bound by the latency of its window-level kernels)
        n_slots = stream_slots();
        set_span(P.n_win);
        int mx[SL_COUNT] = {0};
        for (const WinDesc& d : P.win) {
            const int plg = (d.n_sblk_plain + c.schur_span - 1) / c.schur_span, gpg = (d.n_sblk - d.n_sblk_plain + c.schur_span_gp - 1) / c.schur_span_gp;
            mx[SL_LBLK] = std::max(mx[SL_LBLK], (int)d.n_lblk);
            mx[SL_TBLK] = std::max(mx[SL_TBLK], (int)d.n_blk);
            mx[SL_SPLAIN] = std::max(mx[SL_SPLAIN], d.schur_fast ? plg : 0);
            mx[SL_SFGP] = std::max(mx[SL_SFGP], d.schur_fast ? gpg : 0);
            mx[SL_SGEN] = std::max(mx[SL_SGEN], d.schur_fast ? 0 : plg + gpg);
        }
        mx[SL_WIN] = 1;
        mx[SL_TLBLK] = mx[SL_LBLK];
        mx[SL_TWIN] = 1;
        if (!d_sched_ctl && dmalloc((void**)&d_sched_ctl, sizeof(int32_t) * 8)) return LIMO_ERR_RUNTIME;
        HIP_TRY(ctx, hipEventCreateWithFlags(&start_ev, hipEventDisableTiming));
        groups.resize(n_groups);
        for (int gi = 0; gi < n_groups; ++gi) {
            StreamGroup& g = groups[gi];
            g.n_slots = n_slots / n_groups + (gi < n_slots % n_groups ? 1 : 0);
            g.sv = bv;
            size_t total = 0;
            for (int k = 0; k < SL_COUNT; ++k) {
                g.mx[k] = mx[k];
                g.cap[k] = g.n_slots * mx[k];
                g.sv.sched_off[k] = (int32_t)total;
                total += 1 + (size_t)std::max(1, g.cap[k]);
            }
            if (dmalloc((void**)&g.d_lists, sizeof(int32_t) * total)) return LIMO_ERR_RUNTIME;
            if (dmalloc((void**)&g.d_slot_win, sizeof(int32_t) * std::max(1, g.n_slots))) return LIMO_ERR_RUNTIME;
            if (dmalloc((void**)&g.d_slot_cnt, sizeof(int32_t) * (SL_COUNT + 1) * std::max(1, g.n_slots))) return LIMO_ERR_RUNTIME;
            HIP_TRY(ctx, hipMemsetAsync(g.d_lists, 0, sizeof(int32_t) * total, ctx->stream));
            HIP_TRY(ctx, ctx->host_alloc((void**)&g.h_done, 64));
            HIP_TRY(ctx, hipHostGetDevicePointer((void**)&g.d_h_done, g.h_done, 0));
            for (auto& e : g.round_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&g.sched_ev, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&g.trim_ev, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&g.done_ev, hipEventDisableTiming));
            HIP_TRY(ctx, hipStreamCreateWithFlags(&g.trim_stream, hipStreamNonBlocking));
            if (gi == 0) {
                g.stream = nullptr;  // the context's stream, looked up at solve time (limo_ctx_set_stream)
            } else {
                HIP_TRY(ctx, hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
                g.own_stream = true;
            }
            g.sv.counted = 1;
            g.sv.n_slots = g.n_slots;
            g.sv.slot_win = g.d_slot_win;
            g.sv.sched_ctl = d_sched_ctl;
            g.sv.slot_cnt = g.d_slot_cnt;
            g.sv.sched_lists = g.d_lists;
            g.sv.sched_done_host = g.d_h_done;
            g.sv.n_active_host = nullptr;
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        stream_groups_built = n_groups;
        return LIMO_OK;
    }

    // One round of one group = scheduler + (side stream) trimming of the windows whose trimming solve just ended + one LM
    // iteration of every window in the group's slots.
    // in_flight_max: an upper bound on the windows this group can have in its slots in this round (the batch's unfinished windows as
    // of the pinned done-counter the host read last).  The list-driven kernels are launched over bound x (entries per window)
    // workgroups instead of the lists' capacities: while the batch drains, a round of 200 windows in 2048 slots no longer
    // dispatches 16 000 workgroups per kernel that read one count word and leave (~3 ns each: 50 us per kernel, and a round has
    // a dozen of them - the "launch-latency floor" of the drain was mostly this).
    void enqueue_round(StreamGroup& g, int round, bool time_kernels, int in_flight_max) {
        hipStream_t s = g.stream;
        const BatchView& sv = g.sv;
        int cap[SL_COUNT];
        const int bound = std::max(1, std::min(g.n_slots, in_flight_max));
        for (int k = 0; k < SL_COUNT; ++k) cap[k] = std::min(g.cap[k], bound * g.mx[k]);
        auto L = [&](int k) { return (const int32_t*)(g.d_lists + sv.sched_off[k] + 1); };
        if (round > 0) note(hipStreamWaitEvent(s, g.trim_ev, 0), "wait trim");  // last round's trimming re-armed its windows
        hipLaunchKernelGGL(k_sched_advance, dim3(cdiv(g.n_slots, 256)), dim3(256), 0, s, sv, c);
        hipLaunchKernelGGL(k_sched_scan, dim3(1), dim3(kSchedThreads), 0, s, sv, round);
        hipLaunchKernelGGL(k_sched_fill, dim3(cdiv(g.n_slots, 4)), dim3(256), 0, s, sv, c, fused_train ? 1 : 0);
        LAUNCH_CHECK("scheduler kernels");
        // ---- trimming of the windows whose trimming solve just ended, on the side stream: k_trim_select is a
        //      latency-bound sort (one workgroup per window, ~0.3 ms) - the other windows iterate meanwhile, the
        //      trimmed ones join again in the next round (k_trim_select arms their next solve)
        note(hipEventRecord(g.sched_ev, s), "record sched");
        note(hipStreamWaitEvent(g.trim_stream, g.sched_ev, 0), "wait sched");
        if (cap[SL_TBLK]) hipLaunchKernelGGL(k_trim_residual, dim3(cap[SL_TBLK]), dim3(kBlock), 0, g.trim_stream, sv, d_plane_rep, d_plane_dep, L(SL_TBLK));
        if (cap[SL_TLBLK]) hipLaunchKernelGGL(k_trim_max, dim3(cap[SL_TLBLK]), dim3(kBlock), 0, g.trim_stream, sv, (const double*)d_plane_rep, (const double*)d_plane_dep, 0, 1);
        hipLaunchKernelGGL(k_trim_select, dim3(cap[SL_TWIN]), dim3(kBlock), trim_bytes, g.trim_stream, sv, c);
        LAUNCH_CHECK("trim kernels");
        note(hipEventRecord(g.trim_ev, g.trim_stream), "record trim");
        // ---- linearisation of the windows that need it (their per-view constants: k_sched_fill above, or the launch of its own)
        if (!fused_train) hipLaunchKernelGGL(k_view_consts, dim3(cap[SL_WIN]), dim3(64), 0, s, sv);
        {
            EventPair* ep = time_kernels ? timed(LIMO_KERNEL_LINEARIZE, s) : nullptr;
            if (cap[SL_LBLK]) launch_lin_lm(cap[SL_LBLK], s, sv, L(SL_LBLK));
            if (ep) note(hipEventRecord(ep->b, s), "hipEventRecord");
        }
        hipLaunchKernelGGL(k_cam_assemble, dim3(cap[SL_WIN]), dim3(kBlock), asm_bytes, s, sv, c, L(SL_WIN));
        LAUNCH_CHECK("linearisation kernels");
        // ---- trust-region step of the windows that iterate
        if (!fused_train && cap[SL_LBLK]) hipLaunchKernelGGL(k_lm_damp, dim3(cap[SL_LBLK]), dim3(kBlock), 0, s, sv, c, L(SL_LBLK));
        {
            EventPair* ep = time_kernels ? timed(LIMO_KERNEL_SCHUR, s) : nullptr;
            int span = c.schur_span, span_gp = c.schur_span_gp;
            // few windows in flight (the batch drains): both fast-class lists in one launch (kba_kernels.hip:k_schur_lean_pair)
            static const int pair_bound = std::getenv("KBA_SCHUR_PAIR_BOUND") ? std::atoi(std::getenv("KBA_SCHUR_PAIR_BOUND")) : kSchurPairBound;
            const bool pair = schur_pair_ok && bound <= pair_bound && cap[SL_SPLAIN] && cap[SL_SFGP];
            if (pair) {
                const int32_t *wlp = L(SL_SPLAIN), *wlg = L(SL_SFGP);
                int n_plain_cap = cap[SL_SPLAIN];
                void* args[] = {(void*)&sv, (void*)&wlp, (void*)&n_plain_cap, (void*)&wlg, (void*)&span, (void*)&span_gp};
                note(hipLaunchKernel((const void*)k_schur_lean_pair<2, 3>, dim3(cap[SL_SPLAIN] + cap[SL_SFGP]), dim3(64), args, std::max(plain_lds_bytes, leangp_lds_bytes), s),
                     "launch k_schur_lean_pair");
            }
            if (!pair && cap[SL_SPLAIN]) {
                const int32_t* wlp = L(SL_SPLAIN);
                void* args[] = {(void*)&sv, (void*)&wlp, (void*)&span, (void*)&span_gp};
                note(hipLaunchKernel(schur_fn_plain, dim3(cap[SL_SPLAIN]), dim3(64), args, plain_lds_bytes, s), "launch k_schur_lean");
            }
            if (!pair && cap[SL_SFGP]) {
                const int32_t* wlp = L(SL_SFGP);
                void* args[] = {(void*)&sv, (void*)&wlp, (void*)&span, (void*)&span_gp};
                note(hipLaunchKernel(schur_fn_leangp, dim3(cap[SL_SFGP]), dim3(64), args, leangp_lds_bytes, s), "launch k_schur_lean (gp)");
            }
            if (cap[SL_SGEN]) {
                const int32_t* wlp = L(SL_SGEN);
                void* args[] = {(void*)&sv, (void*)&wlp, (void*)&span, (void*)&span_gp};
                note(hipLaunchKernel(schur_fn_gen, dim3(cap[SL_SGEN]), dim3(64 * kWideWaves), args, wide_lds_bytes, s), "launch k_schur_wide");
            }
            if (ep) note(hipEventRecord(ep->b, s), "hipEventRecord");
        }
        hipLaunchKernelGGL(k_cam_solve, dim3(cap[SL_WIN]), dim3(kBlock), solve_bytes, s, sv, c, L(SL_WIN));
        if (cap[SL_LBLK]) hipLaunchKernelGGL(k_backsub, dim3(cap[SL_LBLK]), dim3(kBlock), 0, s, sv, c, L(SL_LBLK));
        hipLaunchKernelGGL(k_step_decide, dim3(cap[SL_WIN]), dim3(64), 0, s, sv, c, L(SL_WIN));
        if (cap[SL_LBLK]) {
            if (fused_train)  // accepted landmarks / re-damping in one launch (kba_kernels.hip:k_after_step)
                hipLaunchKernelGGL(k_after_step, dim3(cap[SL_LBLK]), dim3(kBlock), 0, s, sv, c, L(SL_LBLK));
            else
                hipLaunchKernelGGL(k_accept, dim3(cap[SL_LBLK]), dim3(256), 0, s, sv);
        }
        LAUNCH_CHECK("step kernels");
        note(hipEventRecord(g.round_ev[round & 3], s), "record round");
    }

    // ---- a whole solve in ONE launch (kba_kernels.hip:k_solve_wg): windows without free landmarks - adjustPoseOnly -
    // whose landmark workgroups a single workgroup walks through in a few microseconds.  KBA_NO_WG_SOLVE=1 (read per
    // call) keeps the lock-step launches: the tests compare the two paths bit by bit.
    static constexpr int kWgMaxLblk = 8;  // <= 2048 landmarks per window
    bool wg_solve_applies() const {
        if (shard_P != 1 || P.evaluate_only || P.n_sblk != 0 || P.n_win < 1) return false;
        if (const char* e = std::getenv("KBA_NO_WG_SOLVE"))
            if (std::atoi(e) != 0) return false;
        for (const WinDesc& d : P.win)
            if (d.n_lblk > kWgMaxLblk) return false;
        return wg_lds_bytes() <= kCamLdsCapBytes;
    }
    int wg_lds_bytes() const { return std::max(std::max(asm_bytes, solve_bytes), std::max(trim_bytes, lin_lm_lds_bytes(P.Vmax, true, true))); }
    void solve_wg() {
        const int lds = wg_lds_bytes();
        set_span(P.n_win);
        note(hipFuncSetAttribute((const void*)k_solve_wg, hipFuncAttributeMaxDynamicSharedMemorySize, lds), "hipFuncSetAttribute(k_solve_wg)");
        const long long cap_ticks = opts.max_solver_time_sec > 0.0 ? std::max(1ll, (long long)(opts.max_solver_time_sec * 1e8)) : 0ll;
        hipLaunchKernelGGL(k_solve_wg, dim3(P.n_win), dim3(kBlock), lds, ctx->stream, bv, c, cap_ticks, d_plane_rep, d_plane_dep);
        LAUNCH_CHECK("k_solve_wg");
    }

    // ---- one window (a few windows), ONE cooperative launch (kba_kernels.hip:k_solve_coop): G workgroups per window that
    // meet at device-wide barriers where the lock-step solve has launch boundaries.  KBA_NO_COOP_SOLVE=1 (read per call)
    // keeps the lock-step launches (the tests compare the two paths bit by bit).
    // At most 64 windows per launch (>= 4 workgroups per window): measured on C2 windows (scripts/gpu_small_batch.py), one launch
    // vs streaming solve: 16 windows 10.8 / 19.1 ms, 64: 17.4 / 22.4 ms, 128: 25.0 / 24.0 ms, 256: 34.0 / 27.7 ms
    // (KBA_COOP_MAX_WIN: timing aid, up to 256).
    static constexpr int kCoopMaxWg = 256, kCoopMaxWin = 64;
    int32_t* d_coop_bar = nullptr;
    double* d_coop_red = nullptr;
    bool coop_launched = false;
    int coop_G = 0, coop_xcd = 1;
    int coop_lds_bytes() const {
        const int wave = (std::max(plain_lds_bytes, leangp_lds_bytes) + 15) / 16 * 16;
        return std::max(std::max(std::max(asm_bytes, solve_bytes), std::max(trim_bytes, lin_lm_lds_bytes(P.Vmax, true, true))), (kBlock / 64) * wave);
    }
    bool coop_solve_applies() {
        if (shard_P != 1 || P.evaluate_only || P.n_win < 1) return false;
        if (const char* e = std::getenv("KBA_NO_COOP_SOLVE"))
            if (std::atoi(e) != 0) return false;
        set_span(P.n_win);
        int G = 1;
        for (const WinDesc& d : P.win) {
            if (!d.schur_fast || d.cam_scr_off >= 0 || d.nf_pad * d.nf_pad > kCoopRedStride) return false;
            const int tasks = (d.n_sblk_plain + c.schur_span - 1) / c.schur_span + (d.n_sblk - d.n_sblk_plain + c.schur_span_gp - 1) / c.schur_span_gp;
            // enough workgroups for one landmark workgroup each, and for one Schur group per wave next to workgroup 0
            G = std::max(G, std::max((int)d.n_lblk, tasks ? 1 + (tasks + kBlock / 64 - 1) / (kBlock / 64) : 1));
        }
        G = std::min(G, 32);
        if (const char* e = std::getenv("KBA_COOP_G")) G = std::max(1, std::min(64, std::atoi(e)));  // (timing aid)
        // one workgroup per CU, all resident: a batch of up to kCoopMaxWin windows shares the chip with fewer workgroups per
        // window (64 windows: 4 each) - still far ahead of ten launches per iteration over 64 slots
        if (P.n_win > kCoopMaxWg || coop_lds_bytes() > kCamLdsCapBytes) return false;
        // the workgroups of a window on one XCD (k_solve_coop): the grid is 8 G ceil(n_win / 8), of which n_win G blocks work
        coop_xcd = 1;
        if (const char* e = std::getenv("KBA_COOP_XCD")) coop_xcd = std::atoi(e) != 0;  // (read per call: A/B timing)
        const int per8 = ((int)P.n_win + 7) / 8;
        if (coop_xcd && 8 * per8 > kCoopMaxWg) coop_xcd = 0;
        G = std::max(1, std::min(G, coop_xcd ? kCoopMaxWg / (8 * per8) : kCoopMaxWg / (int)P.n_win));
        coop_G = G;
        return true;
    }
    // false: the launch was refused (nothing ran) - the caller takes the lock-step path
    bool solve_coop() {
        const int lds = coop_lds_bytes();
        if (!d_coop_bar && dmalloc((void**)&d_coop_bar, sizeof(int32_t) * 8 * P.n_win)) return false;
        if (!d_coop_red && dmalloc((void**)&d_coop_red, sizeof(double) * kCoopRedStride * P.n_win)) return false;
        if (hipFuncSetAttribute((const void*)k_solve_coop, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        note(hipMemsetAsync(d_coop_bar, 0, sizeof(int32_t) * 8 * P.n_win, ctx->stream), "memset barrier words");
        h_active[8] = 0;
        CoopParams cp;
        cp.G = coop_G;
        cp.xcd_map = coop_xcd;
        const int coop_grid = coop_xcd ? 8 * coop_G * (((int)P.n_win + 7) / 8) : (int)P.n_win * coop_G;
        cp.vp = schur_vp;
        cp.vg = schur_vg;
        cp.schur_lds = (std::max(plain_lds_bytes, leangp_lds_bytes) + 15) / 16 * 16 / (int)sizeof(double);
        cp.cap_ticks = opts.max_solver_time_sec > 0.0 ? std::max(1ll, (long long)(opts.max_solver_time_sec * 1e8)) : 0ll;
        // barrier timeout in ticks of the 100 MHz constant clock (read per call; KBA_COOP_TIMEOUT_MS=0: give up at the first wait -
        // how the tests reach the recovery path of limo_ba_batch_solve)
        // default 50 ms (the whole call is ~5 ms; the longest phase between two barriers - the quantile trimming - well under 1 ms):
        // a barrier that is not met by then is lost, and the caller of a 10 Hz pipeline should not wait seconds to learn it
        cp.timeout_ticks = 5000000ll;
        if (const char* e = std::getenv("KBA_COOP_TIMEOUT_MS")) cp.timeout_ticks = std::max(0ll, (long long)(std::atof(e) * 1e5));
        cp.bar = d_coop_bar;
        cp.abort_host = d_h_active + 8;
        cp.plane_rep = d_plane_rep;
        cp.plane_dep = d_plane_dep;
        cp.red = d_coop_red;
        void* args[] = {(void*)&bv, (void*)&c, (void*)&cp};
        // A PLAIN launch by default, with the co-residency of the grid checked here against the occupancy the runtime reports
        // (the workgroups of other kernels on the device all finish, so every workgroup of this grid gets its CU; a barrier that
        // waits too long is recovered, coop_sync).  hipLaunchCooperativeKernel gives the same guarantee from the runtime, but
        // after the first such launch of a process every LATER solve that uses several streams ran 30-50 % slower (a 2048-window
        // batch 91 -> 118-137 ms, scripts/gpu_groups_sequence.py; with plain launches 91 ms) - KBA_COOP_PLAIN_LAUNCH=0 takes that API.
        static const bool plain_launch = !(std::getenv("KBA_COOP_PLAIN_LAUNCH") && std::atoi(std::getenv("KBA_COOP_PLAIN_LAUNCH")) == 0);
        if (plain_launch) {
            int per_cu = 0, n_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_solve_coop, kBlock, (size_t)lds) != hipSuccess ||
                hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || per_cu * n_cu < coop_grid) {
                (void)hipGetLastError();
                return false;  // (not all workgroups resident at once: the launch sequence)
            }
        }
        const hipError_t e = plain_launch ? hipLaunchKernel((const void*)k_solve_coop, dim3(coop_grid), dim3(kBlock), args, lds, ctx->stream)
                                          : hipLaunchCooperativeKernel((const void*)k_solve_coop, dim3(coop_grid), dim3(kBlock), args, lds, ctx->stream);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        coop_launched = true;
        return true;
    }

    // The host only enqueues; it learns that all windows are done from a pinned word the scheduler writes, two rounds
    // late (so the streams never drain inside a solve).
    int solve_streaming() {
        // two slot groups from 512 windows on (A/B at 256 .. 1536 windows: 5-9 % on the re-solve at every size; the FIRST solve of a batch
        // pays the second group's streams and events, which a one-shot batch of 256 windows does not earn back: 44 vs 36 ms)
        // THREE groups once a group still holds ~1400 slots (4096 slots = batches of 16384 windows: 36.4 vs 35.8 k windows/s, alternating runs
        // on one box; at 1024-2048 slots a third group costs 1-4 %, a fourth 7 % at 4096: profiles/r06_experiment_launch_train.txt)
        int n_groups = stream_slots() >= 4096 ? 3 : P.n_win >= 512 ? 2 : 1;
        if (const char* e = std::getenv("KBA_GROUPS")) n_groups = std::max(1, std::min(4, std::atoi(e)));
        if (stream_setup(n_groups) != LIMO_OK) return LIMO_ERR_RUNTIME;
        set_span(P.n_win);
        hipStream_t s0 = ctx->stream;
        groups[0].stream = s0;
        HIP_TRY(ctx, hipMemsetAsync(d_sched_ctl, 0, sizeof(int32_t) * 8, s0));
        fused_train = !(std::getenv("KBA_UNFUSED_TRAIN") && std::atoi(std::getenv("KBA_UNFUSED_TRAIN")) != 0);
        for (StreamGroup& g : groups) {
            HIP_TRY(ctx, hipMemsetAsync(g.d_slot_win, 0xFF, sizeof(int32_t) * std::max(1, g.n_slots), s0));
            for (int i = 0; i < 4; ++i) g.h_done[i] = 0;
        }
        HIP_TRY(ctx, hipEventRecord(start_ev, s0));
        for (size_t gi = 1; gi < groups.size(); ++gi) HIP_TRY(ctx, hipStreamWaitEvent(groups[gi].stream, start_ev, 0));
        const bool time_kernels = groups.size() == 1;  // kernel timing by events is only meaningful without overlap
        constexpr int kLag = 2;
        // KBA_SCHED_TRACE=1 (profiling aid): windows finished as of every round -> how full the slots are over the solve
        static const bool sched_trace = std::getenv("KBA_SCHED_TRACE") != nullptr;
        std::vector<int> done_curve;
        double enq_us = 0.0;  // host time inside enqueue_round (KBA_SCHED_TRACE)
        const auto t_solve0 = std::chrono::steady_clock::now();
        bool finished = false;
        int done_seen = 0;  // windows finished, as last read from the pinned ring (monotone, kLag + 1 rounds old when it is used)
        static const bool shrink = !(std::getenv("KBA_NO_GRID_SHRINK") && std::atoi(std::getenv("KBA_NO_GRID_SHRINK")) != 0);
        for (int round = 0; !finished; ++round) {
            const int in_flight_max = shrink ? (int)P.n_win - done_seen : (int)P.n_win;
            const auto t_enq0 = std::chrono::steady_clock::now();
            for (StreamGroup& g : groups) enqueue_round(g, round, time_kernels, in_flight_max);
            if (sched_trace) enq_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enq0).count();
            if (rc != LIMO_OK) break;
            if (round >= kLag) {
                finished = true;
                for (StreamGroup& g : groups) {
                    note(hipEventSynchronize(g.round_ev[(round - kLag) & 3]), "sync round");
                    if (rc != LIMO_OK) break;
                    const int d = g.h_done[(round - kLag) & 3];
                    if (d < P.n_win) finished = false;
                    done_seen = std::max(done_seen, d);
                }
                if (sched_trace) done_curve.push_back(groups[0].h_done[(round - kLag) & 3]);
            }
            if (round > 200000) {
                rc = LIMO_ERR_RUNTIME;
                ctx->err = "streaming solve did not terminate";
                break;
            }
        }
        if (sched_trace && !done_curve.empty()) {
            // in flight at round r = min(n_slots, windows not finished); printed as deciles of the round count
            const int R = (int)done_curve.size();
            long long occ = 0;
            std::fprintf(stderr, "[kba] streaming solve: %d windows, %d slots, %d groups, %d rounds; in flight at 0,10,..100 %% of the rounds:", (int)P.n_win, n_slots, (int)groups.size(), R);
            for (int r = 0; r < R; ++r) occ += std::min(n_slots, (int)P.n_win - done_curve[r]);
            for (int d = 0; d <= 10; ++d) std::fprintf(stderr, " %d", std::min(n_slots, (int)P.n_win - done_curve[std::min(R - 1, d * R / 10)]));
            std::fprintf(stderr, "; mean occupancy %.3f; host: %.1f us per round inside the enqueue calls, %.1f us per round wall\n", (double)occ / ((double)R * n_slots),
                         enq_us / (R + kLag), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_solve0).count() / (R + kLag));
        }
        for (StreamGroup& g : groups) {  // everything joins the context's stream again
            note(hipStreamWaitEvent(g.stream, g.trim_ev, 0), "wait trim");
            if (g.stream != s0) {
                note(hipEventRecord(g.done_ev, g.stream), "record done");
                note(hipStreamWaitEvent(s0, g.done_ev, 0), "wait group");
            }
        }
        return rc;
    }

    int collect_linearize_events() {
        if (ev_used) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            for (size_t i = 0; i < ev_used; ++i) {
                float ms = 0.f;
                HIP_TRY(ctx, hipEventElapsedTime(&ms, ev_pool[i].a, ev_pool[i].b));
                k_ms_acc[ev_pool[i].kernel] += ms;
                k_launches[ev_pool[i].kernel] += 1;
            }
            ev_used = 0;
        }
        return LIMO_OK;
    }
};

// =============================================================================================== C-ABI
extern "C" {

int limo_abi_version(void) {
    return LIMO_ABI_VERSION;
}

int limo_ctx_create(int device, limo_ctx** out) {
    if (!out) return LIMO_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return LIMO_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return LIMO_ERR_NO_DEVICE;
    limo_ctx* c = new limo_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return LIMO_ERR_RUNTIME;
    }
    c->stream = c->own;
    *out = c;
    return LIMO_OK;
}

void* limo_host_alloc(size_t bytes) {
    void* p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1) == hipSuccess ? p : nullptr;
}
void limo_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

void limo_ctx_destroy(limo_ctx* ctx) {
    if (!ctx) return;
    if (ctx->depth_ws && ctx->depth_ws_free) ctx->depth_ws_free(ctx->depth_ws);
    if (ctx->comm) (void)ncclCommDestroy((ncclComm_t)ctx->comm);
    ctx->pool_release();
    if (ctx->own) (void)hipStreamDestroy(ctx->own);
    delete ctx;
}

int limo_ctx_set_stream(limo_ctx* ctx, void* hip_stream) {
    if (!ctx) return LIMO_ERR_INVALID;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own;
    return LIMO_OK;
}

const char* limo_last_error(const limo_ctx* ctx) {
    return ctx ? ctx->err.c_str() : "null context";
}

void limo_ba_default_options(limo_ba_options* o) {
    if (!o) return;
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

static int batch_create_impl(limo_ctx* ctx, int32_t n, const limo_ba_window* windows, const limo_ba_options* opts,
                             const PackOptions& po, limo_ba_batch** out, bool shards_are_ranks = false) {
    if (!ctx || !out) return LIMO_ERR_INVALID;
    *out = nullptr;
    limo_ba_options o;
    if (opts)
        o = *opts;
    else
        limo_ba_default_options(&o);
    limo_ba_batch* b = new limo_ba_batch();
    b->ctx = ctx;
    b->opts = o;
    b->c = make_consts(o);
    const auto t_c0 = std::chrono::steady_clock::now();
    // Large batches pack into the context's pinned arena when no other live batch holds it (limo_ctx.hpp:pack_arena; KBA_NO_PACK_ARENA=1
    // keeps the heap).  The first large batch of a context only measures what it would have needed; the arena is made right behind its
    // packing, for the next one.
    static const bool arena_off = std::getenv("KBA_NO_PACK_ARENA") && std::atoi(std::getenv("KBA_NO_PACK_ARENA")) != 0;
    PackArena lend;
    const bool lending = !arena_off && n >= 128 && !ctx->pack_arena_busy && hipSetDevice(ctx->device) == hipSuccess;
    auto arena_resize = [&](size_t wanted) {  // (only while no batch holds the arena)
        const size_t step = size_t(64) << 20;
        const size_t cap = std::min(limo_ctx::kPackArenaMax, (wanted + wanted / 8 + step - 1) / step * step);
        if (cap <= ctx->pack_arena_cap) return;
        if (ctx->pack_arena) {
            pack_arena_register(ctx->pack_arena, ctx->pack_arena_cap, false);
            (void)hipHostFree(ctx->pack_arena);
            ctx->pack_arena = nullptr;
            ctx->pack_arena_cap = 0;
        }
        if (hipHostMalloc(&ctx->pack_arena, cap) == hipSuccess) {
            ctx->pack_arena_cap = cap;
            pack_arena_register(ctx->pack_arena, cap, true);
        } else {
            (void)hipGetLastError();
            ctx->pack_arena = nullptr;
        }
    };
    if (lending) {
        if (ctx->pack_arena && ctx->pack_arena_wanted > ctx->pack_arena_cap) arena_resize(ctx->pack_arena_wanted);  // the last batch did not fit
        lend.base = static_cast<char*>(ctx->pack_arena);
        lend.cap = ctx->pack_arena_cap;
        pack_arena_lend(&lend);
    }
    int rc = pack_windows(n, windows, o, po, b->P, ctx->err);
    pack_arena_lend(nullptr);
    if (lending) {
        ctx->pack_arena_wanted = std::max(ctx->pack_arena_wanted, lend.wanted);
        if (lend.used > 0) {
            ctx->pack_arena_busy = true;
            b->holds_pack_arena = true;
        } else if (!ctx->pack_arena && lend.wanted > 0) {
            arena_resize(lend.wanted);  // the first large batch of the context: ready for the next one
        }
    }
    const auto t_c1 = std::chrono::steady_clock::now();
    if (rc == LIMO_OK) {
        if (hipSetDevice(ctx->device) != hipSuccess) rc = LIMO_ERR_NO_DEVICE;
    }
    b->shard_P = b->P.n_shards;
    b->shard_virtual = !shards_are_ranks;
    b->shard_rank = shards_are_ranks ? ctx->comm_rank : 0;
    if (b->shard_P > 1) {  // shard s lives on rank s mod world (all of them here when there is no communicator)
        const int world = shards_are_ranks ? ctx->comm_world : 1, me = shards_are_ranks ? ctx->comm_rank : 0;
        for (int r = 0; r < b->shard_P; ++r)
            if (r % world == me) b->local_shards.push_back(r);
        if (b->local_shards.size() > 8) {
            ctx->err = "at most 8 shards per GPU";
            delete b;
            return LIMO_ERR_INVALID;
        }
    }
    if (rc == LIMO_OK) rc = b->upload();
    if (std::getenv("KBA_PACK_TRACE"))
        std::fprintf(stderr, "[kba] create: pack %.1f ms, upload %.1f ms\n", std::chrono::duration<double, std::milli>(t_c1 - t_c0).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_c1).count());
    if (rc != LIMO_OK) {
        delete b;
        return rc;
    }
    *out = b;
    return LIMO_OK;
}

int limo_ba_batch_create(limo_ctx* ctx, int32_t n_windows, const limo_ba_window* windows, limo_ba_batch** out) {
    // min_landmarks_for_trimming decides do_trim at pack time; take it from the default options here and
    // refresh at solve() if the caller passes different options.
    return batch_create_impl(ctx, n_windows, windows, nullptr, PackOptions(), out);
}

int limo_ba_batch_solve(limo_ba_batch* b, const limo_ba_options* opts) {
    if (!b) return LIMO_ERR_INVALID;
    limo_ctx* ctx = b->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) return LIMO_ERR_NO_DEVICE;
    if (opts) {
        if (opts->min_landmarks_for_trimming != b->opts.min_landmarks_for_trimming) {
            for (int w = 0; w < b->P.n_win; ++w) b->P.win[w].do_trim = b->P.win[w].n_lm > opts->min_landmarks_for_trimming;
            if (b->shard_P > 1) {  // (the consumer view of a sharded batch carries its own descriptors: both copies follow)
                const std::vector<WinDesc> wc = exchange_consumer_windows(b->P);
                HIP_TRY(ctx, hipMemcpy((void*)b->d_win_orig, b->P.win.data(), sizeof(WinDesc) * b->P.n_win, hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMemcpy((void*)b->bv.win, wc.data(), sizeof(WinDesc) * b->P.n_win, hipMemcpyHostToDevice));
            } else {
                HIP_TRY(ctx, hipMemcpyAsync((void*)b->bv.win, b->P.win.data(), sizeof(WinDesc) * b->P.n_win, hipMemcpyHostToDevice,
                                            ctx->stream));
            }
        }
        b->opts = *opts;
        b->c = make_consts(*opts);
    }
    b->rc = LIMO_OK;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(ctx, hipEventRecord(b->ev_total_a, ctx->stream));
    // Windows of a batch converge after very different numbers of iterations: from a few windows on they stream through
    // slots (k_sched) instead of advancing in lock-step.  Not for sharded solves (exchange steps between the kernels)
    // and not with a wall-clock cap (a per-solve clock, run_schedule keeps it).
    // ... and small batches (<= 256 windows of the common shape) run as ONE launch: a workgroup per window (no free
    // landmark) or G workgroups per window that meet at device-wide barriers (k_solve_coop).  All paths give the same bits.
    const bool env_stream = std::getenv("KBA_STREAM_MIN") != nullptr;  // (read per call: the tests switch paths)
    const int stream_min = env_stream ? std::atoi(std::getenv("KBA_STREAM_MIN")) : 16;
    const int coop_max = std::getenv("KBA_COOP_MAX_WIN") ? std::min(std::atoi(std::getenv("KBA_COOP_MAX_WIN")), (int)limo_ba_batch::kCoopMaxWg) : limo_ba_batch::kCoopMaxWin;
    const bool can_stream = b->shard_P == 1 && b->opts.max_solver_time_sec <= 0.0 && b->P.n_win >= stream_min && !b->P.evaluate_only;
    const bool one_launch = !(env_stream && can_stream) && b->P.n_win <= coop_max;  // (KBA_STREAM_MIN set: the caller asks for the streaming solve)
    auto launch_sequence = [&]() {
        if (can_stream)
            b->solve_streaming();
        else
            run_schedule(*b, b->opts);
    };
    if (one_launch && b->wg_solve_applies())
        b->solve_wg();
    // (the cooperative solve only from the pristine state - its timeout recovery restores THAT state, a warm re-solve would lose the
    // first solve's result - and not any more in a context whose launches keep timing out: something shares the GPU)
    // (three strikes switch the one-launch path off - but not for the life of the context: after kCoopRetryAfter solves through the launch
    // sequence ONE more attempt is made (a profiler session or a neighbour process that has gone away); its success clears the strikes)
    else if (one_launch && b->pristine && b->coop_solve_applies() && (ctx->coop_strikes < 3 || ++ctx->coop_benched >= limo_ctx::kCoopRetryAfter) && b->solve_coop())
        ctx->coop_benched = 0;
    else
        launch_sequence();
    b->pristine = false;
    if (b->coop_launched) {
        // A device-wide barrier of k_solve_coop that was not met in time aborts the launch (kba_kernels.hip:coop_sync) and leaves
        // poses, landmarks and LM state half-updated.  That is recoverable: the batch's initial state is restored (the pristine
        // copies limo_ba_batch_reset uses) and the same windows go through the launch sequence - same results, bit for bit.
        b->coop_launched = false;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (b->h_active[8] == 0) ctx->coop_strikes = 0;
        if (b->h_active[8] != 0) {
            b->h_active[8] = 0;
            ++ctx->coop_fallbacks;
            ++ctx->coop_strikes;
            if (b->reset_state() != LIMO_OK) return LIMO_ERR_RUNTIME;
            b->rc = LIMO_OK;
            launch_sequence();
            b->pristine = false;
        }
    }
    if (b->shard_P > 1 && !b->shard_virtual) {  // every rank ends with every landmark: sum of "owned, else zero"
        hipLaunchKernelGGL(k_lm_owned, dim3(cdiv(b->P.TL, 256)), dim3(256), 0, ctx->stream, b->bv, b->d_lm_tmp, b->shard_rank, b->shard_P,
                           ctx->comm_world);
        if (ctx->xfn) {
            b->host_exchange(b->d_lm_tmp, b->bv.lm, (size_t)3 * b->P.TL, 1);
            if (b->rc != LIMO_OK) return b->rc;
        } else {
            ncclResult_t r = ncclAllReduce(b->d_lm_tmp, b->bv.lm, (size_t)3 * b->P.TL, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream);
            if (r != ncclSuccess) {
                ctx->err = std::string("ncclAllReduce(landmarks): ") + ncclGetErrorString(r);
                return LIMO_ERR_RUNTIME;
            }
        }
        ++b->n_exchanges;
        b->exchange_bytes += (int64_t)sizeof(double) * 3 * b->P.TL;
    }
    HIP_TRY(ctx, hipEventRecord(b->ev_total_b, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, b->ev_total_a, b->ev_total_b));
    b->total_ms_acc += ms;
    b->last_solve_sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return b->rc;
}

int limo_ba_batch_reset(limo_ba_batch* b) {
    if (!b) return LIMO_ERR_INVALID;
    if (hipSetDevice(b->ctx->device) != hipSuccess) return LIMO_ERR_NO_DEVICE;
    return b->reset_state();
}

int limo_ba_batch_download(limo_ba_batch* b, limo_ba_window* windows_out, limo_ba_report* reports) {
    if (!b) return LIMO_ERR_INVALID;
    limo_ctx* ctx = b->ctx;
    const PackedBatch& P = b->P;
    std::vector<double> pose_v, pdir_v, pdist_v, lm_v;
    std::vector<WinState> st_v;
    const double *pose = nullptr, *pdir = nullptr, *pdist = nullptr, *lm = nullptr;
    const WinState* st = nullptr;
    {
        // [st | pose | pdir | pdist | lm] are neighbours in the batch's device block (kba_buffers.hpp order, upload()):
        // a small batch comes back with ONE copy into the context's pinned staging buffer instead of five into pageable
        // vectors (each of those is a blocking staged copy inside the runtime).
        const char* lo = reinterpret_cast<const char*>(b->bv.st);
        const char* hi = reinterpret_cast<const char*>(b->bv.lm) + sizeof(double) * 3 * (size_t)P.TL;
        auto inside = [&](const void* q) { return reinterpret_cast<const char*>(q) >= lo && reinterpret_cast<const char*>(q) < hi; };
        // A LARGE batch takes the same single copy through a pinned buffer the context keeps for it (grow-only, up to kBigStageMax):
        // five copies into fresh pageable vectors were staged by the runtime and paid their page faults - 6 of the 10 ms a
        // 1024-window download took.
        const bool neighbours = hi > lo && inside(b->bv.pose) && inside(b->bv.pdir) && inside(b->bv.pdist) && (P.TL == 0 || inside(b->bv.lm));
        const size_t span = neighbours ? (size_t)(hi - lo) : 0;
        void* hbuf = nullptr;
        if (neighbours && span <= ctx->staging_cap && ctx->staging) {
            hbuf = ctx->staging;
        } else if (neighbours && span <= limo_ctx::kBigStageMax) {
            if (ctx->staging_big_cap < span) {
                if (ctx->staging_big) (void)hipHostFree(ctx->staging_big);
                ctx->staging_big = nullptr;
                ctx->staging_big_cap = 0;
                const size_t cap = std::min(limo_ctx::kBigStageMax, span + span / 4);
                if (hipHostMalloc(&ctx->staging_big, cap) == hipSuccess)
                    ctx->staging_big_cap = cap;
                else
                    (void)hipGetLastError();  // (no pinned memory to be had: the pageable path below)
            }
            if (ctx->staging_big_cap >= span) hbuf = ctx->staging_big;
        }
        if (hbuf) {
            HIP_TRY(ctx, hipMemcpyAsync(hbuf, lo, span, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            const char* h = static_cast<const char*>(hbuf);
            auto at = [&](const void* q) { return h + (reinterpret_cast<const char*>(q) - lo); };
            st = reinterpret_cast<const WinState*>(at(b->bv.st));
            pose = reinterpret_cast<const double*>(at(b->bv.pose));
            pdir = reinterpret_cast<const double*>(at(b->bv.pdir));
            pdist = reinterpret_cast<const double*>(at(b->bv.pdist));
            lm = reinterpret_cast<const double*>(at(b->bv.lm));
        } else {
            pose_v.resize((size_t)P.TK * 7);
            pdir_v.resize((size_t)P.TK * 3);
            pdist_v.resize(P.TK);
            lm_v.resize((size_t)P.TL * 3);
            st_v.resize(P.n_win);
            HIP_TRY(ctx, hipMemcpyAsync(pose_v.data(), b->bv.pose, sizeof(double) * pose_v.size(), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(pdir_v.data(), b->bv.pdir, sizeof(double) * pdir_v.size(), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(pdist_v.data(), b->bv.pdist, sizeof(double) * pdist_v.size(), hipMemcpyDeviceToHost, ctx->stream));
            if (P.TL) HIP_TRY(ctx, hipMemcpyAsync(lm_v.data(), b->bv.lm, sizeof(double) * lm_v.size(), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(st_v.data(), b->bv.st, sizeof(WinState) * st_v.size(), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            pose = pose_v.data();
            pdir = pdir_v.data();
            pdist = pdist_v.data();
            lm = lm_v.data();
            st = st_v.data();
        }
    }
    if (windows_out)
        for (int w = 0; w < P.n_win; ++w)
            if (windows_out[w].n_kf != P.win[w].n_kf || windows_out[w].n_lm != P.win[w].n_lm) {
                ctx->err = "download: window shape differs from create()";
                return LIMO_ERR_INVALID;
            }
    // windows are independent (every window writes its own arrays): host threads share them when the batch is large
    auto one_window = [&](int w) {
        const WinDesc& d = P.win[w];
        if (windows_out) {
            limo_ba_window& W = windows_out[w];
            std::memcpy(W.kf_pose, pose + 7 * (size_t)d.kf0, sizeof(double) * 7 * d.n_kf);
            std::memcpy(W.kf_plane_dir, pdir + 3 * (size_t)d.kf0, sizeof(double) * 3 * d.n_kf);
            std::memcpy(W.kf_plane_dist, pdist + d.kf0, sizeof(double) * d.n_kf);
            for (int l = 0; l < d.n_lm; ++l)  // packed order -> the caller's landmark order
                std::memcpy(W.lm_pos + 3 * (size_t)P.lm_id[d.lm0 + l], lm + 3 * (size_t)(d.lm0 + l), sizeof(double) * 3);
        }
        if (reports) {
            limo_ba_report& r = reports[w];
            const WinState& s = st[w];
            std::memset(&r, 0, sizeof(r));
            r.termination = s.term;
            r.num_solves = s.acc_solves;
            r.iterations_total = s.acc_iters;
            r.iterations_final = s.last_iters;
            r.successful_steps = s.acc_success;
            r.n_depth_blocks = d.n_depth;
            r.n_repr_blocks = d.n_repr;
            r.n_gp_blocks = d.n_gp;
            r.n_trimmed_landmarks = s.n_trimmed;
            r.num_linearizations = s.acc_lin;
            r.initial_cost = s.first_initial_cost;
            r.final_cost = s.solve_final_cost;
            r.time_sec = b->last_solve_sec;
        }
    };
    unsigned nt = std::min({std::thread::hardware_concurrency(), 32u, (unsigned)(P.n_win / 32)});
    if (const char* e = std::getenv("KBA_PACK_THREADS")) nt = (unsigned)std::max(1, std::atoi(e));
    if (nt <= 1) {
        for (int w = 0; w < P.n_win; ++w) one_window(w);
    } else {
        std::atomic<int> next{0};
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&] {
                for (int w0 = next.fetch_add(8); w0 < P.n_win; w0 = next.fetch_add(8))
                    for (int w = w0; w < std::min(w0 + 8, (int)P.n_win); ++w) one_window(w);
            });
        for (auto& th : pool) th.join();
    }
    return LIMO_OK;
}

int limo_ba_batch_trimmed(limo_ba_batch* b, int32_t w, uint8_t* removed) {
    if (!b || !removed || w < 0 || w >= b->P.n_win) return LIMO_ERR_INVALID;
    limo_ctx* ctx = b->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) return LIMO_ERR_NO_DEVICE;
    const WinDesc& d = b->P.win[w];
    std::vector<uint8_t> st((size_t)std::max(1, (int)d.n_lm));
    if (d.n_lm) {
        HIP_TRY(ctx, hipMemcpyAsync(st.data(), b->bv.lm_state + d.lm0, (size_t)d.n_lm, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    for (int l = 0; l < d.n_lm; ++l)  // packed order -> the caller's order; in the problem at create(), out of it now
        removed[b->P.lm_id[d.lm0 + l]] = (b->P.lm_state[d.lm0 + l] != 0 && st[l] == 0) ? 1 : 0;
    return LIMO_OK;
}

void limo_ba_batch_destroy(limo_ba_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    delete b;
}

int limo_ba_batch_kernel_stats(limo_ba_batch* b, int reset, double* linearize_ms, int64_t* linearize_launches,
                               double* total_ms) {
    if (!b) return LIMO_ERR_INVALID;
    int rc = b->collect_linearize_events();
    if (rc != LIMO_OK) return rc;
    if (linearize_ms) *linearize_ms = b->k_ms_acc[LIMO_KERNEL_LINEARIZE];
    if (linearize_launches) *linearize_launches = b->k_launches[LIMO_KERNEL_LINEARIZE];
    if (total_ms) *total_ms = b->total_ms_acc;
    if (reset) {
        for (int k = 0; k < 2; ++k) {
            b->k_ms_acc[k] = 0.0;
            b->k_launches[k] = 0;
        }
        b->total_ms_acc = 0.0;
    }
    return LIMO_OK;
}

int limo_ba_batch_kernel_time(limo_ba_batch* b, int kernel, double* ms, int64_t* launches) {
    if (!b || kernel < 0 || kernel > LIMO_KERNEL_SCHUR) return LIMO_ERR_INVALID;
    int rc = b->collect_linearize_events();
    if (rc != LIMO_OK) return rc;
    if (ms) *ms = b->k_ms_acc[kernel];
    if (launches) *launches = b->k_launches[kernel];
    return LIMO_OK;
}

int limo_comm_unique_id(unsigned char id[LIMO_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) <= LIMO_COMM_ID_BYTES, "ncclUniqueId does not fit");
    if (!id) return LIMO_ERR_INVALID;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return LIMO_ERR_RUNTIME;
    std::memset(id, 0, LIMO_COMM_ID_BYTES);
    std::memcpy(id, &u, sizeof(u));
    return LIMO_OK;
}

int limo_ctx_comm_init(limo_ctx* ctx, const unsigned char id[LIMO_COMM_ID_BYTES], int rank, int world) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return LIMO_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return LIMO_ERR_NO_DEVICE;
    if (ctx->comm) {
        (void)ncclCommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    ncclResult_t r = ncclCommInitRank(&comm, world, u, rank);
    if (r != ncclSuccess) {
        ctx->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
        return LIMO_ERR_RUNTIME;
    }
    ctx->comm = comm;
    ctx->xfn = nullptr;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return LIMO_OK;
}

int limo_ctx_comm_init_host(limo_ctx* ctx, limo_exchange_fn fn, void* user, int rank, int world) {
    if (!ctx || (fn && (world < 1 || rank < 0 || rank >= world))) return LIMO_ERR_INVALID;
    if (ctx->comm) {
        (void)ncclCommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->xfn = fn;
    ctx->xuser = user;
    ctx->comm_rank = fn ? rank : 0;
    ctx->comm_world = fn ? world : 1;
    return LIMO_OK;
}

int limo_ba_solve_sharded(limo_ctx* ctx, limo_ba_window* window, const limo_ba_options* opts, int n_shards,
                          limo_ba_report* report) {
    if (!ctx || !window) return LIMO_ERR_INVALID;
    const bool ranks = ctx->has_transport();
    if (n_shards < 1 || (ranks && n_shards % ctx->comm_world != 0) || n_shards > 8 * (ranks ? ctx->comm_world : 1)) {
        ctx->err = "limo_ba_solve_sharded: n_shards must be a multiple of the communicator size, at most 8 shards per GPU";
        return LIMO_ERR_INVALID;
    }
    PackOptions po;
    po.shards = n_shards;
    limo_ba_batch* b = nullptr;
    int rc = batch_create_impl(ctx, 1, window, opts, po, &b, ranks);
    if (rc != LIMO_OK) return rc;
    rc = limo_ba_batch_solve(b, opts);
    limo_ba_report rep;
    if (rc == LIMO_OK) rc = limo_ba_batch_download(b, window, &rep);
    if (rc == LIMO_OK && report) *report = rep;
    ctx->exchange_stats[0] = b->n_exchanges;
    ctx->exchange_stats[1] = b->exchange_bytes;
    ctx->exchange_stats[2] = rc == LIMO_OK ? rep.iterations_total : 0;
    limo_ba_batch_destroy(b);
    return rc;
}

// One window per call (limo_ba_solve, limo_ba_adjust_pose_only): create, solve, download, destroy.
// KBA_HOST_TRACE=1 prints where the host time of the call goes.
static int solve_one(limo_ctx* ctx, limo_ba_window* window, const limo_ba_options* opts, const PackOptions& po, limo_ba_report* report,
                     const char* what) {
    using Clock = std::chrono::steady_clock;
    static const bool trace = std::getenv("KBA_HOST_TRACE") != nullptr;
    const auto t_a = Clock::now();
    limo_ba_batch* b = nullptr;
    int rc = batch_create_impl(ctx, 1, window, opts, po, &b);
    if (rc != LIMO_OK) return rc;
    const auto t0 = Clock::now();
    rc = limo_ba_batch_solve(b, nullptr);
    const auto t1 = Clock::now();
    if (rc == LIMO_OK) rc = limo_ba_batch_download(b, window, report);
    const auto t2 = Clock::now();
    if (report) report->time_sec = std::chrono::duration<double>(t2 - t0).count();
    limo_ba_batch_destroy(b);
    if (trace) {
        auto us = [](Clock::time_point a, Clock::time_point c) { return std::chrono::duration<double, std::micro>(c - a).count(); };
        std::fprintf(stderr, "[kba] %s: create %.0f us, solve %.0f us, download %.0f us, destroy %.0f us\n", what, us(t_a, t0), us(t0, t1), us(t1, t2),
                     us(t2, Clock::now()));
    }
    return rc;
}

int64_t limo_ctx_coop_fallbacks(const limo_ctx* ctx) { return ctx ? (int64_t)ctx->coop_fallbacks : 0; }

int limo_ctx_exchange_stats(limo_ctx* ctx, int64_t* stats3) {
    if (!ctx || !stats3) return LIMO_ERR_INVALID;
    for (int i = 0; i < 3; ++i) stats3[i] = ctx->exchange_stats[i];
    return LIMO_OK;
}

int limo_ba_solve(limo_ctx* ctx, limo_ba_window* window, const limo_ba_options* opts, limo_ba_report* report) {
    if (!ctx || !window) return LIMO_ERR_INVALID;
    return solve_one(ctx, window, opts, PackOptions(), report, "limo_ba_solve");
}

int limo_ba_adjust_pose_only(limo_ctx* ctx, limo_ba_window* window, const limo_speed_prior* prior,
                             const limo_ba_options* opts, limo_ba_report* report) {
    if (!ctx || !window) return LIMO_ERR_INVALID;
    PackOptions po;
    po.pose_only = true;
    po.prior = prior;
    return solve_one(ctx, window, opts, po, report, "limo_ba_adjust_pose_only");
}

int limo_ba_evaluate(limo_ctx* ctx, const limo_ba_window* window, const limo_ba_options* opts, int apply_loss,
                     double* cost, double* residuals, double* jac_pose, double* jac_lm, uint8_t* valid) {
    if (!ctx || !window) return LIMO_ERR_INVALID;
    PackOptions po;
    po.evaluate_only = true;
    limo_ba_batch* b = nullptr;
    int rc = batch_create_impl(ctx, 1, window, opts, po, &b);
    if (rc != LIMO_OK) return rc;
    const PackedBatch& P = b->P;
    const int M = P.TO, NB = std::max(1, P.n_echunk);
    double* d_cost = nullptr;
    uint8_t* d_valid = nullptr;
    rc = b->dmalloc((void**)&d_cost, sizeof(double) * NB);
    if (rc == LIMO_OK) rc = b->dmalloc((void**)&d_valid, (size_t)std::max(1, M) + 2);
    if (rc == LIMO_OK && P.n_echunk) {
        hipLaunchKernelGGL(k_view_consts_all, dim3(cdiv(P.TV, 256)), dim3(256), 0, ctx->stream, b->bv);
        hipLaunchKernelGGL(k_evaluate, dim3(evaluate_grid(P.n_echunk)), dim3(kBlock), 0, ctx->stream, b->bv, b->c, apply_loss, d_cost, d_valid);
        if (hipGetLastError() != hipSuccess) rc = LIMO_ERR_RUNTIME;
    }
    // device layout (kba_layout.hpp): rows u, v as planes over the observations, the depth row as compact planes over the depth observations
    const size_t SO = (size_t)P.SO, SD = (size_t)P.SD;
    std::vector<double> hr(2 * SO + SD), hjp(12 * SO + 6 * SD), hjl(6 * SO + 3 * SD), hc(NB, 0.0);
    std::vector<uint8_t> hv(std::max(1, M));
    if (rc == LIMO_OK) {
        hipError_t e = hipMemcpyAsync(hr.data(), b->bv.obs_r, sizeof(double) * hr.size(), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(hjp.data(), b->bv.obs_Jp, sizeof(double) * hjp.size(), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(hjl.data(), b->bv.obs_Jl, sizeof(double) * hjl.size(), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && P.n_echunk) e = hipMemcpyAsync(hc.data(), d_cost, sizeof(double) * NB, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(hv.data(), d_valid, std::max(1, M), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            ctx->err = std::string("evaluate: ") + hipGetErrorString(e);
            rc = LIMO_ERR_RUNTIME;
        }
    }
    if (rc == LIMO_OK) {
        double total = 0.0;
        for (int q = 0; q < P.n_echunk; ++q) total += hc[q];
        for (int o = 0; o < M; ++o) {
            const int src = P.obs_src[o];
            if (src < 0) continue;  // (padding of a view's segment)
            const bool dep = P.obs_rank[o] >= 0;
            const size_t rank = dep ? (size_t)P.obs_rank[o] : 0;
            if (valid) valid[src] = hv[o];
            if (residuals) {
                residuals[3 * (size_t)src + 0] = hr[o];
                residuals[3 * (size_t)src + 1] = hr[SO + o];
                residuals[3 * (size_t)src + 2] = dep ? hr[2 * SO + rank] : 0.0;
            }
            if (jac_pose)
                for (int i = 0; i < 6; ++i) {
                    jac_pose[18 * (size_t)src + i] = hjp[(size_t)i * SO + o];
                    jac_pose[18 * (size_t)src + 6 + i] = hjp[(size_t)(6 + i) * SO + o];
                    jac_pose[18 * (size_t)src + 12 + i] = dep ? hjp[12 * SO + (size_t)i * SD + rank] : 0.0;
                }
            if (jac_lm)
                for (int i = 0; i < 3; ++i) {
                    jac_lm[9 * (size_t)src + i] = hjl[(size_t)i * SO + o];
                    jac_lm[9 * (size_t)src + 3 + i] = hjl[(size_t)(3 + i) * SO + o];
                    jac_lm[9 * (size_t)src + 6 + i] = dep ? hjl[6 * SO + (size_t)i * SD + rank] : 0.0;
                }
        }
        if (cost) *cost = total;
    }
    limo_ba_batch_destroy(b);
    return rc;
}

int limo_ba_evaluate_rows(limo_ctx* ctx, const limo_ba_window* window, const limo_speed_prior* prior, int pose_only,
                          const limo_ba_options* opts, int32_t cap, limo_ba_row* rows, int32_t* n_rows) {
    if (!ctx || !window || !n_rows || cap < 0 || (cap > 0 && !rows)) return LIMO_ERR_INVALID;
    PackOptions po;
    po.pose_only = pose_only != 0;
    po.prior = pose_only ? prior : nullptr;
    limo_ba_batch* b = nullptr;
    int rc = batch_create_impl(ctx, 1, window, opts, po, &b);  // the SOLVE problem: ground-plane wiring, regularisers, constness
    if (rc != LIMO_OK) return rc;
    const PackedBatch& P = b->P;
    const WinDesc& wd = P.win[0];
    const int n_reg = reg_row_count(wd), n_gp = wd.n_gp;
    RegRow* d_rows = nullptr;
    int32_t* d_fixed = nullptr;
    rc = b->dmalloc((void**)&d_rows, sizeof(RegRow) * std::max(1, n_reg));
    if (rc == LIMO_OK) rc = b->dmalloc((void**)&d_fixed, sizeof(int32_t) * std::max(1, n_reg));
    std::vector<RegRow> hrows(std::max(1, n_reg));
    std::vector<int32_t> hfixed(std::max(1, n_reg));
    std::vector<double> gr(std::max<int64_t>(1, P.SG)), gF((size_t)std::max<int64_t>(1, P.SG) * 10), gE((size_t)std::max<int64_t>(1, P.SG) * 3), gcost(std::max(1, P.TG));
    if (rc == LIMO_OK) {
        hipLaunchKernelGGL(k_eval_rows, dim3(1), dim3(kBlock), 0, ctx->stream, b->bv, 0, d_rows, d_fixed);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && n_reg) e = hipMemcpyAsync(hrows.data(), d_rows, sizeof(RegRow) * n_reg, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && n_reg) e = hipMemcpyAsync(hfixed.data(), d_fixed, sizeof(int32_t) * n_reg, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && n_gp) e = hipMemcpyAsync(gr.data(), b->bv.gp_r, sizeof(double) * P.SG, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && n_gp) e = hipMemcpyAsync(gF.data(), b->bv.gp_F, sizeof(double) * P.SG * 10, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && n_gp) e = hipMemcpyAsync(gE.data(), b->bv.gp_E, sizeof(double) * P.SG * 3, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && n_gp) e = hipMemcpyAsync(gcost.data(), b->bv.gp_cost, sizeof(double) * P.TG, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            ctx->err = std::string("evaluate_rows: ") + hipGetErrorString(e);
            rc = LIMO_ERR_RUNTIME;
        }
    }
    if (rc == LIMO_OK) *n_rows = rows_from_linearisation(P, 0, gr.data(), gF.data(), gE.data(), gcost.data(), hrows.data(), hfixed.data(), cap, rows);
    limo_ba_batch_destroy(b);
    return rc;
}

int limo_ba_evaluate_batch_time(limo_ctx* ctx, int32_t n, const limo_ba_window* windows, const limo_ba_options* opts, int32_t reps,
                                double* device_ms) {
    if (!ctx || !windows || n <= 0 || reps <= 0 || !device_ms) return LIMO_ERR_INVALID;
    PackOptions po;
    po.evaluate_only = true;
    limo_ba_batch* b = nullptr;
    int rc = batch_create_impl(ctx, n, windows, opts, po, &b);
    if (rc != LIMO_OK) return rc;
    const int M = b->P.TO;
    double* d_cost = nullptr;
    uint8_t* d_valid = nullptr;
    rc = b->dmalloc((void**)&d_cost, sizeof(double) * std::max(1, b->P.n_echunk));
    if (rc == LIMO_OK) rc = b->dmalloc((void**)&d_valid, (size_t)std::max(1, M) + 2);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (rc == LIMO_OK && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = LIMO_ERR_RUNTIME;
    if (rc == LIMO_OK && b->P.n_echunk) {
        hipStream_t s = ctx->stream;
        // (an evaluation = the per-view constants + the materialised pass: both launches are inside the timed region)
        auto eval = [&]() {
            hipLaunchKernelGGL(k_view_consts_all, dim3(cdiv(b->P.TV, 256)), dim3(256), 0, s, b->bv);
            hipLaunchKernelGGL(k_evaluate, dim3(evaluate_grid(b->P.n_echunk)), dim3(kBlock), 0, s, b->bv, b->c, 1, d_cost, d_valid);
        };
        eval();  // warm-up
        (void)hipEventRecord(e0, s);
        for (int r = 0; r < reps; ++r) eval();
        (void)hipEventRecord(e1, s);
        float ms = 0.f;
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
            ctx->err = "limo_ba_evaluate_batch_time: launch failed";
            rc = LIMO_ERR_RUNTIME;
        }
        *device_ms = ms / reps;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    limo_ba_batch_destroy(b);
    return rc;
}

}  // extern "C"
