// kba_pack.hpp — host-side flattening of limo_ba_window[] into the batch layout of kba_layout.hpp, and the
// phase orchestration (solveTrimmed schedule) that drives an Executor (HIP kernels, or the test emulator).
//
// Replaces the host logic of BundleAdjusterKeyframes::solve()/adjustPoseOnly() that decides WHICH residuals and
// parameters exist (reference: keyframe_bundle_adjustment/src/bundle_adjuster_keyframes.cpp:498-562 ground-plane
// wiring, :704-736 scale / ground-plane regularisation and constness rules, :740-758 trimming schedule) and the
// outer loop of robust_optimization::solveTrimmed (robust_optimization/src/robust_solving.cpp:140-248).
#pragma once
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "kba_layout.hpp"

namespace kba {

// std::vector whose resize() leaves trivially-constructible elements uninitialised: the big per-observation /
// per-landmark arrays are written exactly once by the (threaded) packing, zero-filling them first only costs a
// serial pass of page faults.
// The big arrays may come out of an ARENA the caller lends to one pack_windows call (the HIP library: a pinned host block the
// context keeps between batches - no page faults on first touch, and the upload is a DMA straight out of where the pack wrote).
// A bump allocator: requests of kPackArenaMin bytes or more are carved from it while it has room, everything else (and everything
// when no arena is lent) comes from the heap; memory of an arena is never freed piecewise - its owner recycles the block when the
// PackedBatch is gone (pack_arena_owns: deallocate() of such a pointer is a no-op).
struct PackArena {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    size_t wanted = 0;  // bytes the arena-sized requests of this pack added up to (what the arena should hold next time)
};
constexpr size_t kPackArenaMin = 64u << 10;
void pack_arena_lend(PackArena* a);              // this thread's pack_windows carves from *a until pack_arena_lend(nullptr)
void pack_arena_register(const void* base, size_t cap, bool add);  // the ranges pack_arena_owns knows (process-wide)
bool pack_arena_owns(const void* p);
void* pack_arena_take(size_t bytes);             // nullptr: no arena lent / no room

template <typename T>
struct default_init_allocator : std::allocator<T> {
    template <typename U>
    struct rebind {
        using other = default_init_allocator<U>;
    };
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes >= kPackArenaMin)
            if (void* p = pack_arena_take(bytes)) return static_cast<T*>(p);
        return static_cast<T*>(::operator new(bytes));
    }
    void deallocate(T* p, size_t) noexcept {
        if (pack_arena_owns(p)) return;
        ::operator delete(p);
    }
    template <typename U>
    void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) {
        ::new (static_cast<void*>(p)) U;
    }
    template <typename U, typename... Args>
    void construct(U* p, Args&&... args) {
        ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
    }
};
template <typename T>
using uvec = std::vector<T, default_init_allocator<T>>;

struct PackedBatch {
    int32_t n_win = 0, TK = 0, TL = 0, TO = 0, TV = 0, TG = 0, n_blk = 0, n_lblk = 0, n_sblk = 0, Vmax = 0;
    int64_t SO = 0, SL = 0, SG = 0;
    int64_t SD = 0;              // evaluate-only batches: stride of the COMPACT depth-row planes (depth observations only, kba_layout.hpp)
    int32_t TD = 0;              // ... and how many depth observations (d > 0) the batch has
    int64_t hcc_total = 0, spart_total = 0, sred_total = 0, camscr_total = 0, lvpart_total = 0, xlv_total = 0;
    bool evaluate_only = false;  // planes hold the full Jacobians instead of the factored form
    int32_t n_shards = 1;        // landmark shards; owner of landmark l = (caller's index of l) mod n_shards
    std::vector<int32_t> blk_owner, lblk_owner, sblk_owner, gp_owner;  // owning shard of every workgroup / gp row
    std::vector<WinDesc> win;
    std::vector<double> pose, pdir, pdist;  // initial parameters
    uvec<double> lm;
    std::vector<int32_t> kf_win, kf_blk0, kf_nblk, kf_gp0, kf_ngp;
    std::vector<uint8_t> cmask, cpresent;
    std::vector<int32_t> cslot;
    uvec<int32_t> lm_gp;
    uvec<int32_t> lm_win, lm_slot;
    uvec<int32_t> lm_id;  // packed landmark -> index in the caller's window
    uvec<double> lm_weight;
    uvec<uint8_t> lm_state;
    std::vector<int32_t> view_kf, view_win;
    std::vector<double> view_cam;
    std::vector<int32_t> blk_view, blk_obs0, blk_n;
    // evaluate-only batches: the wave-sized work items of k_evaluate (kba_layout.hpp:EvalChunk), n_echunk of them
    std::vector<EvalChunk> echunk;
    int32_t n_echunk = 0;
    std::vector<int32_t> obs_rank;  // evaluate-only batches: index of the observation's depth row in the depth planes or -1
    uvec<int32_t> obs_lm;
    uvec<float> obs_u, obs_v, obs_d;
    uvec<int32_t> obs_src;  // packed observation -> index in the caller's window
    std::vector<int32_t> lblk_win, lblk_lm0, lblk_n, sblk_win, sblk_lm0, sblk_n;
    std::vector<int32_t> gp_lm, gp_kf;
    std::vector<double> gp_w;
};

struct PackOptions {
    bool pose_only = false;
    bool evaluate_only = false;  // no problem-build logic (no ground plane / regularisers), every parameter free
    const limo_speed_prior* prior = nullptr;
    int shards = 1;  // > 1: lay the batch out for landmark sharding (no workgroup straddles two shards)
};

// Returns LIMO_OK or a negative limo_status; err receives a message.
int pack_windows(int32_t n, const limo_ba_window* windows, const limo_ba_options& opts, const PackOptions& po,
                 PackedBatch& out, std::string& err);

SolveConsts make_consts(const limo_ba_options& o);

// One ceres-style solve phase / trimming step, implemented by the HIP library and by the test emulator.
struct Executor {
    virtual ~Executor() {}
    // select = 0: every window; 1: windows with do_trim; 2: of those, the ones whose last solve did not
    // reduce the cost (robust_solving.cpp:172-181)
    virtual void solve_init(int max_iter, int select) = 0;
    virtual void linearize() = 0;     // linearise windows that need it + IterationZero / successful-step tail
    virtual int active_count() = 0;   // number of windows still iterating (synchronises)
    virtual void step() = 0;          // trust-region step, candidate evaluation, accept / reject
    virtual void trim() = 0;          // quantile trimming of windows with do_trim
    virtual void expire(int) {}       // wall-clock cap reached: stop every active window (NO_CONVERGENCE)
};

// solveTrimmed schedule over a whole batch (all windows advance in lock-step, finished windows idle).
void run_schedule(Executor& ex, const limo_ba_options& o);

}  // namespace kba
