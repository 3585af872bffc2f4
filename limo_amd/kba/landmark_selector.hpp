// landmark_selector.hpp — which landmarks enter the next solve.  Same public surface and semantics as
// keyframe_bundle_adjustment/include/keyframe_bundle_adjustment/landmark_selector.hpp:40-345 and
// internal/landmark_selection_scheme_{base,cheirality,random}.hpp; the selection itself stays on the host
// (it defines the INPUT of the hot path).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <iterator>
#include <random>
#include <string>

#include "keyframe.hpp"

namespace keyframe_bundle_adjustment {

// The same landmarks as a LandmarkMap, as a vector sorted by id (entries carry the member names of a map entry).  The selector
// hands the landmarks from scheme to scheme in this form: every std::map / std::set of ~3000 landmarks that the map-based
// interface makes it build costs 0.2 - 0.3 ms of node allocations per solve(), and there were a dozen of them.
struct LandmarkEntry {
    LandmarkId first;
    Landmark::ConstPtr second;
};
using LandmarkView = std::vector<LandmarkEntry>;

class LandmarkSchemeBase {
public:
    using LandmarkMap = std::map<LandmarkId, Landmark::ConstPtr>;
    using KeyframeMap = std::map<KeyframeId, Keyframe::ConstPtr>;
    virtual ~LandmarkSchemeBase() = default;
    virtual std::set<LandmarkId> getSelection(const LandmarkMap& landmarks, const KeyframeMap& keyframes) const = 0;
    // The same selection from / as sorted vectors - what LandmarkSelector calls.  A scheme that only implements the reference's
    // interface above is served by this default (it builds the map for it); the schemes of this package override it.
    virtual std::vector<LandmarkId> getSelectionSorted(const LandmarkView& landmarks, const KeyframeMap& keyframes) const {
        LandmarkMap m;
        for (const auto& e : landmarks) m.insert(m.end(), {e.first, e.second});
        const std::set<LandmarkId> s = getSelection(m, keyframes);
        return std::vector<LandmarkId>(s.begin(), s.end());
    }
    std::string identifier = "";
};
struct LandmarkSelectionSchemeBase : LandmarkSchemeBase {  // landmarks that MUST be taken
    using Ptr = std::shared_ptr<LandmarkSelectionSchemeBase>;
    using ConstPtr = std::shared_ptr<const LandmarkSelectionSchemeBase>;
};
struct LandmarkRejectionSchemeBase : LandmarkSchemeBase {  // landmarks that must NOT be taken
    using Ptr = std::shared_ptr<LandmarkRejectionSchemeBase>;
    using ConstPtr = std::shared_ptr<const LandmarkRejectionSchemeBase>;
};
struct LandmarkSparsificationSchemeBase : LandmarkSchemeBase {  // thin out what is left
    using Ptr = std::shared_ptr<LandmarkSparsificationSchemeBase>;
    using ConstPtr = std::shared_ptr<const LandmarkSparsificationSchemeBase>;
};

// internal/landmark_selection_scheme_base.hpp: schemes may also categorise (near / middle / far field)
class LandmarkCategorizatonInterface {
public:
    enum class Category { NearField, MiddleField, FarField };
    virtual ~LandmarkCategorizatonInterface() = default;
    virtual std::map<LandmarkId, Category> getCategorizedSelection(const LandmarkSchemeBase::LandmarkMap& landmarks,
                                                                   const LandmarkSchemeBase::KeyframeMap& keyframes) const = 0;
    // (id, category) sorted by id, from a sorted vector (see LandmarkSchemeBase::getSelectionSorted)
    virtual std::vector<std::pair<LandmarkId, Category>> getCategorizedSelectionSorted(const LandmarkView& landmarks,
                                                                                      const LandmarkSchemeBase::KeyframeMap& keyframes) const {
        LandmarkSchemeBase::LandmarkMap m;
        for (const auto& e : landmarks) m.insert(m.end(), {e.first, e.second});
        const auto c = getCategorizedSelection(m, keyframes);
        return std::vector<std::pair<LandmarkId, Category>>(c.begin(), c.end());
    }
};

// landmark_selection_scheme_cheirality.cpp:22-60: keep a landmark iff it lies in front (z >= 0) of every camera
// that measured it, in every active keyframe.
class LandmarkRejectionSchemeCheirality : public LandmarkRejectionSchemeBase {
public:
    std::set<LandmarkId> getSelection(const LandmarkMap& landmarks, const KeyframeMap& keyframes) const override {
        const std::vector<LandmarkId> v = kept(landmarks, keyframes);
        return std::set<LandmarkId>(v.begin(), v.end());
    }
    std::vector<LandmarkId> getSelectionSorted(const LandmarkView& landmarks, const KeyframeMap& keyframes) const override {
        return kept(landmarks, keyframes);
    }
    template <class Range>  // a LandmarkMap or a LandmarkView: entries (first = id, second = landmark) in id order
    static std::vector<LandmarkId> kept(const Range& landmarks, const KeyframeMap& keyframes) {
        // Same test as Keyframe::getProjectedLandmarkPosition per (landmark, keyframe) - camera <- vehicle <- origin applied
        // to the landmark, z < 0 rejects - as ONE merge pass per keyframe over its measurements and the landmarks (both
        // sorted by id), the keyframe's and the cameras' transforms formed once: the per-pair std::map the accessor
        // returns, a lookup per pair and a quaternion conversion per pair were 1.3 ms of every solve() at 1500 landmarks.
        std::vector<char> bad(landmarks.size(), 0);
        for (const auto& id_kf : keyframes) {
            const Keyframe& kf = *id_kf.second;
            if (!kf.is_active_) continue;
            const EigenPose T = kf.getEigenPose();
            std::map<CameraId, EigenPose> cam_T;
            for (const auto& c : kf.cameras_) cam_T[c.first] = c.second->getEigenPose();
            Keyframe::MeasurementTableScope table_scope;  // (the table is trusted for the duration of this call: keyframe.hpp)
            const auto& rows = kf.measurementTable();  // (id, camera) rows in id order: the merge touches no map node
            auto il = landmarks.cbegin();
            size_t im = 0, i = 0;
            while (il != landmarks.cend() && im < rows.size()) {
                if (il->first < rows[im].id) {
                    ++il;
                    ++i;
                } else if (rows[im].id < il->first) {
                    ++im;
                } else {
                    const Vector3d p_vehicle = T * Vector3d(il->second->pos.data());
                    for (; im < rows.size() && rows[im].id == il->first; ++im)
                        if ((cam_T.at(rows[im].cam) * p_vehicle).z() < 0.) bad[i] = 1;
                    ++il;
                    ++i;
                }
            }
        }
        std::vector<LandmarkId> out;
        out.reserve(landmarks.size());
        size_t i = 0;
        for (const auto& lm : landmarks) {
            if (!bad[i]) out.push_back(lm.first);
            ++i;
        }
        return out;
    }
    static Ptr create() { return Ptr(new LandmarkRejectionSchemeCheirality()); }
    static ConstPtr createConst() { return ConstPtr(new LandmarkRejectionSchemeCheirality()); }
};

// landmark_selection_scheme_random.cpp:14-33: at most num_landmarks, drawn uniformly (seeded -> deterministic here)
class LandmarkSparsificationSchemeRandom : public LandmarkSparsificationSchemeBase {
public:
    explicit LandmarkSparsificationSchemeRandom(size_t num_landmarks, uint64_t seed = 1) : n_(num_landmarks), seed_(seed) {}
    std::set<LandmarkId> getSelection(const LandmarkMap& landmarks, const KeyframeMap&) const override {
        std::vector<LandmarkId> ids;
        for (const auto& l : landmarks) ids.push_back(l.first);
        std::mt19937_64 g(seed_);
        std::shuffle(ids.begin(), ids.end(), g);
        if (ids.size() > n_) ids.resize(n_);
        return std::set<LandmarkId>(ids.begin(), ids.end());
    }
    static ConstPtr createConst(size_t n) { return ConstPtr(new LandmarkSparsificationSchemeRandom(n)); }

private:
    size_t n_;
    uint64_t seed_;
};

class LandmarkSelector {
public:
    using LandmarkMap = LandmarkSchemeBase::LandmarkMap;
    using KeyframeMap = LandmarkSchemeBase::KeyframeMap;

    void addScheme(LandmarkSelectionSchemeBase::ConstPtr s) { selection_schemes_.push_back(s); }
    void addScheme(LandmarkSparsificationSchemeBase::ConstPtr s) { sparsification_schemes_.push_back(s); }
    void addScheme(LandmarkRejectionSchemeBase::ConstPtr s) { rejection_schemes_.push_back(s); }

    // landmark_selector.hpp:118-253: drop flagged outliers, apply every rejection scheme (set shrinks), collect the
    // must-have selections, sparsify the rest, union, remember what was not selected.
    std::set<LandmarkId> select(const LandmarkMap& landmarks, const KeyframeMap& kfs) {
        LandmarkView view;
        view.reserve(landmarks.size());
        for (const auto& el : landmarks) view.push_back({el.first, el.second});
        return select(view, kfs);
    }
    // The same on a sorted vector of (id, landmark) - what BundleAdjusterKeyframes::solve() calls (no std::map of the active
    // landmarks is built for it).  Between the schemes the landmarks travel as sorted vectors (LandmarkSchemeBase::
    // getSelectionSorted); the decisions are those of the map-based statements, id for id.
    std::set<LandmarkId> select(const LandmarkView& landmarks, const KeyframeMap& kfs) {
        Keyframe::MeasurementTableScope table_scope;  // (measurement tables are trusted inside one public call: keyframe.hpp)
        using clk = std::chrono::steady_clock;
        static const bool trace = std::getenv("LIMO_SHIM_TRACE") != nullptr;
        const auto t0 = clk::now();
        LandmarkView pool;
        if (outlier_ids_.empty()) {
            pool = landmarks;
        } else {
            pool.reserve(landmarks.size());
            auto io = outlier_ids_.cbegin();
            for (const auto& el : landmarks) {
                while (io != outlier_ids_.cend() && *io < el.first) ++io;
                if (io == outlier_ids_.cend() || *io != el.first) pool.push_back(el);
            }
        }
        for (const auto& scheme : rejection_schemes_) pool = restrict(pool, run(scheme, pool, kfs));
        const auto t1 = clk::now();
        std::vector<LandmarkId> must;  // sorted union of the selection schemes' picks (all inside pool)
        for (const auto& scheme : selection_schemes_) {
            const std::vector<LandmarkId> ids = ids_of(restrict(pool, run(scheme, pool, kfs)));
            std::vector<LandmarkId> u;
            u.reserve(must.size() + ids.size());
            std::set_union(must.begin(), must.end(), ids.begin(), ids.end(), std::back_inserter(u));
            must.swap(u);
        }
        const auto t2 = clk::now();
        // thin = pool passed through every sparsification scheme in turn
        LandmarkView thin_store;
        const LandmarkView* thin = &pool;
        for (const auto& scheme : sparsification_schemes_) {
            LandmarkView next = restrict(pool, run(scheme, *thin, kfs));
            thin_store.swap(next);
            thin = &thin_store;
        }
        const auto t3 = clk::now();
        // selection = ids of thin and of the must-have landmarks (both sorted by id: merged)
        std::set<LandmarkId> selection;
        {
            auto ia = thin->cbegin();
            auto ib = must.cbegin();
            while (ia != thin->cend() || ib != must.cend()) {
                if (ib == must.cend() || (ia != thin->cend() && ia->first < *ib)) {
                    selection.insert(selection.end(), ia->first);
                    ++ia;
                } else if (ia == thin->cend() || *ib < ia->first) {
                    selection.insert(selection.end(), *ib);
                    ++ib;
                } else {
                    selection.insert(selection.end(), ia->first);
                    ++ia;
                    ++ib;
                }
            }
        }
        const auto t4 = clk::now();
        if (!kfs.empty()) {
            // markUnselected(id, newest) for every landmark that was not selected, then clean(newest - 10 s)
            // (landmark_selector.hpp:226-252) - as ONE merge pass: landmarks, selection and the bookkeeping entries are all sorted
            // by id; entries last marked more than ten seconds ago are dropped on the way.  (Two std::maps with a lookup each per
            // unselected landmark and a queue of the marks were 0.5 ms of every solve().)
            TimestampNSec newest = 0;
            for (const auto& kf : kfs) newest = std::max(newest, kf.second->timestamp_);
            const TimestampNSec ten_s = convert(TimestampSec(10.));
            const TimestampNSec oldest = newest > ten_s ? newest - ten_s : 0;
            std::vector<Unselected> merged;
            merged.reserve(unselected_.size() + landmarks.size());
            auto is = selection.cbegin();
            auto iu = unselected_.cbegin();
            for (const auto& lm : landmarks) {
                while (is != selection.cend() && *is < lm.first) ++is;
                if (is != selection.cend() && *is == lm.first) continue;
                const LandmarkId id = lm.first;
                for (; iu != unselected_.cend() && iu->id < id; ++iu)
                    if (!(iu->last_seen < oldest)) merged.push_back(*iu);
                if (iu != unselected_.cend() && iu->id == id) {
                    merged.push_back({id, iu->count + 1, newest});
                    ++iu;
                } else {
                    merged.push_back({id, 1u, newest});
                }
            }
            for (; iu != unselected_.cend(); ++iu)
                if (!(iu->last_seen < oldest)) merged.push_back(*iu);
            unselected_.swap(merged);
            unselected_stale_ = true;
        }
        last_selected_lms_ = selection;
        if (trace) {
            auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            std::fprintf(stderr, "[shim] select: rejection %.0f us, must-have %.0f us, sparsification %.0f us, union %.0f us, unselected bookkeeping %.0f us (%zu tracked)\n",
                         us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4), us(t4, clk::now()), unselected_.size());
        }
        return selection;
    }

    void markUnselected(LandmarkId lm_id, TimestampNSec last_time_seen) {
        auto it = std::lower_bound(unselected_.begin(), unselected_.end(), lm_id, [](const Unselected& u, LandmarkId id) { return u.id < id; });
        if (it == unselected_.end() || it->id != lm_id) it = unselected_.insert(it, {lm_id, 0u, last_time_seen});
        it->count += 1;
        it->last_seen = last_time_seen;
        unselected_stale_ = true;
    }
    // Forget the landmarks last marked before `oldest` (landmark_selector.hpp:240-252).
    void clean(TimestampNSec oldest) {
        unselected_.erase(std::remove_if(unselected_.begin(), unselected_.end(), [&](const Unselected& u) { return u.last_seen < oldest; }), unselected_.end());
        unselected_stale_ = true;
    }
    const std::map<LandmarkId, unsigned int>& getUnselectedLandmarks() const {
        if (unselected_stale_) {  // (the map the API returns is built when somebody asks for it)
            unselected_lms_.clear();
            for (const auto& u : unselected_) unselected_lms_.insert(unselected_lms_.end(), {u.id, u.count});
            unselected_stale_ = false;
        }
        return unselected_lms_;
    }
    const std::map<LandmarkId, LandmarkCategorizatonInterface::Category>& getLandmarkCategories() const {
        if (categories_stale_) {  // (the map is built when somebody asks for it, not in every solve())
            landmark_categories_.clear();
            for (const auto& el : categories_) landmark_categories_.insert(landmark_categories_.end(), el);
            categories_stale_ = false;
        }
        return landmark_categories_;
    }
    const std::set<LandmarkId>& getLastSelection() const { return last_selected_lms_; }
    void clearOutliers() { outlier_ids_.clear(); }
    const std::set<LandmarkId>& getOutliers() const { return outlier_ids_; }
    void setOutlier(LandmarkId id) { outlier_ids_.insert(id); }
    void setOutlier(const std::set<LandmarkId>& ids) {
        for (const auto& el : ids) setOutlier(el);
    }

public:
    std::vector<LandmarkSelectionSchemeBase::ConstPtr> selection_schemes_;
    std::vector<LandmarkSparsificationSchemeBase::ConstPtr> sparsification_schemes_;
    std::vector<LandmarkRejectionSchemeBase::ConstPtr> rejection_schemes_;
    std::set<LandmarkId> outlier_ids_;

private:
    template <typename S>
    std::vector<LandmarkId> run(const std::shared_ptr<const S>& scheme, const LandmarkView& lms, const KeyframeMap& kfs) {
        auto cat = std::dynamic_pointer_cast<const LandmarkCategorizatonInterface>(scheme);
        if (!cat) return scheme->getSelectionSorted(lms, kfs);
        categories_ = cat->getCategorizedSelectionSorted(lms, kfs);
        categories_stale_ = true;
        std::vector<LandmarkId> out;
        out.reserve(categories_.size());
        for (const auto& el : categories_) out.push_back(el.first);
        return out;
    }
    // the entries of `all` whose id is in `ids` (both sorted: one merge pass, appended in order)
    static LandmarkView restrict(const LandmarkView& all, const std::vector<LandmarkId>& ids) {
        LandmarkView out;
        out.reserve(std::min(all.size(), ids.size()));
        auto ia = all.cbegin();
        auto ii = ids.cbegin();
        while (ia != all.cend() && ii != ids.cend()) {
            if (ia->first < *ii)
                ++ia;
            else if (*ii < ia->first)
                ++ii;
            else {
                out.push_back(*ia);
                ++ia;
                ++ii;
            }
        }
        return out;
    }
    static std::vector<LandmarkId> ids_of(const LandmarkView& v) {
        std::vector<LandmarkId> out;
        out.reserve(v.size());
        for (const auto& e : v) out.push_back(e.first);
        return out;
    }
    struct Unselected {
        LandmarkId id;
        unsigned int count;      // how often it was passed over
        TimestampNSec last_seen; // stamp of the newest keyframe when it last was
    };
    std::vector<Unselected> unselected_;  // sorted by id
    mutable std::map<LandmarkId, unsigned int> unselected_lms_;
    mutable bool unselected_stale_ = false;
    std::set<LandmarkId> last_selected_lms_;
    std::vector<std::pair<LandmarkId, LandmarkCategorizatonInterface::Category>> categories_;  // of the last categorising scheme, by id
    mutable std::map<LandmarkId, LandmarkCategorizatonInterface::Category> landmark_categories_;  // the same as the map the API returns
    mutable bool categories_stale_ = false;
};

}  // namespace keyframe_bundle_adjustment
