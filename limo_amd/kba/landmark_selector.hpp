// landmark_selector.hpp — which landmarks enter the next solve.  Same public surface and semantics as
// keyframe_bundle_adjustment/include/keyframe_bundle_adjustment/landmark_selector.hpp:40-345 and
// internal/landmark_selection_scheme_{base,cheirality,random}.hpp; the selection itself stays on the host
// (it defines the INPUT of the hot path).
#pragma once
#include <algorithm>
#include <random>
#include <string>

#include "keyframe.hpp"

namespace keyframe_bundle_adjustment {

class LandmarkSchemeBase {
public:
    using LandmarkMap = std::map<LandmarkId, Landmark::ConstPtr>;
    using KeyframeMap = std::map<KeyframeId, Keyframe::ConstPtr>;
    virtual ~LandmarkSchemeBase() = default;
    virtual std::set<LandmarkId> getSelection(const LandmarkMap& landmarks, const KeyframeMap& keyframes) const = 0;
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
};

// landmark_selection_scheme_cheirality.cpp:22-60: keep a landmark iff it lies in front (z >= 0) of every camera
// that measured it, in every active keyframe.
class LandmarkRejectionSchemeCheirality : public LandmarkRejectionSchemeBase {
public:
    std::set<LandmarkId> getSelection(const LandmarkMap& landmarks, const KeyframeMap& keyframes) const override {
        std::set<LandmarkId> out;
        for (const auto& lm : landmarks) {
            bool ok = true;
            for (const auto& id_kf : keyframes) {
                if (!id_kf.second->is_active_) continue;
                for (const auto& cam_lm : id_kf.second->getProjectedLandmarkPosition(lm))
                    if (cam_lm.second.z() < 0.) ok = false;
            }
            if (ok) out.insert(lm.first);
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
        LandmarkMap pool = landmarks;
        for (const auto& id : outlier_ids_) pool.erase(id);
        for (const auto& scheme : rejection_schemes_) pool = restrict(landmarks, run(scheme, pool, kfs));
        LandmarkMap must;
        for (const auto& scheme : selection_schemes_)
            for (const auto& el : restrict(pool, run(scheme, pool, kfs))) must[el.first] = el.second;
        LandmarkMap thin = pool;
        for (const auto& scheme : sparsification_schemes_) thin = restrict(pool, run(scheme, thin, kfs));
        for (const auto& el : must) thin[el.first] = el.second;
        std::set<LandmarkId> selection;
        for (const auto& el : thin) selection.insert(el.first);
        if (!kfs.empty()) {
            TimestampNSec newest = 0;
            for (const auto& kf : kfs) newest = std::max(newest, kf.second->timestamp_);
            for (const auto& lm : landmarks)
                if (!selection.count(lm.first)) markUnselected(lm.first, newest);
            const TimestampNSec ten_s = convert(TimestampSec(10.));
            clean(newest > ten_s ? newest - ten_s : 0);
        }
        last_selected_lms_ = selection;
        return selection;
    }

    void markUnselected(LandmarkId lm_id, TimestampNSec last_time_seen) {
        unselected_lms_[lm_id] += 1;
        last_time_seen_[lm_id] = last_time_seen;
    }
    void clean(TimestampNSec oldest) {
        for (auto it = last_time_seen_.begin(); it != last_time_seen_.end();) {
            if (it->second < oldest) {
                unselected_lms_.erase(it->first);
                it = last_time_seen_.erase(it);
            } else {
                ++it;
            }
        }
    }
    const std::map<LandmarkId, unsigned int>& getUnselectedLandmarks() const { return unselected_lms_; }
    const std::map<LandmarkId, LandmarkCategorizatonInterface::Category>& getLandmarkCategories() const {
        return landmark_categories_;
    }
    std::set<LandmarkId> getLastSelection() const { return last_selected_lms_; }
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
    std::set<LandmarkId> run(const std::shared_ptr<const S>& scheme, const LandmarkMap& lms, const KeyframeMap& kfs) {
        auto cat = std::dynamic_pointer_cast<const LandmarkCategorizatonInterface>(scheme);
        if (!cat) return scheme->getSelection(lms, kfs);
        landmark_categories_ = cat->getCategorizedSelection(lms, kfs);
        std::set<LandmarkId> out;
        for (const auto& el : landmark_categories_) out.insert(el.first);
        return out;
    }
    static LandmarkMap restrict(const LandmarkMap& all, const std::set<LandmarkId>& ids) {
        LandmarkMap out;
        for (const auto& id : ids) {
            auto it = all.find(id);
            if (it != all.end()) out[id] = it->second;
        }
        return out;
    }
    std::map<LandmarkId, unsigned int> unselected_lms_;
    std::map<LandmarkId, TimestampNSec> last_time_seen_;
    std::set<LandmarkId> last_selected_lms_;
    std::map<LandmarkId, LandmarkCategorizatonInterface::Category> landmark_categories_;
};

}  // namespace keyframe_bundle_adjustment
