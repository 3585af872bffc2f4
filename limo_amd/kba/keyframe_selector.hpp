// keyframe_selector.hpp — which frames become keyframes.  Host-side control logic next to the hot path;
// mirrors the public surface of
//   keyframe_bundle_adjustment/include/keyframe_bundle_adjustment/keyframe_selector.hpp:36-86
//   .../internal/keyframe_schemes_base.hpp, keyframe_rejection_scheme_flow.hpp,
//   keyframe_selection_scheme_pose.hpp, keyframe_sparsification_scheme_time.hpp
//   (src/keyframe_selector.cpp:106-133, src/keyframe_*_scheme_*.cpp)
// Decision per frame (KeyframeSelector::select): a frame is dropped when a REJECTION scheme refuses it; otherwise it
// becomes a keyframe when a SELECTION scheme asks for it or every SPARSIFICATION scheme lets it pass.
//
// One deliberate difference: the reference matches the three passes through per-pass running indices and erases while
// iterating (keyframe_selector.cpp:85-103); for the single-frame calls the node makes (mono_lidar.cpp:217-219) that is
// the rule above, and the rule above is what is implemented for any number of frames.
#pragma once
#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <set>
#include <vector>

#include "keyframe.hpp"

namespace keyframe_bundle_adjustment {

struct KeyframeSchemeBase {
    using Ptr = std::shared_ptr<KeyframeSchemeBase>;
    using ConstPtr = std::shared_ptr<const KeyframeSchemeBase>;
    virtual ~KeyframeSchemeBase() = default;
    // may `new_frame` be used, given the keyframes chosen so far?
    virtual bool isUsable(const Keyframe::Ptr& new_frame, const std::map<KeyframeId, Keyframe::Ptr>& last_frames) const = 0;

protected:
    static const Keyframe::Ptr& newest(const std::map<KeyframeId, Keyframe::Ptr>& frames) {
        return std::max_element(frames.cbegin(), frames.cend(),
                                [](const auto& a, const auto& b) { return a.second->timestamp_ < b.second->timestamp_; })
            ->second;
    }
};
struct KeyframeRejectionSchemeBase : KeyframeSchemeBase {
    using Ptr = std::shared_ptr<KeyframeRejectionSchemeBase>;
    using ConstPtr = std::shared_ptr<const KeyframeRejectionSchemeBase>;
};
struct KeyframeSelectionSchemeBase : KeyframeSchemeBase {
    using Ptr = std::shared_ptr<KeyframeSelectionSchemeBase>;
    using ConstPtr = std::shared_ptr<const KeyframeSelectionSchemeBase>;
};
struct KeyframeSparsificationSchemeBase : KeyframeSchemeBase {
    using Ptr = std::shared_ptr<KeyframeSparsificationSchemeBase>;
    using ConstPtr = std::shared_ptr<const KeyframeSparsificationSchemeBase>;
};

// Refuse frames that barely moved in the image: (mean pixel displacement against the newest keyframe)^2 must exceed
// min_median_flow^2 (the reference's variable is called "median" but holds the mean, keyframe_rejection_scheme_flow.cpp:59-66).
class KeyframeRejectionSchemeFlow : public KeyframeRejectionSchemeBase {
public:
    explicit KeyframeRejectionSchemeFlow(double min_median_flow) : min_median_flow_squared_(min_median_flow * min_median_flow) {}
    bool isUsable(const Keyframe::Ptr& new_frame, const std::map<KeyframeId, Keyframe::Ptr>& last_frames) const override {
        if (last_frames.empty()) return true;
        if (new_frame->measurements_.empty()) return false;
        const Keyframe::Ptr& last = newest(last_frames);
        double sum = 0.;
        size_t n = 0;
        // (both keyframes' measurement tables are in (landmark id, camera id) order: one merge pass over two arrays instead of two tree
        // searches per measurement; the terms are added in the order of the statement it replaces - the new frame's landmarks, their cameras)
        Keyframe::MeasurementTableScope table_scope;  // (the table is trusted for the duration of this call: keyframe.hpp)
        const auto& rn = new_frame->measurementTable();
        const auto& rl = last->measurementTable();
        size_t il = 0;
        for (const auto& row : rn) {
            while (il < rl.size() && (rl[il].id < row.id || (rl[il].id == row.id && rl[il].cam < row.cam))) ++il;
            if (il == rl.size()) break;
            if (rl[il].id != row.id || rl[il].cam != row.cam) continue;
            const Measurement& o = *rl[il].m;
            const double du = double(row.m->u) - double(o.u), dv = double(row.m->v) - double(o.v);
            sum += std::sqrt(du * du + dv * dv);
            ++n;
        }
        const double mean = sum / static_cast<double>(n);  // n == 0 -> NaN -> not usable, as in the reference
        return mean * mean > min_median_flow_squared_;
    }
    static KeyframeRejectionSchemeBase::ConstPtr createConst(double f) { return std::make_shared<const KeyframeRejectionSchemeFlow>(f); }
    static KeyframeRejectionSchemeBase::Ptr create(double f) { return std::make_shared<KeyframeRejectionSchemeFlow>(f); }

private:
    double min_median_flow_squared_;
};

// Rotation angle between two poses' quaternions, src/definitions.cpp:104-111 (Eigen AngleAxis of q1^-1 q0).
inline double calcQuaternionDiff(const Pose& p0, const Pose& p1) {
    const double n1 = p1[0] * p1[0] + p1[1] * p1[1] + p1[2] * p1[2] + p1[3] * p1[3];
    const double w1 = p1[0] / n1, x1 = -p1[1] / n1, y1 = -p1[2] / n1, z1 = -p1[3] / n1;  // q1^-1
    const double w = w1 * p0[0] - x1 * p0[1] - y1 * p0[2] - z1 * p0[3];
    const double x = w1 * p0[1] + x1 * p0[0] + y1 * p0[3] - z1 * p0[2];
    const double y = w1 * p0[2] - x1 * p0[3] + y1 * p0[0] + z1 * p0[1];
    const double z = w1 * p0[3] + x1 * p0[2] - y1 * p0[1] + z1 * p0[0];
    const double n = std::sqrt(x * x + y * y + z * z);
    return n != 0. ? 2. * std::atan2(n, std::fabs(w)) : 0.;
}

// Take a frame when the vehicle has turned: rotation against the newest keyframe above a threshold [rad].
class KeyframeSelectionSchemePose : public KeyframeSelectionSchemeBase {
public:
    explicit KeyframeSelectionSchemePose(double critical_quaternion_difference) : critical_quaternion_diff_(critical_quaternion_difference) {}
    bool isUsable(const Keyframe::Ptr& new_frame, const std::map<KeyframeId, Keyframe::Ptr>& last_frames) const override {
        if (last_frames.empty()) return false;  // otherwise the very first frames would always be taken
        return calcQuaternionDiff(new_frame->pose_, newest(last_frames)->pose_) > critical_quaternion_diff_;
    }
    static KeyframeSelectionSchemeBase::ConstPtr createConst(double d) { return std::make_shared<const KeyframeSelectionSchemePose>(d); }
    static KeyframeSelectionSchemeBase::Ptr create(double d) { return std::make_shared<KeyframeSelectionSchemePose>(d); }

private:
    double critical_quaternion_diff_;
};

// Let a frame pass when enough time went by since the newest keyframe (unsigned nanosecond arithmetic, as the reference).
class KeyframeSparsificationSchemeTime : public KeyframeSparsificationSchemeBase {
public:
    explicit KeyframeSparsificationSchemeTime(double time_difference_nano_sec) : time_difference_nano_sec_(time_difference_nano_sec) {}
    bool isUsable(const Keyframe::Ptr& new_frame, const std::map<KeyframeId, Keyframe::Ptr>& last_frames) const override {
        if (last_frames.empty()) return true;
        const TimestampNSec max_ts = newest(last_frames)->timestamp_;
        return static_cast<double>(new_frame->timestamp_ - max_ts) > time_difference_nano_sec_;
    }
    static KeyframeSparsificationSchemeBase::ConstPtr createConst(double t) { return std::make_shared<const KeyframeSparsificationSchemeTime>(t); }
    static KeyframeSparsificationSchemeBase::Ptr create(double t) { return std::make_shared<KeyframeSparsificationSchemeTime>(t); }

private:
    double time_difference_nano_sec_;
};

class KeyframeSelector {
public:
    using Keyframes = std::set<Keyframe::Ptr>;

    void addScheme(KeyframeSelectionSchemeBase::ConstPtr scheme) { selection_schemes_.push_back(scheme); }
    void addScheme(KeyframeRejectionSchemeBase::ConstPtr scheme) { rejection_schemes_.push_back(scheme); }
    void addScheme(KeyframeSparsificationSchemeBase::ConstPtr scheme) { sparsification_schemes_.push_back(scheme); }

    // frames: candidates (the node passes one); buffer_selected_frames: the keyframes already in the optimisation
    Keyframes select(const Keyframes& frames, std::map<KeyframeId, Keyframe::Ptr> buffer_selected_frames) const {
        Keyframe::MeasurementTableScope table_scope;  // (measurement tables are trusted inside one public call: keyframe.hpp)
        // frames in time order; a frame accepted earlier in this call counts as a keyframe for the later ones
        std::vector<Keyframe::Ptr> ordered(frames.begin(), frames.end());
        std::sort(ordered.begin(), ordered.end(), [](const auto& a, const auto& b) { return a->timestamp_ < b->timestamp_; });
        std::map<KeyframeId, Keyframe::Ptr> taken;
        Keyframes out;
        auto all_usable = [&](const auto& schemes, const Keyframe::Ptr& f) {
            for (const auto& s : schemes)
                if (!s->isUsable(f, buffer_selected_frames) || !s->isUsable(f, taken)) return false;
            return true;
        };
        auto any_usable = [&](const auto& schemes, const Keyframe::Ptr& f) {
            for (const auto& s : schemes)
                if (s->isUsable(f, buffer_selected_frames) || s->isUsable(f, taken)) return true;
            return false;
        };
        for (const auto& f : ordered) {
            if (!all_usable(rejection_schemes_, f)) continue;
            if (any_usable(selection_schemes_, f) || all_usable(sparsification_schemes_, f)) {
                out.insert(f);
                taken[f->timestamp_] = f;
            }
        }
        return out;
    }

private:
    std::vector<KeyframeSchemeBase::ConstPtr> selection_schemes_, rejection_schemes_, sparsification_schemes_;
};

}  // namespace keyframe_bundle_adjustment
