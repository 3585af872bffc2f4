// bundle_adjuster_keyframes.hpp — the reference's VO-backend class with the HIP library behind it.
//
// Same class name, method names, argument meaning, exceptions and PUBLIC DATA MEMBERS as
// keyframe_bundle_adjustment/include/keyframe_bundle_adjustment/bundle_adjuster_keyframes.hpp:40-335, so callers
// (mono_lidar.cpp:88-373, the gtests) keep working.  What changed: solve() / adjustPoseOnly() no longer build a
// ceres::Problem; they flatten the window and call limo_ba_solve / limo_ba_adjust_pose_only (include/limo_hip.h).
// Results are written back in place into Keyframe::pose_, Keyframe::local_ground_plane_, Landmark::pos.
#pragma once
#include <exception>
#include <string>

#include "keyframe.hpp"
#include "landmark_selector.hpp"

struct limo_ctx;
struct limo_ray;

namespace keyframe_bundle_adjustment {

class BundleAdjusterKeyframes {
public:
    using UPtr = std::unique_ptr<BundleAdjusterKeyframes>;
    using Ptr = std::shared_ptr<BundleAdjusterKeyframes>;

    struct NotEnoughKeyframesException : public std::exception {
        NotEnoughKeyframesException(size_t num_is, size_t num_should_be);
        const char* what() const noexcept override { return msg.c_str(); }
        size_t num_is;
        size_t num_should_be;
        std::string msg;  // (the reference returns a dangling c_str(); we keep the text alive)
    };
    struct KeyframeNotFoundException : public std::exception {
        explicit KeyframeNotFoundException(TimestampNSec timestamp);
        const char* what() const noexcept override { return msg.c_str(); }
        TimestampNSec ts_;
        std::string msg;
    };
    struct OutlierRejectionOptions {  // bundle_adjuster_keyframes.hpp:79-89
        double depth_thres{0.16};
        double reprojection_thres{1.6};
        double depth_quantile{0.95};
        double reprojection_quantile{0.95};
        int num_iterations{1};
    };

    BundleAdjusterKeyframes();
    ~BundleAdjusterKeyframes();
    BundleAdjusterKeyframes(const BundleAdjusterKeyframes&) = delete;
    BundleAdjusterKeyframes& operator=(const BundleAdjusterKeyframes&) = delete;

    void push(const Keyframe& kf);
    void push(Keyframe&& kf);  // (same, without copying the keyframe's measurement maps: for callers that hand the keyframe over)
    void push(const std::vector<Keyframe>& kfs);
    std::string solve();
    void deactivateKeyframes(int min_num_connecting_landmarks = 3, int min_size_optimization_window = 4,
                             int max_size_optimization_window = 20);
    const Keyframe& getKeyframe(TimestampSec timestamp = -1.) const;
    std::map<KeyframeId, Keyframe::Ptr> getActiveKeyframePtrs() const;
    std::map<KeyframeId, Keyframe::ConstPtr> getActiveKeyframeConstPtrs() const;
    std::vector<Keyframe::Ptr> getSortedActiveKeyframePtrs() const;
    std::vector<std::pair<KeyframeId, Keyframe::Ptr>> getSortedIdsWithActiveKeyframePtrs() const;
    std::map<LandmarkId, Landmark::ConstPtr> getActiveLandmarkConstPtrs() const;
    std::map<LandmarkId, Landmark::ConstPtr> getSelectedLandmarkConstPtrs() const;
    std::string adjustPoseOnly(Keyframe&);
    bool calculateLandmark(const Keyframe& kf, const LandmarkId& lId, Vector3d& posAbs);
    bool calculateLandmark(const LandmarkId& lId, Vector3d& posAbs);
    void set_solver_time(double solver_time_sec);
    void updateLabels(const Tracklets& t, double shrubbery_weight = 1.);

    // numbers behind the last report string (not in the reference; handy for tests and drivers)
    struct LastReport {
        int termination = -1, num_solves = 0, iterations_total = 0, n_trimmed_landmarks = 0;
        int n_depth_blocks = 0, n_repr_blocks = 0, n_gp_blocks = 0;
        double initial_cost = 0, final_cost = 0, time_sec = 0;
    } last_report_;

public:
    std::map<KeyframeId, Keyframe::Ptr> keyframes_;
    std::map<LandmarkId, Landmark::Ptr> landmarks_;
    std::set<KeyframeId> active_keyframe_ids_;
    std::set<LandmarkId> active_landmark_ids_;
    std::set<LandmarkId> selected_landmark_ids_;
    OutlierRejectionOptions outlier_rejection_options_;
    std::unique_ptr<LandmarkSelector> landmark_selector_;
    // cityscapes labels, bundle_adjuster_keyframes.hpp:229-255
    std::map<std::string, std::set<int>> labels_{{"outliers", {23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33}},
                                                 {"shrubbery", {21}},
                                                 {"ground", {6, 7, 8, 9, 10}}};

private:
    std::map<LandmarkId, Landmark::ConstPtr> filterLandmarksById(const std::set<LandmarkId>& ids) const;
    void collectRays(const Keyframe& kf, const LandmarkId& lId, std::vector<limo_ray>& rays) const;  // this keyframe's views
    void collectRays(const LandmarkId& lId, std::vector<limo_ray>& rays) const;                      // every active keyframe
    limo_ctx* context();
    // landmark ids a pushed keyframe measures, in id order (the keys of its measurements_ as one contiguous array: the window cut
    // merges them instead of walking the maps node by node); rebuilt from the map whenever it does not look like it any more
    const std::vector<LandmarkId>& measuredIds(const Keyframe& kf);
    limo_ctx* ctx_ = nullptr;
    double solver_time_sec;
};

}  // namespace keyframe_bundle_adjustment
