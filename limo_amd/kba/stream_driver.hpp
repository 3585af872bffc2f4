// stream_driver.hpp — ROS-free streaming driver: the per-frame call order of the reference's node
//   keyframe_bundle_adjustment_ros_tool/src/mono_lidar/mono_lidar.cpp:88-373 (MonoLidar::callbackSubscriber)
// with the depth assignment of the (separate) tracklets_depth node in front of it, on the MI355X library:
//   LiDAR sweep + tracked features of the frame
//     -> limo_depth_estimate            FeaturePoint::d of every track's newest point (include/limo_hip.h)
//     -> motion prior                   external (the node: tf), else the five-point direction scaled by the last keyframes'
//                                       speed (:157-186, five_point.hpp; StreamParams::motion_prior) or - the default here -
//                                       constant velocity from the last two poses
//     -> Keyframe(prior) + adjustPoseOnly            (:192-211)
//     -> KeyframeSelector::select                    (:218-222)
//     -> push                                        (:231-233)
//     -> deactivateKeyframes + updateLabels + solve  (:245-263, every time_between_keyframes)
//     -> pose of the frame in KITTI form             (:281-294: camera_0 <- camera_t, 12 numbers per row)
// The very first frame is a Pose-fixed keyframe at the identity (:305-326).
// Everything numerical happens behind the C-ABI; this file is host-side control flow only.
#pragma once
#include <array>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <ostream>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/limo_hip.h"
#include "bundle_adjuster_keyframes.hpp"
#include "five_point.hpp"
#include "keyframe_selector.hpp"
#include "landmark_selection_voxel.hpp"

namespace keyframe_bundle_adjustment {

struct StreamParams {
    // optimisation window (MonoLidar.rosif / keyframe_ba_monolid.launch)
    int max_size_optimization_window = 5;
    int min_number_connecting_landmarks = 3;
    double time_between_keyframes_sec = 0.09;   // solve at most this often (:245), also the time sparsification scheme
    double critical_rotation_difference = 0.1;  // KeyframeSelectionSchemePose
    double min_median_flow = 3.;                // KeyframeRejectionSchemeFlow
    double shrubbery_weight = 0.9;
    double height_over_ground = 0.31;           // Plane::distance of new keyframes (:191)
    double solver_time_sec = 20.;               // wall-clock cap of a solve; <= 0: none (deterministic)
    double prior_speed = 11.;                   // m/s along the vehicle's x axis while no motion has been estimated yet (the
                                                // node scales its five-point direction by interface_.prior_speed, :160-165)
    // Motion prior of a frame that comes without an external one (mono_lidar.cpp:157-186): the node takes the direction from
    // the five-point algorithm on the matches between the last keyframe and the frame and the length from the speed of the last
    // two keyframes (five_point.hpp restates it); constant velocity from the last two frame poses is this driver's default (no
    // RANSAC in the frame loop; adjustPoseOnly refines either against the fixed landmarks).
    enum class MotionPrior { ConstantVelocity, FivePoint };
    MotionPrior motion_prior = MotionPrior::ConstantVelocity;
    // landmark selection (keyframe_ba_monolid.launch:36-38; wiring of MonoLidar::reconfigureRequest, mono_lidar.cpp:396-430)
    unsigned max_number_landmarks_near_bin = 200, max_number_landmarks_middle_bin = 200, max_number_landmarks_far_bin = 100;
    double roi_middle = 15., roi_far = 40.;                            // :405-408
    std::array<double, 3> voxel_size_xyz{{0.5, 0.5, 0.3}};             // :404
    int add_ground_landmarks_per_keyframe = 50;                        // :421-423: for EVERY index of the window the 50
                                                                       // nearest ground-plane landmarks are kept
    int add_depth_landmarks_newest = 0;                                // the node has this rule commented out (:412-416)
    // depth assignment
    bool assign_depth = true;
    // How the depth assignment of a frame announced with announceNextFrame runs ahead of the frame before it (the reference's depth
    // estimator is a process of its own beside the BA node, kitti_standalone.launch:14-23, :55: it works on frame t+1 while the
    // bundle adjuster works on frame t):
    //   Thread  a thread of this driver - the "depth node" - does all of it (limo_depth_estimate on the driver's own context, the
    //           per-track depth history, the depth of every point of the tracker's message as one array the calling thread copies
    //           into the message when it takes the frame) while the calling thread runs frame t
    //   Stream  the calling thread starts it (limo_depth_estimate_begin: copies and kernels on the driver's own stream) and collects
    //           it at the start of the next process() call (limo_depth_estimate_end): no second thread, only the GPU time is hidden
    //   None    announceNextFrame is ignored
    // The results are the same bits in all three.
    enum class DepthAhead { None, Stream, Thread };
    DepthAhead depth_ahead = DepthAhead::Thread;
    int image_width = 1241, image_height = 376;
    std::vector<int> ground_labels{6, 7, 8, 9, 10};  // cityscapes labels treated as ground (labels_["ground"])
};

class StreamDriver {
public:
    struct Stats {
        int frames = 0, keyframes = 0, solves = 0, solves_on_non_keyframes = 0, features = 0, features_with_depth = 0, depth_prefetched = 0;
        double sec_depth = 0., sec_pose_only = 0., sec_push = 0., sec_solve = 0., sec_total = 0.;
        // host-side bookkeeping of a frame: the Keyframe object of the frame (measurement maps of its tracks), keyframe
        // selection, window cut + labels; and of sec_pose_only / sec_solve the share spent inside the C-ABI calls
        double sec_keyframe = 0., sec_select = 0., sec_window = 0., sec_abi_pose_only = 0., sec_abi_solve = 0., sec_depth_history = 0.;
        double sec_depth_thread = 0.;  // time the depth thread worked (beside the calling thread: not part of sec_total)
    };

    StreamDriver(const StreamParams& p, Camera::Ptr camera, const EigenPose& T_camera_lidar) : p_(p), camera_(camera), T_cam_lidar_(T_camera_lidar) {
        ba_.set_solver_time(p.solver_time_sec);
        selector_.addScheme(KeyframeRejectionSchemeFlow::createConst(p.min_median_flow));
        selector_.addScheme(KeyframeSelectionSchemePose::createConst(p.critical_rotation_difference));
        selector_.addScheme(KeyframeSparsificationSchemeTime::createConst(p.time_between_keyframes_sec * 1e9));
        LandmarkSparsificationSchemeVoxel::Parameters vp;
        vp.max_num_landmarks_near = p.max_number_landmarks_near_bin;
        vp.max_num_landmarks_middle = p.max_number_landmarks_middle_bin;
        vp.max_num_landmarks_far = p.max_number_landmarks_far_bin;
        vp.voxel_size_xyz = p.voxel_size_xyz;
        vp.roi_far_xyz = {{p.roi_far, p.roi_far, p.roi_far}};
        vp.roi_middle_xyz = {{p.roi_middle, p.roi_middle, p.roi_middle}};
        ba_.landmark_selector_->addScheme(LandmarkSparsificationSchemeVoxel::createConst(vp));
        LandmarkSelectionSchemeAddDepth::Parameters ap;  // "depth assurance" (:409-426)
        if (p.add_depth_landmarks_newest > 0)
            ap.params_per_keyframe.push_back(std::make_tuple(0, p.add_depth_landmarks_newest, [](const Landmark::ConstPtr& lm) { return lm->has_measured_depth; },
                                                             [](const Measurement& m, const Vector3d&) { return m.d; }));
        for (int i = 0; i < p.max_size_optimization_window && p.add_ground_landmarks_per_keyframe > 0; ++i)
            ap.params_per_keyframe.push_back(std::make_tuple(i, p.add_ground_landmarks_per_keyframe, [](const Landmark::ConstPtr& lm) { return lm->is_ground_plane; },
                                                             [](const Measurement&, const Vector3d& local) { return (float)local.norm(); }));
        ba_.landmark_selector_->addScheme(LandmarkSelectionSchemeAddDepth::createConst(ap));
        int dev = 0;  // LIMO_DEVICE: the GPU of this sequence (one process per GPU and sequence, scripts/stream_replicas.py)
        if (const char* e = std::getenv("LIMO_DEVICE")) dev = std::atoi(e);
        if (limo_ctx_create(dev, &ctx_) != LIMO_OK) throw std::runtime_error("StreamDriver: no HIP device (limo_ctx_create)");
        limo_depth_default_params(&depth_params_);
    }
    ~StreamDriver() {
        try {
            cancelPrefetch();
        } catch (...) {  // (an error of a depth job nobody picked up)
        }
        if (worker_.joinable()) {
            {
                std::lock_guard<std::mutex> lk(job_mutex_);
                job_state_ = JobState::Exit;
            }
            job_cv_.notify_all();
            worker_.join();
        }
        if (ctx_) limo_ctx_destroy(ctx_);
    }
    StreamDriver(const StreamDriver&) = delete;
    StreamDriver& operator=(const StreamDriver&) = delete;

    // Start the depth assignment of a frame that process() will be given LATER (normally the next one): copies and kernels go onto
    // the stream of this driver's own context and run while the host thread and the bundle adjuster's context work on the current
    // frame - the reference's depth estimator is a process of its own beside the BA node (kitti_standalone.launch:14-23, :55).
    // `cloud_xyzi` must stay valid until that process() call (page-locked memory from limo_host_alloc makes the copy a DMA).
    // process() recognises the frame by its stamp and feature count; a prefetch that is not picked up is dropped.  Results are
    // those of the unprefetched call, bit for bit.
    void prefetchDepth(const Tracklets& tracklets, const float* cloud_xyzi, size_t n_pts) {
        if (!p_.assign_depth || !cloud_xyzi || !n_pts || tracklets.tracks.empty() || tracklets.stamps.empty()) return;
        using clk = std::chrono::steady_clock;
        const auto t0 = clk::now();
        closeOpenDepthCall();
        stageFeatures(tracklets);
        const Pose T = convert(T_cam_lidar_);
        const int rc = limo_depth_estimate_begin(ctx_, cloud_xyzi, n_pts, T.data(), camera_->focal_length, camera_->principal_point[0],
                                                 camera_->principal_point[1], p_.image_width, p_.image_height, uv_.data(), tracklets.tracks.size(),
                                                 ground_.data(), &depth_params_);
        if (rc != LIMO_OK) throw std::runtime_error(std::string("limo_depth_estimate_begin: ") + limo_last_error(ctx_));
        prefetch_open_ = true;
        prefetch_stamp_ = tracklets.stamps.front();
        prefetch_n_ = tracklets.tracks.size();
        stats_.sec_depth += std::chrono::duration<double>(clk::now() - t0).count();
    }

    // The frame after the one the NEXT process() call is given: that call starts its depth assignment (StreamParams::depth_ahead)
    // as soon as it has finished its own - the depths of a track's older points come from the frames before, so the frames'
    // assignments run in order.  `tracklets` is the very object that will be handed to process() afterwards (an rvalue of it):
    // the depth thread reads it; it and the sweep must stay valid and untouched until then.
    void announceNextFrame(const Tracklets& tracklets, const float* cloud_xyzi, size_t n_pts) {
        next_tracklets_ = &tracklets;
        next_cloud_ = cloud_xyzi;
        next_n_pts_ = n_pts;
    }

    // Thread mode: returns when the depth thread has finished the frame it was given (a caller that times process() calls puts the
    // thread's work inside the timed region with it; process() waits by itself otherwise).
    void waitForDepthAhead() { waitForDepthThread(); }

    // Drop what prefetchDepth / announceNextFrame started (the sweep buffer is about to go away).
    void cancelPrefetch() {
        next_tracklets_ = nullptr;
        waitForDepthThread();
        ahead_valid_ = false;
        closeOpenDepthCall();
    }

    // One frame.  `tracklets`: the tracker's message - stamps[0] = this frame, every track's feature_points[0] = its point
    // in this frame (newest first).  Depths the caller already knows stay; the others (d < 0) are filled from the sweep:
    // point 0 by limo_depth_estimate on `cloud_xyzi` (KITTI velodyne layout, n_pts x 4 floats; may be null), points 1..
    // from what this driver assigned to the same track in earlier frames.  external_prior: keyframe <- origin pose of the
    // frame if the caller has one (the node's tf prior), else constant velocity.
    // Returns the frame's pose estimate (keyframe <- origin).
    EigenPose process(const Tracklets& message, const float* cloud_xyzi, size_t n_pts, const EigenPose* external_prior = nullptr) {
        return process(Tracklets(message), cloud_xyzi, n_pts, external_prior);
    }
    EigenPose process(Tracklets&& message, const float* cloud_xyzi, size_t n_pts, const EigenPose* external_prior = nullptr) {
        using clk = std::chrono::steady_clock;
        const auto t_begin = clk::now();
        waitForDepthThread();  // (it may be writing into `message`)
        Tracklets tracklets(std::move(message));
        if (tracklets.stamps.empty()) return last_pose_;  // (:98-104)
        // what the depth thread prepared belongs to this message iff stamp, track count and point count agree (a copy of the announced
        // message is as good as the object itself; anything else is assigned here and the prepared depths are dropped)
        size_t n_points = 0;
        for (const auto& tr : tracklets.tracks) n_points += tr.feature_points.size();
        const bool depth_done = ahead_valid_ && ahead_stamp_ == tracklets.stamps.front() && ahead_tracks_ == tracklets.tracks.size() && point_d_.size() == n_points;
        ahead_valid_ = false;
        const TimestampNSec stamp = tracklets.stamps.front();
        if (depth_done) {  // computed by the depth thread: into the message
            applyDepths(tracklets, point_d_);
            ++stats_.depth_prefetched;
        } else {
            assignDepth(tracklets, cloud_xyzi, n_pts);
        }
        stats_.sec_depth += std::chrono::duration<double>(clk::now() - t_begin).count();
        if (next_tracklets_) {  // announceNextFrame
            const Tracklets* next = next_tracklets_;
            next_tracklets_ = nullptr;
            if (p_.depth_ahead == StreamParams::DepthAhead::Stream)
                prefetchDepth(*next, next_cloud_, next_n_pts_);
            else if (p_.depth_ahead == StreamParams::DepthAhead::Thread)
                startDepthThreadJob(next, next_cloud_, next_n_pts_);
        }
        Plane ground_plane;
        ground_plane.distance = p_.height_over_ground;
        EigenPose pose = EigenPose::Identity();
        bool is_keyframe = false;
        if (ba_.keyframes_.empty()) {  // first frame: Pose-fixed keyframe at the origin (:305-326)
            Keyframe first(stamp, tracklets, camera_, EigenPose::Identity(), Keyframe::FixationStatus::Pose, ground_plane);
            first.freezeMeasurements();  // (this driver owns its keyframes and never edits their measurements_: the table may outlive the call)
            ba_.push(std::move(first));
            is_keyframe = true;
        } else {
            EigenPose prior = external_prior ? *external_prior
                              : p_.motion_prior == StreamParams::MotionPrior::FivePoint ? fivePointPrior(stamp, tracklets) : constantVelocityPrior(stamp);
            {   // a product of estimated transforms: back onto SO(3) (convert() does not normalise, definitions.cpp:14-28)
                Pose q = convert(prior);
                const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                for (int i = 0; i < 4; ++i) q[i] /= n;
                prior = convert(q);
            }
            const auto t_kf = clk::now();
            auto cur = std::make_shared<Keyframe>(stamp, tracklets, camera_, prior, Keyframe::FixationStatus::None, ground_plane);
            cur->freezeMeasurements();  // (as above: built once per keyframe, kept for its life in the window)
            stats_.sec_keyframe += std::chrono::duration<double>(clk::now() - t_kf).count();
            if (!external_prior) {  // a prior without scale is always refined against the fixed landmarks of the last
                                    // selection (:200-211; before the first solve() that selection is empty)
                const auto t0 = clk::now();
                ba_.adjustPoseOnly(*cur);
                stats_.sec_pose_only += std::chrono::duration<double>(clk::now() - t0).count();
                stats_.sec_abi_pose_only += ba_.last_report_.time_sec;
                trace("pose-only", stamp, cur->pose_, ba_.last_report_.final_cost);
            }
            pose = cur->getEigenPose();
            const auto t_sel = clk::now();
            const auto selected = selector_.select({cur}, ba_.getActiveKeyframePtrs());
            const auto t1 = clk::now();
            stats_.sec_select += std::chrono::duration<double>(t1 - t_sel).count();
            for (const auto& kf : selected) {
                if (kf == cur)  // (this frame's keyframe object is not looked at again below: hand it over instead of copying its maps)
                    ba_.push(std::move(*cur));
                else
                    ba_.push(*kf);
            }
            stats_.sec_push += std::chrono::duration<double>(clk::now() - t1).count();
            is_keyframe = !selected.empty();
            const double now_sec = convert(stamp);
            // bundle adjustment every time_between_keyframes, whether or not THIS frame became a keyframe (:245-246)
            if (ba_.keyframes_.size() > 2 && now_sec - last_solved_sec_ > 0.98 * p_.time_between_keyframes_sec) {
                const auto t_w = clk::now();
                ba_.deactivateKeyframes(p_.min_number_connecting_landmarks, 3, p_.max_size_optimization_window);
                const auto t_w1 = clk::now();
                ba_.updateLabels(tracklets, p_.shrubbery_weight);
                const auto t2 = clk::now();
                if (std::getenv("LIMO_SHIM_TRACE"))
                    std::fprintf(stderr, "[shim] window cut %.0f us, labels %.0f us\n", std::chrono::duration<double, std::micro>(t_w1 - t_w).count(),
                                 std::chrono::duration<double, std::micro>(t2 - t_w1).count());
                stats_.sec_window += std::chrono::duration<double>(t2 - t_w).count();
                last_summary_ = ba_.solve();
                stats_.sec_solve += std::chrono::duration<double>(clk::now() - t2).count();
                stats_.sec_abi_solve += ba_.last_report_.time_sec;
                ++stats_.solves;
                trace("solve", stamp, ba_.getKeyframe().pose_, ba_.last_report_.final_cost);
                stats_.solves_on_non_keyframes += !is_keyframe;
                last_solved_sec_ = now_sec;
            }
            // the optimised pose if the frame became a keyframe, its prior otherwise (:281-294)
            if (ba_.getKeyframe().timestamp_ == stamp) pose = ba_.getKeyframe().getEigenPose();
        }
        if (have_last_) last_motion_ = pose * last_pose_.inverse();
        have_motion_ = have_last_;
        last_pose_ = pose;
        last_stamp_ = stamp;
        have_last_ = true;
        ++stats_.frames;
        stats_.keyframes += is_keyframe;
        poses_.push_back(pose);
        stats_.sec_total += std::chrono::duration<double>(clk::now() - t_begin).count();
        return pose;
    }

    // KITTI odometry row of a frame (helpers::poseToString, mono_lidar.cpp:281-294): camera_0 <- camera_t =
    // T_cam_veh * (keyframe <- origin)^-1 * T_cam_veh^-1, first three rows, row-major, 12 numbers.
    void writeKittiPose(std::ostream& os, const EigenPose& pose_kf_origin) const {
        const EigenPose cv = camera_->getEigenPose();
        const EigenPose m = cv * pose_kf_origin.inverse() * cv.inverse();
        char buf[512];
        std::snprintf(buf, sizeof(buf), "%.12g %.12g %.12g %.12g %.12g %.12g %.12g %.12g %.12g %.12g %.12g %.12g", m.R[0], m.R[1], m.R[2], m.t[0],
                      m.R[3], m.R[4], m.R[5], m.t[1], m.R[6], m.R[7], m.R[8], m.t[2]);
        os << buf << "\n";
    }
    void writeKittiTrajectory(std::ostream& os) const {
        for (const auto& p : poses_) writeKittiPose(os, p);
    }

    BundleAdjusterKeyframes& adjuster() { return ba_; }
    const std::vector<EigenPose>& poses() const { return poses_; }
    const Stats& stats() const { return stats_; }
    const five_point::Motion& lastFivePoint() const { return last_five_point_; }
    const std::string& lastSummary() const { return last_summary_; }
    limo_depth_params& depthParams() { return depth_params_; }

private:
    // LIMO_STREAM_TRACE=1: one line per adjustPoseOnly / solve with the resulting pose at full precision (two drives on
    // different back ends are compared call by call with it)
    void trace(const char* what, TimestampNSec stamp, const Pose& p, double cost) const {
        static const bool on = std::getenv("LIMO_STREAM_TRACE") != nullptr;
        if (on) {
            uint64_t h = 1469598103934665603ull;  // FNV-1a over the selected ids: two drives select the same SET iff the hashes agree
            for (const auto& id : ba_.selected_landmark_ids_) h = (h ^ (uint64_t)id) * 1099511628211ull;
            std::fprintf(stderr, "trace %s %llu cost %.17g pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g selected %zu set %016llx\n", what,
                         (unsigned long long)stamp, cost, p[0], p[1], p[2], p[3], p[4], p[5], p[6], ba_.selected_landmark_ids_.size(), (unsigned long long)h);
        }
    }
    // mono_lidar.cpp:157-186: direction of the motion since the last keyframe from the five-point algorithm (straight ahead
    // when the image flow is too small for it), length = speed of the last two keyframes x time since the last one
    // (prior_speed while there is only one keyframe), applied to the last keyframe's pose.
    EigenPose fivePointPrior(TimestampNSec stamp, const Tracklets& tracklets) {
        const Keyframe& last_kf = ba_.getKeyframe();
        EigenPose motion = five_point::motionUnscaled(camera_->focal_length, camera_->principal_point, stamp, last_kf.timestamp_, tracklets,
                                                      camera_->getEigenPose(), p_.prior_speed, (uint64_t)stamp, &last_five_point_);
        const auto kfs = ba_.getSortedActiveKeyframePtrs();
        if (kfs.size() > 1) {
            const Keyframe &k1 = *kfs[kfs.size() - 1], &k0 = *kfs[kfs.size() - 2];
            const double speed = (k1.getEigenPose() * k0.getEigenPose().inverse()).translation().norm() / (convert(k1.timestamp_) - convert(k0.timestamp_));
            const double n = std::sqrt(motion.t[0] * motion.t[0] + motion.t[1] * motion.t[1] + motion.t[2] * motion.t[2]);
            for (int i = 0; i < 3; ++i) motion.t[i] = motion.t[i] / std::max(0.0001, n) * speed * (convert(stamp) - convert(k1.timestamp_));
        }
        return motion * (kfs.empty() ? last_kf.getEigenPose() : kfs.back()->getEigenPose());
    }
    // constant velocity from the last two poses; before there are two: straight ahead at prior_speed
    EigenPose constantVelocityPrior(TimestampNSec stamp) const {
        if (have_motion_) return last_motion_ * last_pose_;
        EigenPose m = EigenPose::Identity();
        m.t[0] = -p_.prior_speed * (convert(stamp) - convert(last_stamp_));  // new vehicle <- old vehicle
        return m * last_pose_;
    }

    // pixel coordinates and ground flags of the newest point of every track, as limo_depth_estimate wants them
    void stageFeatures(const Tracklets& ts) {
        const size_t n = ts.tracks.size();
        uv_.resize(2 * n);
        ground_.resize(n);
        for (size_t i = 0; i < n; ++i) {
            uv_[2 * i] = ts.tracks[i].feature_points[0].u;
            uv_[2 * i + 1] = ts.tracks[i].feature_points[0].v;
            ground_[i] = 0;
            for (int l : p_.ground_labels) ground_[i] |= ts.tracks[i].label == l;
        }
    }

    // ---- the depth thread (StreamParams::DepthAhead::Thread): one job at a time = assignDepth of the announced frame
    enum class JobState { Idle, Pending, Running, Exit };
    void startDepthThreadJob(const Tracklets* ts, const float* cloud, size_t n_pts) {
        if (!worker_.joinable()) worker_ = std::thread([this] { depthThreadMain(); });
        {
            std::lock_guard<std::mutex> lk(job_mutex_);
            job_ts_ = ts;
            job_cloud_ = cloud;
            job_n_pts_ = n_pts;
            job_error_ = nullptr;
            job_state_ = JobState::Pending;
        }
        job_cv_.notify_all();
    }
    // returns when no job is pending or running; a job that has finished leaves its frame's identity in ahead_* (its error is rethrown here)
    void waitForDepthThread() {
        if (!worker_.joinable()) return;
        std::unique_lock<std::mutex> lk(job_mutex_);
        job_cv_.wait(lk, [this] { return job_state_ == JobState::Idle; });
        if (job_ts_) {  // (the frame's identity was taken by the thread when it STARTED the job: the caller's message may be gone by now)
            ahead_valid_ = job_has_stamp_;
            ahead_stamp_ = job_stamp_;
            ahead_tracks_ = job_tracks_;
            job_ts_ = nullptr;
        }
        if (job_error_) {
            std::exception_ptr e = job_error_;
            job_error_ = nullptr;
            ahead_valid_ = false;
            std::rethrow_exception(e);
        }
    }
    void depthThreadMain() {
        std::unique_lock<std::mutex> lk(job_mutex_);
        for (;;) {
            job_cv_.wait(lk, [this] { return job_state_ == JobState::Pending || job_state_ == JobState::Exit; });
            if (job_state_ == JobState::Exit) return;
            job_state_ = JobState::Running;
            const Tracklets* ts = job_ts_;
            const float* cloud = job_cloud_;
            const size_t n_pts = job_n_pts_;
            job_has_stamp_ = !ts->stamps.empty();
            job_stamp_ = job_has_stamp_ ? ts->stamps.front() : 0;
            job_tracks_ = ts->tracks.size();
            lk.unlock();
            std::exception_ptr err;
            const auto t0 = std::chrono::steady_clock::now();
            try {
                if (!ts->stamps.empty()) computeDepths(*ts, cloud, n_pts, point_d_);
            } catch (...) {
                err = std::current_exception();
            }
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            lk.lock();
            stats_.sec_depth_thread += sec;
            job_error_ = err;
            if (job_state_ != JobState::Exit) job_state_ = JobState::Idle;
            job_cv_.notify_all();
        }
    }

    void closeOpenDepthCall() {  // a prefetch nobody will pick up: its limo_depth_estimate_end, results dropped
        if (!prefetch_open_) return;
        prefetch_open_ = false;
        std::vector<float> sink(prefetch_n_);
        (void)limo_depth_estimate_end(ctx_, sink.data(), prefetch_n_);
    }

    // FeaturePoint::d of the tracks: newest point from the sweep (limo_depth_estimate), history from earlier frames.
    void assignDepth(Tracklets& ts, const float* cloud, size_t n_pts) {
        computeDepths(ts, cloud, n_pts, point_d_);
        applyDepths(ts, point_d_);
    }
    // ... in two steps: computeDepths reads the message and writes the depth of EVERY point of every track, in message order, into
    // `out`; applyDepths copies them into the message.  (The depth thread does the first step only: what it hands over is one
    // contiguous array - had it written into the message, the calling thread would later fetch every track's points from the
    // other core's cache, which cost as much as the thread saved.)
    void computeDepths(const Tracklets& ts, const float* cloud, size_t n_pts, std::vector<float>& out) {
        const size_t n = ts.tracks.size();
        // A frame may come here twice (announced to the depth thread, then handed to process() with another message for the same
        // stamp, or announced and never processed): the statistics count a stamp once and the per-track history REPLACES the
        // entry of a stamp it already holds instead of pushing a second one.
        const bool counted = counted_any_ && !ts.stamps.empty() && counted_stamp_ == ts.stamps.front();
        if (!counted) stats_.features += (int)n;
        const bool from_sweep = p_.assign_depth && cloud && n_pts && n;
        if (from_sweep) {
            depth_.assign(n, -1.f);
            if (prefetch_open_ && prefetch_stamp_ == ts.stamps.front() && prefetch_n_ == n) {  // started by prefetchDepth
                prefetch_open_ = false;
                const int rc = limo_depth_estimate_end(ctx_, depth_.data(), n);
                if (rc != LIMO_OK) throw std::runtime_error(std::string("limo_depth_estimate_end: ") + limo_last_error(ctx_));
                ++stats_.depth_prefetched;
            } else {
                closeOpenDepthCall();
                stageFeatures(ts);
                const Pose T = convert(T_cam_lidar_);
                const int rc = limo_depth_estimate(ctx_, cloud, n_pts, T.data(), camera_->focal_length, camera_->principal_point[0],
                                                   camera_->principal_point[1], p_.image_width, p_.image_height, uv_.data(), n, ground_.data(),
                                                   &depth_params_, depth_.data());
                if (rc != LIMO_OK) throw std::runtime_error(std::string("limo_depth_estimate: ") + limo_last_error(ctx_));
            }
        }
        // remember this frame's depth per track, fill the history points from the memory (ring of the last frames)
        const auto t_hist = std::chrono::steady_clock::now();
        const TimestampNSec stamp = ts.stamps.front();
        out.clear();
        for (size_t i = 0; i < n; ++i) {
            const Tracklet& tr = ts.tracks[i];
            if (tr.feature_points.empty()) continue;
            const float d0 = (from_sweep && tr.feature_points[0].d < 0.f) ? depth_[i] : tr.feature_points[0].d;  // (a depth the caller knows stays)
            out.push_back(d0);
            History& h = history_[tr.id];
            if (h.n > 0 && h.stamp[(h.head - 1) & (History::kDepth - 1)] == stamp) {
                h.d[(h.head - 1) & (History::kDepth - 1)] = d0;
            } else {
                h.stamp[h.head] = stamp;
                h.d[h.head] = d0;
                h.head = (h.head + 1) & (History::kDepth - 1);
                h.n = std::min<int>(History::kDepth, h.n + 1);
            }
            for (size_t k = 1; k < tr.feature_points.size(); ++k) {
                float dk = tr.feature_points[k].d;
                if (dk < 0.f && k < ts.stamps.size()) {
                    // (a track seen in consecutive frames has the entry of stamps[k] k places behind the newest: looked at first; stamps
                    // are unique inside a history, so the scan finds the same entry or none)
                    const int e = (h.head - 1 - (int)k) & (History::kDepth - 1);
                    if ((int)k < h.n && h.stamp[e] == ts.stamps[k]) {
                        dk = h.d[e];
                    } else {
                        for (int j = 0; j < h.n; ++j) {
                            const int e2 = (h.head - 1 - j) & (History::kDepth - 1);
                            if (h.stamp[e2] == ts.stamps[k]) dk = h.d[e2];
                        }
                    }
                }
                out.push_back(dk);
            }
            if (!counted) stats_.features_with_depth += d0 > 0.f;
        }
        counted_any_ = true;
        counted_stamp_ = stamp;
        stats_.sec_depth_history += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_hist).count();
        if (++frames_since_gc_ >= 64) {  // forget tracks that ended
            frames_since_gc_ = 0;
            const TimestampNSec oldest = ts.stamps.back();
            for (auto it = history_.begin(); it != history_.end();)
                it = it->second.stamp[(it->second.head - 1) & (History::kDepth - 1)] < oldest ? history_.erase(it) : std::next(it);
        }
    }
    static void applyDepths(Tracklets& ts, const std::vector<float>& d) {
        size_t j = 0;
        for (auto& tr : ts.tracks)
            for (auto& fp : tr.feature_points) fp.d = d[j++];
    }

    StreamParams p_;
    Camera::Ptr camera_;
    EigenPose T_cam_lidar_;
    BundleAdjusterKeyframes ba_;
    KeyframeSelector selector_;
    limo_ctx* ctx_ = nullptr;
    limo_depth_params depth_params_;
    struct History {  // the depths this driver assigned to a track in the last frames it was seen in (ring, newest at head - 1)
        static constexpr int kDepth = 16;
        TimestampNSec stamp[kDepth];
        float d[kDepth];
        int head = 0, n = 0;
    };
    std::unordered_map<unsigned long, History> history_;
    int frames_since_gc_ = 0;
    std::vector<float> uv_, depth_, point_d_;
    std::vector<uint8_t> ground_;
    bool prefetch_open_ = false;  // prefetchDepth: a limo_depth_estimate_begin whose frame process() has not seen yet
    TimestampNSec prefetch_stamp_ = 0;
    size_t prefetch_n_ = 0;
    std::thread worker_;  // the depth thread and its one job slot
    std::mutex job_mutex_;
    std::condition_variable job_cv_;
    JobState job_state_ = JobState::Idle;
    const Tracklets* job_ts_ = nullptr;
    bool job_has_stamp_ = false;  // identity of the job's frame, read from the message when the job starts
    TimestampNSec job_stamp_ = 0;
    size_t job_tracks_ = 0;
    TimestampNSec counted_stamp_ = 0;  // the last frame whose features went into stats_ (computeDepths is idempotent per stamp)
    bool counted_any_ = false;
    const float* job_cloud_ = nullptr;
    size_t job_n_pts_ = 0;
    std::exception_ptr job_error_;
    bool ahead_valid_ = false;  // point_d_ holds the depths the depth thread computed for the message (ahead_stamp_, ahead_tracks_)
    TimestampNSec ahead_stamp_ = 0;
    size_t ahead_tracks_ = 0;
    const Tracklets* next_tracklets_ = nullptr;  // announceNextFrame
    const float* next_cloud_ = nullptr;
    size_t next_n_pts_ = 0;
    std::vector<EigenPose> poses_;
    EigenPose last_pose_ = EigenPose::Identity(), last_motion_ = EigenPose::Identity();
    bool have_last_ = false, have_motion_ = false;
    TimestampNSec last_stamp_ = 0;
    double last_solved_sec_ = -1e30;
    std::string last_summary_;
    Stats stats_;
    five_point::Motion last_five_point_;  // the last five-point estimate (diagnostics: inliers, samples)
};

}  // namespace keyframe_bundle_adjustment
