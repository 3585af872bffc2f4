// keyframe.hpp — per-keyframe measurement store; mirrors
// keyframe_bundle_adjustment/include/keyframe_bundle_adjustment/keyframe.hpp:27-196 and src/keyframe.cpp.
#pragma once
#include <algorithm>

#include "definitions.hpp"

namespace keyframe_bundle_adjustment {

class Keyframe {
public:
    enum class FixationStatus { Pose, Scale, None };
    using Ptr = std::shared_ptr<Keyframe>;
    using ConstPtr = std::shared_ptr<const Keyframe>;

    Keyframe() {}

    // multi-camera constructor (keyframe.cpp:5-17)
    Keyframe(TimestampNSec timestamp, const Tracklets& tracklets, std::map<CameraId, Camera::Ptr> cameras,
             std::map<LandmarkId, CameraIds> landmark_to_cameras, EigenPose p,
             FixationStatus fix_stat = FixationStatus::None, Plane ground_plane = Plane())
            : timestamp_(timestamp), cameras_(cameras), fixation_status_(fix_stat), local_ground_plane_(ground_plane),
              is_active_(true) {
        assignMeasurements(tracklets, landmark_to_cameras);
        assignPose(p);
    }
    // mono constructor (keyframe.cpp:19-31)
    Keyframe(TimestampNSec timestamp, const Tracklets& tracklets, Camera::Ptr camera, EigenPose p,
             FixationStatus fix_stat = FixationStatus::None, Plane ground_plane = Plane())
            : timestamp_(timestamp), fixation_status_(fix_stat), local_ground_plane_(ground_plane), is_active_(true) {
        CameraId cam_id = 0;
        cameras_[cam_id] = camera;
        assignMeasurements(tracklets, cam_id);
        assignPose(p);
    }

    bool operator<(const Keyframe& kf) const { return this->timestamp_ < kf.timestamp_; }

    // keyframe.cpp:61-75: index of this keyframe's stamp in tracklets.stamps selects the feature point of every track
    void assignMeasurements(const Tracklets& tracklets, const CameraId& cam_id) {
        auto iter = std::find(tracklets.stamps.begin(), tracklets.stamps.end(), this->timestamp_);
        int index = static_cast<int>(std::distance(tracklets.stamps.begin(), iter));
        for (const auto& track : tracklets.tracks) {
            if (index < int(track.feature_points.size())) {
                measurements_[track.id][cam_id] = track.feature_points[index];
            }
        }
    }
    // keyframe.cpp:43-59
    void assignMeasurements(const Tracklets& tracklets, const std::map<LandmarkId, CameraIds>& landmark_lookup) {
        std::map<CameraId, Tracklets> out;
        for (const auto& track : tracklets.tracks) {
            for (const auto& cam_id : landmark_lookup.at(track.id)) {
                out[cam_id].stamps = tracklets.stamps;
                out[cam_id].tracks.push_back(track);
            }
        }
        for (const auto& el : out) assignMeasurements(el.second, el.first);
    }
    void assignPose(const EigenPose& p) { pose_ = convert(p); }

    Measurement& getMeasurement(LandmarkId lm_id, CameraId cam_id) { return measurements_.at(lm_id).at(cam_id); }
    const Measurement& getMeasurement(LandmarkId lm_id, CameraId cam_id) const {
        return measurements_.at(lm_id).at(cam_id);
    }
    std::map<CameraId, Measurement> getMeasurements(LandmarkId lm_id) const {
        std::map<CameraId, Measurement> out;
        for (const auto& cam : cameras_)
            if (hasMeasurement(lm_id, cam.first)) out[cam.first] = getMeasurement(lm_id, cam.first);
        return out;
    }
    bool hasMeasurement(const LandmarkId& lm_id, const CameraId& cam_id) const {
        auto it_lm = measurements_.find(lm_id);
        return it_lm != measurements_.cend() && it_lm->second.find(cam_id) != it_lm->second.cend();
    }
    bool hasMeasurement(LandmarkId lm_id) const {
        for (const auto& cam : cameras_)
            if (hasMeasurement(lm_id, cam.first)) return true;
        return false;
    }
    // keyframe.cpp:81-104
    std::map<CameraId, Vector3d> getProjectedLandmarkPosition(
        const std::pair<LandmarkId, Landmark::ConstPtr>& id_lm) const {
        auto it = measurements_.find(id_lm.first);
        if (it == measurements_.cend()) return std::map<CameraId, Vector3d>();
        const Vector3d p_vehicle = this->getEigenPose() * Vector3d(id_lm.second->pos.data());
        std::map<CameraId, Vector3d> out;
        for (const auto& cam_meas : it->second) out[cam_meas.first] = cameras_.at(cam_meas.first)->getEigenPose() * p_vehicle;
        return out;
    }
    EigenPose getEigenPose() const { return convert(pose_); }
    std::shared_ptr<Pose> getPosePtr() const { return std::make_shared<Pose>(pose_); }

public:
    TimestampNSec timestamp_{0};
    std::map<CameraId, Camera::Ptr> cameras_;
    FixationStatus fixation_status_{FixationStatus::None};
    Pose pose_{{1, 0, 0, 0, 0, 0, 0}};  // keyframe <- origin
    Plane local_ground_plane_;
    std::map<LandmarkId, std::map<CameraId, Measurement>> measurements_;
    bool is_active_{true};
};

}  // namespace keyframe_bundle_adjustment
