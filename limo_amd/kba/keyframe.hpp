// keyframe.hpp — per-keyframe measurement store; mirrors
// keyframe_bundle_adjustment/include/keyframe_bundle_adjustment/keyframe.hpp:27-196 and src/keyframe.cpp.
#pragma once
#include <algorithm>
#include <atomic>
#include <iterator>

#include "definitions.hpp"

namespace keyframe_bundle_adjustment {

class Keyframe {
public:
    enum class FixationStatus { Pose, Scale, None };
    using Ptr = std::shared_ptr<Keyframe>;
    using ConstPtr = std::shared_ptr<const Keyframe>;

    Keyframe() {}

    // Rig of several cameras: `landmark_to_cameras` names, per track id, the cameras that measured it.
    // (reference interface: keyframe.hpp:44-50 / src/keyframe.cpp:5-17)
    Keyframe(TimestampNSec timestamp, const Tracklets& tracklets, std::map<CameraId, Camera::Ptr> cameras,
             std::map<LandmarkId, CameraIds> landmark_to_cameras, EigenPose p,
             FixationStatus fix_stat = FixationStatus::None, Plane ground_plane = Plane()) {
        setup(timestamp, std::move(cameras), p, fix_stat, ground_plane);
        assignMeasurements(tracklets, landmark_to_cameras);
    }
    // One camera; it gets id 0.  (reference interface: keyframe.hpp:52-57 / src/keyframe.cpp:19-31)
    Keyframe(TimestampNSec timestamp, const Tracklets& tracklets, Camera::Ptr camera, EigenPose p,
             FixationStatus fix_stat = FixationStatus::None, Plane ground_plane = Plane()) {
        setup(timestamp, {{CameraId(0), std::move(camera)}}, p, fix_stat, ground_plane);
        assignMeasurements(tracklets, CameraId(0));
    }

    bool operator<(const Keyframe& kf) const { return timestamp_ < kf.timestamp_; }

    // Every track carries one feature point per stamp of `tracklets.stamps` (index 0 = newest, src/keyframe.cpp:61-75);
    // the column of THIS keyframe's stamp is the measurement it contributes.  A track shorter than that column was not
    // alive yet.  (A stamp that is not listed gives column == stamps.size(), like std::distance to end() in the reference.)
    void assignMeasurements(const Tracklets& tracklets, const CameraId& cam_id) {
        measurementsChanged();
        const size_t col = stampColumn(tracklets.stamps);
        for (const Tracklet& t : tracklets.tracks) {
            if (col >= t.feature_points.size()) continue;
            // (a tracker hands its tracks over in id order more often than not: an id above everything stored goes to the end of the
            // map without a search; any other id takes the search operator[] would have made)
            auto it = (measurements_.empty() || std::prev(measurements_.end())->first < t.id) ? measurements_.emplace_hint(measurements_.end(), t.id, std::map<CameraId, Measurement>())
                                                                                              : measurements_.try_emplace(t.id).first;
            it->second[cam_id] = t.feature_points[col];
        }
    }
    // The same for a rig (src/keyframe.cpp:43-59): no per-camera copies of the tracklets - one pass, the column is found
    // once.  An id that `landmark_lookup` does not know throws std::out_of_range, as the reference's .at() does.
    void assignMeasurements(const Tracklets& tracklets, const std::map<LandmarkId, CameraIds>& landmark_lookup) {
        measurementsChanged();
        const size_t col = stampColumn(tracklets.stamps);
        for (const Tracklet& t : tracklets.tracks) {
            const CameraIds& seen_by = landmark_lookup.at(t.id);
            if (col >= t.feature_points.size()) continue;
            auto& per_cam = measurements_[t.id];
            for (const CameraId& c : seen_by) per_cam[c] = t.feature_points[col];
        }
    }
    void assignPose(const EigenPose& p) { pose_ = convert(p); }

    Measurement& getMeasurement(LandmarkId lm_id, CameraId cam_id) { return measurements_.at(lm_id).at(cam_id); }
    const Measurement& getMeasurement(LandmarkId lm_id, CameraId cam_id) const {
        return measurements_.at(lm_id).at(cam_id);
    }
    // measurements of a landmark by the cameras of this keyframe's rig (keyframe.hpp:84-100)
    std::map<CameraId, Measurement> getMeasurements(LandmarkId lm_id) const {
        std::map<CameraId, Measurement> out;
        const auto row = measurements_.find(lm_id);
        if (row == measurements_.cend()) return out;
        for (const auto& cm : row->second)
            if (cameras_.count(cm.first)) out.emplace_hint(out.end(), cm.first, cm.second);
        return out;
    }
    bool hasMeasurement(const LandmarkId& lm_id, const CameraId& cam_id) const {
        const auto row = measurements_.find(lm_id);
        return row != measurements_.cend() && row->second.count(cam_id) != 0;
    }
    bool hasMeasurement(LandmarkId lm_id) const {
        const auto row = measurements_.find(lm_id);
        if (row == measurements_.cend()) return false;
        for (const auto& cm : row->second)
            if (cameras_.count(cm.first)) return true;
        return false;
    }
    // Landmark position in the frame of every camera that measured it in this keyframe (empty when none did):
    // camera <- vehicle <- origin (src/keyframe.cpp:81-104).
    std::map<CameraId, Vector3d> getProjectedLandmarkPosition(
        const std::pair<LandmarkId, Landmark::ConstPtr>& id_lm) const {
        std::map<CameraId, Vector3d> in_camera;
        const auto row = measurements_.find(id_lm.first);
        if (row != measurements_.cend()) {
            const Vector3d in_vehicle = getEigenPose() * Vector3d(id_lm.second->pos.data());
            for (const auto& cm : row->second)
                in_camera.emplace_hint(in_camera.end(), cm.first, cameras_.at(cm.first)->getEigenPose() * in_vehicle);
        }
        return in_camera;
    }
    EigenPose getEigenPose() const { return convert(pose_); }
    std::shared_ptr<Pose> getPosePtr() const { return std::make_shared<Pose>(pose_); }

    // measurements_ as ONE array in the order the map iterates it (landmark id, then camera id), each row with a pointer to the
    // Measurement inside the map: what the selector's schemes, the window cut and the flattening merge against sorted landmark lists
    // instead of walking the map node by node (not in the reference; its std::map stays the interface and the owner of the values - a
    // value changed in place is seen through the pointer).
    //
    // measurements_ is a PUBLIC member that a caller may edit between two calls into this library (the reference's interface), and no
    // cheap test sees every such edit.  So the table is only trusted for as long as nobody outside can have touched the map:
    //   * inside ONE call of a public entry point of this library (they open a MeasurementTableScope: the map cannot change while the
    //     call runs - the classes are single-threaded like the reference's) the table is built on first use and reused;
    //   * across calls only for a keyframe whose owner has promised not to edit measurements_ any more: freezeMeasurements() (the
    //     streaming driver does for the keyframes it creates; assignMeasurements() takes the promise back);
    //   * anywhere else it is rebuilt on every use - always right, one walk over the map.
    // A copy of a keyframe builds its own table (the rows point into the source's map); a moved map keeps its nodes.
    struct MeasurementRef {
        LandmarkId id;
        CameraId cam;
        const Measurement* m;
    };
    // Epochs are drawn from ONE process-wide counter: two threads that take turns on the same objects (a multi-threaded spinner
    // behind a lock) can never see equal epochs, so a table built in one thread's call is not trusted in another thread's call.
    struct MeasurementTableScope {  // opened by the public entry points; nests (the outermost call defines the epoch)
        MeasurementTableScope() {
            if (depth()++ == 0) epoch() = nextEpoch().fetch_add(1, std::memory_order_relaxed) + 1;
        }
        ~MeasurementTableScope() { --depth(); }
        MeasurementTableScope(const MeasurementTableScope&) = delete;
        MeasurementTableScope& operator=(const MeasurementTableScope&) = delete;
        static unsigned long long& epoch() {  // the epoch of the call this thread is in
            static thread_local unsigned long long e = 0;
            return e;
        }
        static std::atomic<unsigned long long>& nextEpoch() {
            static std::atomic<unsigned long long> n{1};  // (1 is the "built once" mark of frozen tables: never handed out)
            return n;
        }
        static int& depth() {
            static thread_local int d = 0;
            return d;
        }
    };
    void freezeMeasurements() const { table_.frozen = true; }
    bool measurementsFrozen() const { return table_.frozen; }
    // (kept for callers of the earlier interface: the next use rebuilds the table - and the promise of freezeMeasurements() is off)
    void measurementsChanged() const {
        table_.valid_epoch = 0;
        table_.frozen = false;
    }
    const std::vector<MeasurementRef>& measurementTable() const {
        refreshTable();
        return table_.rows;
    }
    // the landmark ids of measurements_ in ascending order, each once (the keys of the map as one contiguous array)
    const std::vector<LandmarkId>& measuredIds() const {
        refreshTable();
        return table_.ids;
    }

private:
    void setup(TimestampNSec stamp, std::map<CameraId, Camera::Ptr> rig, const EigenPose& p, FixationStatus fix, const Plane& plane) {
        timestamp_ = stamp;
        cameras_ = std::move(rig);
        fixation_status_ = fix;
        local_ground_plane_ = plane;
        is_active_ = true;
        assignPose(p);
    }
    void refreshTable() const {
        const bool in_call = MeasurementTableScope::depth() > 0;
        const bool trusted = table_.valid_epoch != 0 && (table_.frozen || (in_call && table_.valid_epoch == MeasurementTableScope::epoch()));
        if (trusted) return;
        std::vector<MeasurementRef>& rows = table_.rows;
        rows.clear();
        table_.ids.clear();
        rows.reserve(measurements_.size());
        table_.ids.reserve(measurements_.size());
        for (const auto& lm : measurements_) {
            table_.ids.push_back(lm.first);
            for (const auto& cm : lm.second) rows.push_back({lm.first, cm.first, &cm.second});
        }
        // (outside a call the table is never trusted again; a frozen keyframe only needs "built once")
        table_.valid_epoch = in_call ? MeasurementTableScope::epoch() : (table_.frozen ? 1 : 0);
    }
    size_t stampColumn(const std::vector<TimestampNSec>& stamps) const {
        size_t col = 0;
        while (col < stamps.size() && stamps[col] != timestamp_) ++col;
        return col;
    }

    struct TableCache {  // (copying a keyframe does not copy the table: its pointers lead into the source's map; the promise does carry over)
        std::vector<MeasurementRef> rows;
        std::vector<LandmarkId> ids;
        unsigned long long valid_epoch = 0;  // epoch of the call the table was built in (frozen: any non-zero value); 0 = not built
        bool frozen = false;
        TableCache() = default;
        TableCache(const TableCache& o) : frozen(o.frozen) {}
        TableCache& operator=(const TableCache& o) {
            rows.clear();
            ids.clear();
            valid_epoch = 0;
            frozen = o.frozen;
            return *this;
        }
        TableCache(TableCache&&) = default;  // (a moved map keeps its nodes where they are)
        TableCache& operator=(TableCache&&) = default;
    };
    mutable TableCache table_;

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
