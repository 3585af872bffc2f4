// bundle_adjuster_keyframes.cpp — host shim: the reference's window bookkeeping re-implemented on plain containers,
// with the numerical work delegated to the C-ABI (include/limo_hip.h).
// Behaviour followed: keyframe_bundle_adjustment/src/bundle_adjuster_keyframes.cpp (push :289-329, landmark creation
// :332-382, updateLabels :388-431, solve :629-767, adjustPoseOnly :820-888, deactivateKeyframes :907-987,
// getKeyframe :989-1021) and src/definitions.cpp (conversions :14-28,63-69, calcQuaternionDiff :104-111).
#include "bundle_adjuster_keyframes.hpp"

#include <algorithm>
#include <iterator>
#include <chrono>

#include <cstdio>
#include <cstdlib>

#include <cstdlib>
#include <sstream>
#include <stdexcept>

#include "../../include/limo_hip.h"
#include "../csrc/kba_math.hpp"

namespace keyframe_bundle_adjustment {

namespace {
// Successive lookups of (mostly) increasing keys in a std::map: continues from the last position with a short walk instead
// of a descent from the root per key.  landmarks_ holds every landmark of the drive (hundreds of thousands after a few
// minutes) while the ids asked for - the active ones - sit next to each other at its end.
template <class T>
const T& key_of(const T& v) {
    return v;
}
template <class K, class V>
const K& key_of(const std::pair<const K, V>& v) {
    return v.first;
}
template <class Map>
class SortedFinder {
public:
    explicit SortedFinder(const Map& m) : m_(m), it_(m.cend()) {}
    typename Map::const_iterator find(const typename Map::key_type& k) {
        if (!started_ || it_ == m_.cend() || k < key_of(*it_)) {
            it_ = m_.lower_bound(k);
            started_ = true;
        } else {
            int steps = 0;
            while (it_ != m_.cend() && key_of(*it_) < k) {
                ++it_;
                if (++steps > 32) {
                    it_ = m_.lower_bound(k);
                    break;
                }
            }
        }
        return (it_ != m_.cend() && key_of(*it_) == k) ? it_ : m_.cend();
    }

private:
    const Map& m_;
    typename Map::const_iterator it_;
    bool started_ = false;
};
}  // namespace


// Debugging aid: LIMO_KBA_DUMP=<dir>[:<first>[:<last>]] writes the flattened window of the solve() calls number first..last
// (default: all) as <dir>/solve_NNNNNN.bin = int32 n_kf, n_cam, n_lm, n_obs, then the arrays of limo_ba_window in declaration
// order (tests/window_io.py reads them back into a Window): how a window of a long drive becomes a test fixture.
static void dump_window_if_asked(const limo_ba_window& w) {
    static const char* spec = std::getenv("LIMO_KBA_DUMP");
    static int call = -1;
    ++call;
    if (!spec) return;
    std::string dir(spec);
    long first = 0, last = 1L << 40;
    const size_t c1 = dir.find(':');
    if (c1 != std::string::npos) {
        const std::string rest = dir.substr(c1 + 1);
        dir = dir.substr(0, c1);
        first = std::atol(rest.c_str());
        const size_t c2 = rest.find(':');
        last = c2 != std::string::npos ? std::atol(rest.c_str() + c2 + 1) : first;
    }
    if (call < first || call > last) return;
    char name[64];
    std::snprintf(name, sizeof(name), "/solve_%06d.bin", call);
    std::FILE* f = std::fopen((dir + name).c_str(), "wb");
    if (!f) return;
    const int32_t head[4] = {w.n_kf, w.n_cam, w.n_lm, w.n_obs};
    std::fwrite(head, sizeof(int32_t), 4, f);
    std::fwrite(w.kf_pose, sizeof(double), 7 * (size_t)w.n_kf, f);
    std::fwrite(w.kf_plane_dir, sizeof(double), 3 * (size_t)w.n_kf, f);
    std::fwrite(w.kf_plane_dist, sizeof(double), (size_t)w.n_kf, f);
    std::fwrite(w.kf_fixation, sizeof(int32_t), (size_t)w.n_kf, f);
    std::fwrite(w.cam, sizeof(double), 10 * (size_t)w.n_cam, f);
    std::fwrite(w.lm_pos, sizeof(double), 3 * (size_t)w.n_lm, f);
    std::fwrite(w.lm_weight, sizeof(double), (size_t)w.n_lm, f);
    std::fwrite(w.lm_is_ground, 1, (size_t)w.n_lm, f);
    std::fwrite(w.obs_kf, sizeof(int32_t), (size_t)w.n_obs, f);
    std::fwrite(w.obs_lm, sizeof(int32_t), (size_t)w.n_obs, f);
    std::fwrite(w.obs_cam, sizeof(int32_t), (size_t)w.n_obs, f);
    std::fwrite(w.obs_u, sizeof(float), (size_t)w.n_obs, f);
    std::fwrite(w.obs_v, sizeof(float), (size_t)w.n_obs, f);
    std::fwrite(w.obs_d, sizeof(float), (size_t)w.n_obs, f);
    std::fclose(f);
}

// ------------------------------------------------------------------------------------------ conversions
Pose convert(const EigenPose& p) {
    // rotation matrix -> unit quaternion (what Eigen::Quaterniond(Matrix3d) computes)
    const double* R = p.R;
    double w, x, y, z;
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0.0) {
        double s = std::sqrt(tr + 1.0);
        w = 0.5 * s;
        s = 0.5 / s;
        x = (R[7] - R[5]) * s;
        y = (R[2] - R[6]) * s;
        z = (R[3] - R[1]) * s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        double q[3];
        q[i] = 0.5 * s;
        s = 0.5 / s;
        w = (R[k * 3 + j] - R[j * 3 + k]) * s;
        q[j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
        q[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
        x = q[0];
        y = q[1];
        z = q[2];
    }
    return Pose{{w, x, y, z, p.t[0], p.t[1], p.t[2]}};
}

EigenPose convert(const Pose& pose) {
    EigenPose p;
    kba::quat_R(pose.data(), p.R);
    p.t[0] = pose[4];
    p.t[1] = pose[5];
    p.t[2] = pose[6];
    return p;
}

TimestampSec convert(const TimestampNSec& ts) {
    return static_cast<TimestampSec>(ts * 1e-09);
}
TimestampNSec convert(const TimestampSec& ts) {
    return static_cast<TimestampNSec>(ts * 1e09);
}

double calcQuaternionDiff(const Pose& p0, const Pose& p1) {
    // q10 = q1^-1 * q0 ; angle of AngleAxis(q10)
    const double n1 = p1[0] * p1[0] + p1[1] * p1[1] + p1[2] * p1[2] + p1[3] * p1[3];
    const double a[4] = {p1[0] / n1, -p1[1] / n1, -p1[2] / n1, -p1[3] / n1};
    const double* b = p0.data();
    const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    const double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    const double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    const double n = std::sqrt(x * x + y * y + z * z);
    return n != 0.0 ? 2.0 * std::atan2(n, std::fabs(w)) : 0.0;
}

// ------------------------------------------------------------------------------------------ exceptions
BundleAdjusterKeyframes::NotEnoughKeyframesException::NotEnoughKeyframesException(size_t is, size_t should)
        : num_is(is), num_should_be(should) {
    std::stringstream ss;
    ss << "Not enough keyframes available in bundle_adjuster_keyframes. Should be " << num_should_be << " is " << num_is;
    msg = ss.str();
}
BundleAdjusterKeyframes::KeyframeNotFoundException::KeyframeNotFoundException(TimestampNSec timestamp) : ts_(timestamp) {
    std::stringstream ss;
    ss << "keyframe corresponding to timestamp " << ts_ << " nano seconds not found";
    msg = ss.str();
}

// ------------------------------------------------------------------------------------------ lifecycle
BundleAdjusterKeyframes::BundleAdjusterKeyframes() : solver_time_sec(0.2) {
    landmark_selector_ = std::make_unique<LandmarkSelector>();
    landmark_selector_->addScheme(LandmarkRejectionSchemeCheirality::create());  // always first: fast and reliable (:118)
}
BundleAdjusterKeyframes::~BundleAdjusterKeyframes() {
    if (ctx_) limo_ctx_destroy(ctx_);
}

limo_ctx* BundleAdjusterKeyframes::context() {
    if (!ctx_) {
        int dev = 0;
        if (const char* e = std::getenv("LIMO_DEVICE")) dev = std::atoi(e);
        const int rc = limo_ctx_create(dev, &ctx_);
        if (rc != LIMO_OK)
            throw std::runtime_error("limo_ctx_create failed (rc=" + std::to_string(rc) +
                                     "): the keyframe BA hot path needs the HIP library and an MI355X; there is no CPU fallback");
    }
    return ctx_;
}

void BundleAdjusterKeyframes::set_solver_time(double s) {
    this->solver_time_sec = s;
}

// ------------------------------------------------------------------------------------------ push / landmark creation
namespace {
limo_ray make_ray(const Camera& cam, const Keyframe& kf, const Measurement& m) {
    limo_ray r;
    const Pose pc = convert(cam.getEigenPose() * kf.getEigenPose());  // camera <- origin
    for (int i = 0; i < 7; ++i) r.pose_cam_origin[i] = pc[i];
    r.f = cam.focal_length;
    r.cx = cam.principal_point[0];
    r.cy = cam.principal_point[1];
    r.u = m.u;
    r.v = m.v;
    r.d = m.d;
    r.pad = 0;
    return r;
}
}  // namespace

void BundleAdjusterKeyframes::push(const std::vector<Keyframe>& kfs) {
    for (const auto& kf : kfs) push(kf);
}

void BundleAdjusterKeyframes::push(const Keyframe& kf) { push(Keyframe(kf)); }

void BundleAdjusterKeyframes::push(Keyframe&& kf_in) {
    Keyframe::MeasurementTableScope table_scope;  // (measurement tables are trusted inside one public call: keyframe.hpp)
    const TimestampNSec stamp = kf_in.timestamp_;
    const auto stored = std::make_shared<Keyframe>(std::move(kf_in));
    keyframes_[stamp] = stored;
    const Keyframe& kf = *stored;
    active_keyframe_ids_.insert(stamp);
    // Every landmark this keyframe introduces is initialised in ONE device call (the reference does it one by one,
    // :289-330): depth back-projection where the keyframe measures a depth, N-view triangulation otherwise.
    std::vector<LandmarkId> ids;
    std::vector<int32_t> off{0};
    std::vector<limo_ray> rays;
    std::vector<uint8_t> use_depth;
    // The rays of one (keyframe, camera) share pose and intrinsics: that part is formed once per view here, not once per ray (the
    // same statements on the same operands as make_ray - collectRays(..) below, which calculateLandmark still uses - so the same
    // bits); the new ids come in ascending order, so every active keyframe's measurements are walked by a finder of their own
    // instead of two tree searches per (landmark, keyframe).
    struct View {
        const Keyframe* kf;
        CameraId cam;
        limo_ray base;
    };
    auto views_of = [](const Keyframe& k, std::vector<View>& out) {
        for (const auto& id_cam : k.cameras_) {
            Measurement none;
            none.u = none.v = 0.f;
            none.d = -1.f;
            out.push_back({&k, id_cam.first, make_ray(*id_cam.second, k, none)});
        }
    };
    auto ray_of = [](const View& v, const Measurement& m) {
        limo_ray r = v.base;
        r.u = m.u;
        r.v = m.v;
        r.d = m.d;
        return r;
    };
    std::vector<View> own_views, active_views;
    views_of(kf, own_views);
    using MeasFinder = SortedFinder<decltype(kf.measurements_)>;
    std::vector<MeasFinder> finders;  // one per active keyframe, in the order of active_views' keyframes
    std::vector<std::pair<size_t, size_t>> views_of_kf;  // [first, last) of active_views
    std::vector<const Keyframe*> kf_of_finder;            // (a keyframe without cameras has no view to take it from)
    for (const auto& id : active_keyframe_ids_) {
        const Keyframe& k = *keyframes_.at(id);
        const size_t first = active_views.size();
        views_of(k, active_views);
        views_of_kf.push_back({first, active_views.size()});
        finders.emplace_back(k.measurements_);
        kf_of_finder.push_back(&k);
    }
    SortedFinder<decltype(landmarks_)> known(landmarks_);
    for (const auto& m : kf.measurements_) {
        if (known.find(m.first) != landmarks_.cend()) continue;
        bool has_depth = false;  // containsDepth(kf, id)
        for (const auto& cam_meas : m.second) has_depth = has_depth || cam_meas.second.d >= 0;
        const size_t before = rays.size();
        if (has_depth) {  // collectRays(kf, id): this keyframe's views of the landmark
            for (const auto& cam_meas : m.second)
                for (const View& v : own_views)
                    if (v.cam == cam_meas.first) rays.push_back(ray_of(v, cam_meas.second));
        } else {  // collectRays(id): every active keyframe / camera that sees the landmark (getMeasurementsAndPoses, :125-159)
            for (size_t k = 0; k < finders.size(); ++k) {
                const Keyframe& ak = *kf_of_finder[k];
                const auto it = finders[k].find(m.first);
                if (it == ak.measurements_.cend()) continue;
                for (size_t vi = views_of_kf[k].first; vi < views_of_kf[k].second; ++vi) {
                    const auto im = it->second.find(active_views[vi].cam);
                    if (im != it->second.cend()) rays.push_back(ray_of(active_views[vi], im->second));
                }
            }
            if (rays.size() - before < 2) {  // not enough views to triangulate (:363-365)
                rays.resize(before);
                continue;
            }
        }
        ids.push_back(m.first);
        use_depth.push_back(has_depth ? 1 : 0);
        off.push_back((int32_t)rays.size());
    }
    if (!ids.empty()) {
        std::vector<double> pos(3 * ids.size());
        std::vector<uint8_t> ok(ids.size(), 0);
        const int rc = limo_landmark_init(context(), (int32_t)ids.size(), off.data(), rays.data(), use_depth.data(), pos.data(), ok.data());
        if (rc != LIMO_OK) throw std::runtime_error(std::string("limo_landmark_init: ") + limo_last_error(ctx_));
        for (size_t i = 0; i < ids.size(); ++i)
            if (ok[i]) landmarks_[ids[i]] = std::make_shared<Landmark>(Vector3d(pos.data() + 3 * i), use_depth[i] != 0);
    }
    // (measurements and active ids are both in id order: every insertion gets the place of the one before it as its hint)
    SortedFinder<decltype(landmarks_)> now_known(landmarks_);
    auto hint = active_landmark_ids_.begin();
    for (const auto& m : kf.measurements_) {
        if (now_known.find(m.first) == landmarks_.cend()) continue;
        for (int steps = 0; hint != active_landmark_ids_.end() && *hint < m.first; ++hint)
            if (++steps > 8) {  // (far away: a search, not a walk)
                hint = active_landmark_ids_.lower_bound(m.first);
                break;
            }
        hint = active_landmark_ids_.insert(hint, m.first);
    }
}

void BundleAdjusterKeyframes::collectRays(const Keyframe& kf, const LandmarkId& lId, std::vector<limo_ray>& rays) const {
    for (const auto& m : kf.measurements_.at(lId)) rays.push_back(make_ray(*kf.cameras_.at(m.first), kf, m.second));
}

void BundleAdjusterKeyframes::collectRays(const LandmarkId& lId, std::vector<limo_ray>& rays) const {
    // getMeasurementsAndPoses (:125-159): every active keyframe / camera that sees the landmark
    for (const auto& id : active_keyframe_ids_) {
        const Keyframe& kf = *keyframes_.at(id);
        for (const auto& id_cam : kf.cameras_)
            if (kf.hasMeasurement(lId, id_cam.first)) rays.push_back(make_ray(*id_cam.second, kf, kf.getMeasurement(lId, id_cam.first)));
    }
}

bool BundleAdjusterKeyframes::calculateLandmark(const Keyframe& kf, const LandmarkId& lId, Vector3d& posAbs) {
    std::vector<limo_ray> rays;
    collectRays(kf, lId, rays);
    const int32_t off[2] = {0, (int32_t)rays.size()};
    const uint8_t use_depth = 1;
    uint8_t ok = 0;
    double pos[3];
    if (limo_landmark_init(context(), 1, off, rays.data(), &use_depth, pos, &ok) != LIMO_OK || !ok) return false;
    posAbs = Vector3d(pos);
    return true;
}

bool BundleAdjusterKeyframes::calculateLandmark(const LandmarkId& lId, Vector3d& posAbs) {
    std::vector<limo_ray> rays;
    collectRays(lId, rays);
    if (rays.size() < 2) return false;
    const int32_t off[2] = {0, (int32_t)rays.size()};
    const uint8_t use_depth = 0;
    uint8_t ok = 0;
    double pos[3];
    if (limo_landmark_init(context(), 1, off, rays.data(), &use_depth, pos, &ok) != LIMO_OK || !ok) return false;
    posAbs = Vector3d(pos);
    return true;
}

// ------------------------------------------------------------------------------------------ labels
void BundleAdjusterKeyframes::updateLabels(const Tracklets& t, double shrubbery_weight) {
    Keyframe::MeasurementTableScope table_scope;  // (measurement tables are trusted inside one public call: keyframe.hpp)
    // (the three label sets are looked up once, not per track: a std::string key per lookup was a third of this function)
    // (... and turned into tables over the label value once per call: three std::set searches per track were half of what was left)
    struct LabelSet {
        const std::set<int>& s;
        std::vector<char> table;  // labels 0 .. 255; anything else is looked up in the set
        explicit LabelSet(const std::set<int>& set) : s(set), table(256, 0) {
            for (int l : s)
                if (l >= 0 && l < 256) table[l] = 1;
        }
        bool count(int l) const { return (l >= 0 && l < 256) ? table[l] != 0 : s.count(l) > 0; }
    };
    const LabelSet outlier_labels(labels_["outliers"]), shrubbery_labels(labels_["shrubbery"]), ground_labels(labels_["ground"]);
    std::set<LandmarkId> outlier_ids;
    for (const auto& id : landmark_selector_->getOutliers())
        if (active_landmark_ids_.count(id)) outlier_ids.insert(id);
    for (const auto& track : t.tracks)
        if (track.is_outlier || outlier_labels.count(track.label)) outlier_ids.insert(track.id);
    landmark_selector_->clearOutliers();
    landmark_selector_->setOutlier(outlier_ids);
    SortedFinder<decltype(active_landmark_ids_)> active(active_landmark_ids_);
    SortedFinder<decltype(landmarks_)> known(landmarks_);
    for (const auto& track : t.tracks) {
        if (active.find(track.id) == active_landmark_ids_.cend()) continue;
        // (the reference uses landmarks_.at(), which throws for an active id that could not be reconstructed)
        auto it = known.find(track.id);
        if (it == landmarks_.cend()) continue;
        if (shrubbery_labels.count(track.label)) it->second->weight = shrubbery_weight;
        it->second->is_ground_plane = ground_labels.count(track.label);
    }
}

// ------------------------------------------------------------------------------------------ getters
std::map<LandmarkId, Landmark::ConstPtr> BundleAdjusterKeyframes::filterLandmarksById(const std::set<LandmarkId>& ids) const {
    std::map<LandmarkId, Landmark::ConstPtr> out;
    SortedFinder<decltype(landmarks_)> known(landmarks_);
    for (const auto& id : ids) {
        auto it = known.find(id);
        if (it != landmarks_.cend()) out.insert(out.end(), {id, it->second});
    }
    return out;
}
std::map<LandmarkId, Landmark::ConstPtr> BundleAdjusterKeyframes::getActiveLandmarkConstPtrs() const {
    return filterLandmarksById(active_landmark_ids_);
}
std::map<LandmarkId, Landmark::ConstPtr> BundleAdjusterKeyframes::getSelectedLandmarkConstPtrs() const {
    return filterLandmarksById(selected_landmark_ids_);
}
std::map<KeyframeId, Keyframe::Ptr> BundleAdjusterKeyframes::getActiveKeyframePtrs() const {
    std::map<KeyframeId, Keyframe::Ptr> out;
    for (const auto& id : active_keyframe_ids_) out[id] = keyframes_.at(id);
    return out;
}
std::map<KeyframeId, Keyframe::ConstPtr> BundleAdjusterKeyframes::getActiveKeyframeConstPtrs() const {
    std::map<KeyframeId, Keyframe::ConstPtr> out;
    for (const auto& id : active_keyframe_ids_) out[id] = keyframes_.at(id);
    return out;
}
std::vector<std::pair<KeyframeId, Keyframe::Ptr>> BundleAdjusterKeyframes::getSortedIdsWithActiveKeyframePtrs() const {
    std::vector<std::pair<KeyframeId, Keyframe::Ptr>> v;
    for (const auto& id : active_keyframe_ids_) v.push_back({id, keyframes_.at(id)});
    std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return *(a.second) < *(b.second); });
    return v;
}
std::vector<Keyframe::Ptr> BundleAdjusterKeyframes::getSortedActiveKeyframePtrs() const {
    std::vector<Keyframe::Ptr> out;
    for (const auto& el : getSortedIdsWithActiveKeyframePtrs()) out.push_back(el.second);
    return out;
}

const Keyframe& BundleAdjusterKeyframes::getKeyframe(TimestampSec timestamp) const {
    if (keyframes_.size() == 0) throw NotEnoughKeyframesException(keyframes_.size(), 1);
    if (timestamp < 0.) {
        const Keyframe* best = nullptr;
        for (const auto& id : active_keyframe_ids_) {
            const Keyframe* k = keyframes_.at(id).get();
            if (!best || best->timestamp_ < k->timestamp_) best = k;
        }
        if (!best) throw NotEnoughKeyframesException(0, 1);
        return *best;
    }
    const TimestampNSec ts_nsec = convert(timestamp);
    for (const auto& id : active_keyframe_ids_)
        if (keyframes_.at(id)->timestamp_ == ts_nsec) return *keyframes_.at(id);
    for (const auto& el : keyframes_)
        if (el.second->timestamp_ == ts_nsec) return *el.second;
    throw KeyframeNotFoundException(ts_nsec);
}

// ------------------------------------------------------------------------------------------ window cut (:907-987)
// (the ids live with the keyframe's measurement table: trusted inside one public call, or for a frozen keyframe - keyframe.hpp)
const std::vector<LandmarkId>& BundleAdjusterKeyframes::measuredIds(const Keyframe& kf) { return kf.measuredIds(); }

void BundleAdjusterKeyframes::deactivateKeyframes(int min_num_connecting_landmarks, int min_size_optimization_window,
                                                  int max_size_optimization_window) {
    Keyframe::MeasurementTableScope table_scope;  // (measurement tables are trusted inside one public call: keyframe.hpp)
    auto sorted = getSortedIdsWithActiveKeyframePtrs();
    if (sorted.empty()) return;
    const Keyframe& newest = *sorted.back().second;
    int n = 0;  // 0 = newest
    for (auto it = sorted.rbegin(); it != sorted.rend(); ++it, ++n) {
        Keyframe& cur = *it->second;
        if (n > max_size_optimization_window - 1) {
            cur.is_active_ = false;
        } else if (n < min_size_optimization_window - 1) {
            cur.is_active_ = true;
        } else {
            int common = 0;  // landmark ids measured in both keyframes (std::set_intersection of the map keys, :88-111):
                             // one merge pass over the two sorted id arrays
            const std::vector<LandmarkId>&a = measuredIds(cur), &b = measuredIds(newest);
            size_t ia = 0, ib = 0;
            while (ia < a.size() && ib < b.size()) {
                if (a[ia] < b[ib])
                    ++ia;
                else if (b[ib] < a[ia])
                    ++ib;
                else {
                    ++common;
                    ++ia;
                    ++ib;
                }
            }
            cur.is_active_ = common > min_num_connecting_landmarks;
        }
        if (!cur.is_active_) {
            active_keyframe_ids_.erase(it->first);
        }
    }
    // active landmarks that some active keyframe still measures: the union of the active keyframes' id arrays (merged pairwise), then ONE
    // pass over the set of active ids against it; the ids nobody measures any more leave the set in place (a few hundred of ~3000 when a
    // keyframe drops out of the window).  (A merge of every keyframe's measurement MAP with the set, node by node, was 0.2 ms per solve.)
    std::vector<LandmarkId> measured, tmp;
    for (const auto& kf_id : active_keyframe_ids_) {
        const std::vector<LandmarkId>& ids = measuredIds(*keyframes_.at(kf_id));
        tmp.clear();
        tmp.reserve(measured.size() + ids.size());
        std::set_union(measured.begin(), measured.end(), ids.begin(), ids.end(), std::back_inserter(tmp));
        measured.swap(tmp);
    }
    {
        size_t j = 0;
        for (auto it = active_landmark_ids_.begin(); it != active_landmark_ids_.end();) {
            while (j < measured.size() && measured[j] < *it) ++j;
            if (j < measured.size() && measured[j] == *it)
                ++it;
            else
                it = active_landmark_ids_.erase(it);
        }
    }
    // oldest active keyframe fixes the gauge, second oldest carries the scale prior (:962-986)
    auto rest = getSortedIdsWithActiveKeyframePtrs();
    if (rest.size() > 0) rest[0].second->fixation_status_ = Keyframe::FixationStatus::Pose;
    if (rest.size() > 1) rest[1].second->fixation_status_ = Keyframe::FixationStatus::Scale;
}

// ------------------------------------------------------------------------------------------ flattening
namespace {

struct Flat {
    std::vector<double> kf_pose, kf_dir, kf_dist, cam, lm_pos, lm_w;
    std::vector<int32_t> kf_fix, obs_kf, obs_lm, obs_cam;
    std::vector<uint8_t> lm_gp;
    std::vector<float> u, v, d;
    std::vector<Keyframe*> kfs;
    std::vector<Landmark*> lms;
    std::map<const Camera*, int> cam_index;
    limo_ba_window w;

    void reserve_landmarks(size_t n) {
        lms.reserve(n);
        lm_pos.reserve(3 * n);
        lm_w.reserve(n);
        lm_gp.reserve(n);
    }
    int camera(const Camera& c) {
        auto it = cam_index.find(&c);
        if (it != cam_index.end()) return it->second;
        const int idx = (int)cam_index.size();
        cam_index[&c] = idx;
        cam.push_back(c.focal_length);
        cam.push_back(c.principal_point[0]);
        cam.push_back(c.principal_point[1]);
        for (int i = 0; i < 7; ++i) cam.push_back(c.pose_camera_vehicle[i]);
        return idx;
    }
    void add_keyframe(Keyframe& kf) {
        kfs.push_back(&kf);
        for (int i = 0; i < 7; ++i) kf_pose.push_back(kf.pose_[i]);
        for (int i = 0; i < 3; ++i) kf_dir.push_back(kf.local_ground_plane_.direction[i]);
        kf_dist.push_back(kf.local_ground_plane_.distance);
        kf_fix.push_back(kf.fixation_status_ == Keyframe::FixationStatus::Pose
                             ? LIMO_FIX_POSE
                             : kf.fixation_status_ == Keyframe::FixationStatus::Scale ? LIMO_FIX_SCALE : LIMO_FIX_NONE);
    }
    void add_observations(int k, const Keyframe& kf, const std::vector<std::pair<LandmarkId, int>>& lm_index) {
        const size_t room = obs_kf.size() + lm_index.size();
        obs_kf.reserve(room);
        obs_lm.reserve(room);
        obs_cam.reserve(room);
        u.reserve(room);
        v.reserve(room);
        d.reserve(room);
        // (the keyframe's measurement table and the index are both sorted by landmark id: one merge pass over two arrays; the map
        // node of a measurement is only touched when its landmark is in the window)
        Keyframe::MeasurementTableScope table_scope;  // (the table is trusted for the duration of this call: keyframe.hpp)
        const auto& rows = kf.measurementTable();
        auto it = lm_index.cbegin();
        bool have_cam = false;  // (camera id -> index of the flattened camera table, looked up when the id changes)
        CameraId last_cam{};
        int last_cam_index = -1;
        for (const auto& row : rows) {  // addKeyframeToProblem, :569-576
            while (it != lm_index.cend() && it->first < row.id) ++it;
            if (it == lm_index.cend()) break;
            if (it->first != row.id) continue;
            if (!have_cam || !(row.cam == last_cam)) {
                last_cam_index = camera(*kf.cameras_.at(row.cam));
                last_cam = row.cam;
                have_cam = true;
            }
            obs_kf.push_back(k);
            obs_lm.push_back(it->second);
            obs_cam.push_back(last_cam_index);
            u.push_back(row.m->u);
            v.push_back(row.m->v);
            d.push_back(row.m->d);
        }
    }
    void finish() {
        w.n_kf = (int)kfs.size();
        w.n_cam = (int)cam_index.size();
        w.n_lm = (int)lms.size();
        w.n_obs = (int)obs_kf.size();
        w.kf_pose = kf_pose.data();
        w.kf_plane_dir = kf_dir.data();
        w.kf_plane_dist = kf_dist.data();
        w.kf_fixation = kf_fix.data();
        w.cam = cam.data();
        w.lm_pos = lm_pos.data();
        w.lm_weight = lm_w.data();
        w.lm_is_ground = lm_gp.data();
        w.obs_kf = obs_kf.data();
        w.obs_lm = obs_lm.data();
        w.obs_cam = obs_cam.data();
        w.obs_u = u.data();
        w.obs_v = v.data();
        w.obs_d = d.data();
    }
    void write_back(bool landmarks) {
        for (size_t k = 0; k < kfs.size(); ++k) {
            for (int i = 0; i < 7; ++i) kfs[k]->pose_[i] = kf_pose[7 * k + i];
            for (int i = 0; i < 3; ++i) kfs[k]->local_ground_plane_.direction[i] = kf_dir[3 * k + i];
            kfs[k]->local_ground_plane_.distance = kf_dist[k];
        }
        if (landmarks)
            for (size_t l = 0; l < lms.size(); ++l)
                for (int i = 0; i < 3; ++i) lms[l]->pos[i] = lm_pos[3 * l + i];
    }
};

std::string report_string(const limo_ba_report& r, const char* what) {
    static const char* term[] = {"CONVERGENCE", "NO_CONVERGENCE", "FAILURE"};
    std::stringstream ss;
    ss << "Merged summaries (" << what << ", MI355X HIP backend):\n"
       << "  residual blocks: depth " << r.n_depth_blocks << ", reprojection " << r.n_repr_blocks << ", ground plane "
       << r.n_gp_blocks << "\n  solves " << r.num_solves << ", LM iterations " << r.iterations_total << " (final solve "
       << r.iterations_final << "), accepted steps " << r.successful_steps << ", trimmed landmarks "
       << r.n_trimmed_landmarks << "\n  initial cost " << r.initial_cost << " final cost " << r.final_cost
       << " termination " << (r.termination >= 0 && r.termination < 3 ? term[r.termination] : "?") << "\n"
       << "Duration solveTrimmed=" << r.time_sec << " sec\n";
    return ss.str();
}

}  // namespace

std::string BundleAdjusterKeyframes::solve() {
    Keyframe::MeasurementTableScope table_scope;  // (measurement tables are trusted inside one public call: keyframe.hpp)
    if (keyframes_.size() < 3) throw NotEnoughKeyframesException(keyframes_.size(), 3);
    using clk = std::chrono::steady_clock;
    static const bool shim_trace = std::getenv("LIMO_SHIM_TRACE") != nullptr;  // where the host time of solve() goes
    const auto t_s0 = clk::now();
    // (getActiveLandmarkConstPtrs() as a sorted vector: the selector's schemes read it in this form, no map is built for them)
    LandmarkView active_lms;
    {
        active_lms.reserve(active_landmark_ids_.size());
        SortedFinder<decltype(landmarks_)> known(landmarks_);
        for (const auto& id : active_landmark_ids_) {
            auto it = known.find(id);
            if (it != landmarks_.cend()) active_lms.push_back({id, it->second});
        }
    }
    const auto active_kfs = getActiveKeyframeConstPtrs();
    const auto t_s0b = clk::now();
    selected_landmark_ids_ = landmark_selector_->select(active_lms, active_kfs);
    const auto t_s1 = clk::now();

    Flat F;
    for (const auto& id : active_keyframe_ids_) F.add_keyframe(*keyframes_.at(id));
    std::vector<std::pair<LandmarkId, int>> lm_index;  // (selected ids come in id order: sorted as built)
    F.reserve_landmarks(selected_landmark_ids_.size());
    lm_index.reserve(selected_landmark_ids_.size());
    SortedFinder<decltype(landmarks_)> known(landmarks_);
    for (const auto& id : selected_landmark_ids_) {
        auto it = known.find(id);
        if (it == landmarks_.cend()) continue;
        lm_index.push_back({id, (int)F.lms.size()});
        F.lms.push_back(it->second.get());
        for (int i = 0; i < 3; ++i) F.lm_pos.push_back(it->second->pos[i]);
        F.lm_w.push_back(it->second->weight);
        F.lm_gp.push_back(it->second->is_ground_plane ? 1 : 0);
    }
    for (size_t k = 0; k < F.kfs.size(); ++k) F.add_observations((int)k, *F.kfs[k], lm_index);
    F.finish();

    limo_ba_options o;
    limo_ba_default_options(&o);
    o.depth_thres = outlier_rejection_options_.depth_thres;
    o.reprojection_thres = outlier_rejection_options_.reprojection_thres;
    o.depth_quantile = outlier_rejection_options_.depth_quantile;
    o.reprojection_quantile = outlier_rejection_options_.reprojection_quantile;
    o.num_trim_rounds = outlier_rejection_options_.num_iterations;
    o.min_landmarks_for_trimming = 100;  // :741 (compared with selected_landmark_ids_.size())
    o.max_solver_time_sec = solver_time_sec;
    limo_ba_report rep;
    limo_ctx* ctx = context();
    dump_window_if_asked(F.w);
    const auto t_s2 = clk::now();
    const int rc = limo_ba_solve(ctx, &F.w, &o, &rep);
    const auto t_s3 = clk::now();
    if (rc == LIMO_ERR_NOT_ENOUGH_KF) throw NotEnoughKeyframesException(F.kfs.size(), 3);
    if (rc != LIMO_OK) throw std::runtime_error(std::string("limo_ba_solve: ") + limo_last_error(ctx));
    F.write_back(true);
    last_report_ = {rep.termination, rep.num_solves, rep.iterations_total, rep.n_trimmed_landmarks, rep.n_depth_blocks,
                    rep.n_repr_blocks, rep.n_gp_blocks, rep.initial_cost, rep.final_cost, rep.time_sec};
    if (shim_trace) {
        auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        std::fprintf(stderr, "[shim] solve: %zu active landmarks, %zu selected, %d LM iterations in %d solves: active maps %.0f us, selection %.0f us, flatten %.0f us, limo_ba_solve %.0f us, write-back %.0f us\n",
                     active_landmark_ids_.size(), selected_landmark_ids_.size(), rep.iterations_total, rep.num_solves, us(t_s0, t_s0b), us(t_s0b, t_s1), us(t_s1, t_s2),
                     us(t_s2, t_s3), us(t_s3, clk::now()));
    }
    return report_string(rep, "solve");
}

std::string BundleAdjusterKeyframes::adjustPoseOnly(Keyframe& kf) {
    Keyframe::MeasurementTableScope table_scope;  // (measurement tables are trusted inside one public call: keyframe.hpp)
    using clk = std::chrono::steady_clock;
    static const bool shim_trace = std::getenv("LIMO_SHIM_TRACE") != nullptr;
    const auto t_p0 = clk::now();
    selected_landmark_ids_ = landmark_selector_->getLastSelection();  // :828
    const auto t_p1 = clk::now();
    Flat F;
    F.add_keyframe(kf);
    std::vector<std::pair<LandmarkId, int>> lm_index;  // (selected ids come in id order: sorted as built)
    F.reserve_landmarks(selected_landmark_ids_.size());
    lm_index.reserve(selected_landmark_ids_.size());
    SortedFinder<decltype(landmarks_)> known(landmarks_);
    for (const auto& id : selected_landmark_ids_) {
        auto it = known.find(id);
        if (it == landmarks_.cend()) continue;
        lm_index.push_back({id, (int)F.lms.size()});
        F.lms.push_back(it->second.get());
        for (int i = 0; i < 3; ++i) F.lm_pos.push_back(it->second->pos[i]);
        F.lm_w.push_back(it->second->weight);
        F.lm_gp.push_back(0);
    }
    F.add_observations(0, kf, lm_index);
    F.finish();

    limo_speed_prior prior;
    prior.speed_weight = 0.0;
    if (active_keyframe_ids_.size() > 2) {  // :835-853
        auto sorted = getSortedActiveKeyframePtrs();
        const Keyframe& k0 = *sorted[sorted.size() - 1];
        const Keyframe& k1 = *sorted[sorted.size() - 2];
        const double rot_diff = calcQuaternionDiff(k0.pose_, k1.pose_);
        if (rot_diff < 0.03) {
            const double cur_ts = convert(kf.timestamp_), before = convert(k0.timestamp_), before2 = convert(k1.timestamp_);
            const double dt_cur = cur_ts - before, dt_before = before - before2;
            if (dt_cur <= 0. || dt_before <= 0.) throw std::runtime_error("In PoseRegularizationSpeed: invalid timestamps");
            prior.speed_weight = 1. * (1 - rot_diff / 0.03);
            prior.dt_cur = dt_cur;
            const Vector3d v = (k0.getEigenPose() * k1.getEigenPose().inverse()).translation() * (1.0 / dt_before);
            for (int i = 0; i < 3; ++i) prior.vel_prev[i] = v[i];
            for (int i = 0; i < 7; ++i) prior.pose_before[i] = k0.pose_[i];
        }
    }
    limo_ba_options o;
    limo_ba_default_options(&o);
    o.depth_thres = outlier_rejection_options_.depth_thres;
    o.reprojection_thres = outlier_rejection_options_.reprojection_thres;
    o.depth_quantile = outlier_rejection_options_.depth_quantile;
    o.reprojection_quantile = outlier_rejection_options_.reprojection_quantile;
    o.num_trim_rounds = outlier_rejection_options_.num_iterations;
    o.min_landmarks_for_trimming = 30;  // :865
    o.max_solver_time_sec = solver_time_sec;
    limo_ba_report rep;
    limo_ctx* ctx = context();
    const auto t_p2 = clk::now();
    const int rc = limo_ba_adjust_pose_only(ctx, &F.w, prior.speed_weight > 0.0 ? &prior : nullptr, &o, &rep);
    const auto t_p3 = clk::now();
    if (rc != LIMO_OK) throw std::runtime_error(std::string("limo_ba_adjust_pose_only: ") + limo_last_error(ctx));
    F.write_back(false);
    last_report_ = {rep.termination, rep.num_solves, rep.iterations_total, rep.n_trimmed_landmarks, rep.n_depth_blocks,
                    rep.n_repr_blocks, rep.n_gp_blocks, rep.initial_cost, rep.final_cost, rep.time_sec};
    std::string summary = report_string(rep, "adjustPoseOnly");
    if (shim_trace) {
        auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        std::fprintf(stderr, "[shim] adjustPoseOnly: %d landmarks, %d observations: selection copy %.0f us, flatten + prior %.0f us, limo_ba_adjust_pose_only %.0f us, write-back + summary %.0f us\n",
                     F.w.n_lm, F.w.n_obs, us(t_p0, t_p1), us(t_p1, t_p2), us(t_p2, t_p3), us(t_p3, clk::now()));
    }
    return summary;
}

}  // namespace keyframe_bundle_adjustment
