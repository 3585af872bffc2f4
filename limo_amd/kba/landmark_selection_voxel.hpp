// landmark_selection_voxel.hpp — the two landmark schemes the KITTI launch wires into the selector besides cheirality
// (SURVEY §8f-2), restated without PCL / Boost.Geometry / Eigen:
//
//   LandmarkSparsificationSchemeVoxel   internal/landmark_selection_scheme_voxel.hpp:22-60,
//                                       src/landmark_selection_scheme_voxel.cpp:116-233: landmarks in the frame of the
//                                       newest keyframe -> plausibility cut in z -> split by distance to the driven
//                                       path (far / middle / near "pipes") -> voxel-grid thinning of everything inside
//                                       the far pipe -> per field a budget: near = largest image flow, middle = random,
//                                       far = longest tracks.
//   landmark_helpers::*                 src/landmark_selection_scheme_helpers.cpp:14-135
//   LandmarkSelectionSchemeAddDepth     internal/landmark_selection_scheme_add_depth.hpp:24-64,
//                                       src/landmark_selection_scheme_add_depth.cpp:16-76: per configured keyframe,
//                                       force in the N landmarks that pass a predicate and sort lowest by a key
//                                       (e.g. the 20 nearest landmarks with measured depth).
//   LandmarkSparsificationSchemeObservability   internal/landmark_selection_scheme_observability.hpp:28-88,
//                                       src/landmark_selection_scheme_observability.cpp:52-170: the same three fields
//                                       cut by image flow relative to the largest flow instead of by position.
//   LandmarkRejectionSchemeDimensionPlausibility   internal/landmark_selection_scheme_dimension_plausibility.hpp:20-85:
//                                       keep the landmarks inside a box in the frame of the newest keyframe.
//
// Where this restatement decides something the reference leaves to its libraries (documented, not hidden):
//   * voxel thinning: pcl::VoxelGrid emits one CENTROID per occupied voxel and averages the label field with it, so the
//     id the reference maps back is the average of the ids in the voxel; here the representative of a voxel is the
//     landmark closest to the voxel's centroid;
//   * "random" middle-field choice: std::random_shuffle with the global C RNG there, a seeded std::mt19937 here;
//   * ties in the flow / track-length rankings: unspecified by std::partial_sort_copy there, broken by id here;
//   * keyframes are visited in time order (the reference sorts shared_ptr addresses, helpers.cpp:211).
#pragma once
#include <array>
#include <functional>
#include <limits>
#include <random>
#include <tuple>
#include <unordered_map>

#include "landmark_selector.hpp"

namespace keyframe_bundle_adjustment {

namespace landmark_helpers {

// Per landmark: accumulated (or mean) pixel displacement between consecutive keyframes it is seen in, per camera;
// the largest over its cameras.  Landmarks never seen twice by the same camera get no entry.
inline std::map<LandmarkId, double> calcFlow(const std::vector<LandmarkId>& ids,
                                             const std::vector<Keyframe::ConstPtr>& kfs_in_time_order, bool use_mean) {
    std::map<LandmarkId, double> out;
    struct PerCam {  // (a landmark is seen by one or two cameras: a flat list instead of three std::maps per landmark)
        CameraId cam;
        Measurement last;
        double sum;
        int cnt;
    };
    std::vector<PerCam> cams;
    for (const auto& id : ids) {
        cams.clear();
        for (const auto& kf : kfs_in_time_order) {
            const auto im = kf->measurements_.find(id);  // (= getMeasurements(id) without the copy: the cameras that measured it)
            if (im == kf->measurements_.cend()) continue;
            for (const auto& cm : im->second) {
                if (!kf->cameras_.count(cm.first)) continue;
                PerCam* pc = nullptr;
                for (auto& c : cams)
                    if (c.cam == cm.first) pc = &c;
                if (pc) {
                    const double du = double(pc->last.u) - double(cm.second.u), dv = double(pc->last.v) - double(cm.second.v);
                    pc->sum += std::sqrt(du * du + dv * dv);
                    pc->cnt += 1;
                    pc->last = cm.second;
                } else {
                    cams.push_back({cm.first, cm.second, 0., 0});
                }
            }
        }
        double best = -1.;
        bool any = false;
        for (const auto& c : cams) {
            if (c.cnt == 0) continue;
            any = true;
            best = std::max(best, use_mean ? c.sum / c.cnt : c.sum);
        }
        if (any) out.insert(out.end(), {id, best});
    }
    return out;
}
inline std::map<LandmarkId, double> calcFlow(const std::vector<LandmarkId>& ids,
                                             const std::map<KeyframeId, Keyframe::ConstPtr>& keyframes, bool use_mean) {
    std::vector<Keyframe::ConstPtr> kfs;
    for (const auto& k : keyframes) kfs.push_back(k.second);
    std::sort(kfs.begin(), kfs.end(), [](const auto& a, const auto& b) { return a->timestamp_ < b->timestamp_; });
    return calcFlow(ids, kfs, use_mean);
}

// calcFlow for ids SORTED ascending, as (id, flow) in id order: per keyframe ONE merge pass over its measurements and the ids
// instead of a map lookup per (id, keyframe) - same keyframe order, same camera order, same sums as above.
inline std::vector<std::pair<LandmarkId, double>> calcFlowSorted(const std::vector<LandmarkId>& ids_sorted,
                                                                 const std::map<KeyframeId, Keyframe::ConstPtr>& keyframes, bool use_mean) {
    std::vector<Keyframe::ConstPtr> kfs;
    for (const auto& k : keyframes) kfs.push_back(k.second);
    std::sort(kfs.begin(), kfs.end(), [](const auto& a, const auto& b) { return a->timestamp_ < b->timestamp_; });
    struct PerCam {
        CameraId cam;
        Measurement last;
        double sum;
        int cnt;
    };
    constexpr int kCams = 4;  // cameras that see one landmark (more: the general function takes over)
    const size_t n = ids_sorted.size();
    std::vector<PerCam> cams(n * kCams);
    std::vector<unsigned char> n_cams(n, 0);
    bool overflow = false;
    for (const auto& kf : kfs) {
        Keyframe::MeasurementTableScope table_scope;  // (the table is trusted for the duration of this call: keyframe.hpp)
        const auto& rows = kf->measurementTable();  // (landmark id, camera id) ascending: merged with the ids, no map node is visited for an id that is not asked for
        size_t im = 0;
        for (size_t i = 0; i < n && im < rows.size(); ++i) {
            const LandmarkId id = ids_sorted[i];
            if (rows[im].id < id)  // (far ahead most of the time: the near field is a small part of a keyframe's landmarks)
                im = std::lower_bound(rows.begin() + im, rows.end(), id, [](const Keyframe::MeasurementRef& r, LandmarkId v) { return r.id < v; }) - rows.begin();
            if (im == rows.size()) break;
            for (; im < rows.size() && rows[im].id == id; ++im) {
                const CameraId cam = rows[im].cam;
                const Measurement& meas = *rows[im].m;
                if (!kf->cameras_.count(cam)) continue;
                PerCam* pc = nullptr;
                for (int c = 0; c < n_cams[i]; ++c)
                    if (cams[i * kCams + c].cam == cam) pc = &cams[i * kCams + c];
                if (pc) {
                    const double du = double(pc->last.u) - double(meas.u), dv = double(pc->last.v) - double(meas.v);
                    pc->sum += std::sqrt(du * du + dv * dv);
                    pc->cnt += 1;
                    pc->last = meas;
                } else if (n_cams[i] < kCams) {
                    cams[i * kCams + n_cams[i]++] = {cam, meas, 0., 0};
                } else {
                    overflow = true;
                }
            }
        }
    }
    std::vector<std::pair<LandmarkId, double>> out;
    if (overflow) {
        for (const auto& el : calcFlow(ids_sorted, kfs, use_mean)) out.push_back(el);
        return out;
    }
    out.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        double best = -1.;
        bool any = false;
        for (int c = 0; c < n_cams[i]; ++c) {
            const PerCam& pc = cams[i * kCams + c];
            if (pc.cnt == 0) continue;
            any = true;
            best = std::max(best, use_mean ? pc.sum / pc.cnt : pc.sum);
        }
        if (any) out.push_back({ids_sorted[i], best});
    }
    return out;
}
// the max_num ids of largest flow (ties by id), from (id, flow) pairs
inline std::vector<LandmarkId> chooseNearLmIds(size_t max_num, const std::vector<std::pair<LandmarkId, double>>& flow_of_near) {
    std::vector<std::pair<double, LandmarkId>> keyed;
    keyed.reserve(flow_of_near.size());
    for (const auto& el : flow_of_near) keyed.push_back({el.second, el.first});
    std::sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
    keyed.resize(std::min(max_num, keyed.size()));
    std::vector<LandmarkId> ids;
    for (const auto& k : keyed) ids.push_back(k.second);
    return ids;
}
inline std::vector<LandmarkId> chooseNearLmIds(size_t max_num, const std::vector<LandmarkId>& near_ids,
                                               const std::map<LandmarkId, double>& flow) {
    std::vector<std::pair<double, LandmarkId>> keyed;  // (the flow of an id is looked up once, not in every comparison)
    for (const auto& id : near_ids) {
        const auto it = flow.find(id);
        if (it != flow.end()) keyed.push_back({it->second, id});
    }
    std::sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
    keyed.resize(std::min(max_num, keyed.size()));
    std::vector<LandmarkId> ids;
    for (const auto& k : keyed) ids.push_back(k.second);
    return ids;
}
// A uniformly random subset of max_num ids, deterministic in (ids, seed) and STABLE: every id gets the rank
// hash(id, seed) and the max_num smallest ranks are kept, so adding or removing one candidate changes the subset by at
// most one element.  (A shuffle of the candidate list - what std::random_shuffle does in the reference - reorders
// everything when the list grows by one, and a candidate list differs by one as soon as a voxel boundary or a
// representative flips with the last bits of a pose.)
inline std::vector<LandmarkId> chooseMiddleLmIds(size_t max_num, const std::vector<LandmarkId>& middle_ids, uint64_t seed = 0) {
    auto rank = [seed](LandmarkId id) {
        uint64_t x = (uint64_t)id * 0x9E3779B97F4A7C15ull + seed;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    };
    std::vector<std::pair<uint64_t, LandmarkId>> keyed;  // (the rank of an id is computed once, not in every comparison)
    keyed.reserve(middle_ids.size());
    for (const auto& id : middle_ids) keyed.push_back({rank(id), id});
    std::sort(keyed.begin(), keyed.end());
    keyed.resize(std::min(max_num, keyed.size()));
    std::vector<LandmarkId> a;
    a.reserve(keyed.size());
    for (const auto& k : keyed) a.push_back(k.second);
    return a;
}
inline std::vector<LandmarkId> chooseFarLmIds(size_t max_num, const std::vector<LandmarkId>& far_ids,
                                              const std::map<KeyframeId, Keyframe::ConstPtr>& keyframes) {
    std::vector<std::pair<unsigned, LandmarkId>> keyed;
    keyed.reserve(far_ids.size());
    for (const auto& id : far_ids) keyed.push_back({0u, id});
    // Keyframe::hasMeasurement(id) - some camera of the keyframe's rig measured it - per (id, keyframe), as a binary search in the
    // keyframe's measurement table (contiguous) instead of a descent through its map
    for (const auto& kf : keyframes) {
        Keyframe::MeasurementTableScope table_scope;  // (the table is trusted for the duration of this call: keyframe.hpp)
        const auto& rows = kf.second->measurementTable();
        for (auto& k : keyed) {
            auto it = std::lower_bound(rows.begin(), rows.end(), k.second, [](const Keyframe::MeasurementRef& r, LandmarkId id) { return r.id < id; });
            bool seen = false;
            for (; it != rows.end() && it->id == k.second && !seen; ++it) seen = kf.second->cameras_.count(it->cam) != 0;
            k.first += seen ? 1u : 0u;
        }
    }
    std::sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
    keyed.resize(std::min(max_num, keyed.size()));
    std::vector<LandmarkId> ids;
    for (const auto& k : keyed) ids.push_back(k.second);
    return ids;
}

// distance of point p to the polyline through pts (a single point if there is only one)
inline double distanceToPath(const Vector3d& p, const std::vector<Vector3d>& pts) {
    if (pts.empty()) return std::numeric_limits<double>::max();
    if (pts.size() == 1) return (p - pts[0]).norm();
    double best = std::numeric_limits<double>::max();
    for (size_t i = 0; i + 1 < pts.size(); ++i) {
        const Vector3d ab = pts[i + 1] - pts[i], ap = p - pts[i];
        const double l2 = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2];
        double t = l2 > 0. ? (ap[0] * ab[0] + ap[1] * ab[1] + ap[2] * ab[2]) / l2 : 0.;
        t = std::min(1., std::max(0., t));
        best = std::min(best, (p - (pts[i] + ab * t)).norm());
    }
    return best;
}

}  // namespace landmark_helpers

class LandmarkSparsificationSchemeVoxel : public LandmarkSparsificationSchemeBase, public LandmarkCategorizatonInterface {
public:
    struct Parameters {
        std::array<double, 3> voxel_size_xyz{{1.0, 1.0, 0.5}};
        std::array<double, 3> roi_far_xyz{{50., 50., 50.}};     // [0] = radius of the far pipe around the driven path
        std::array<double, 3> roi_middle_xyz{{25., 25., 25.}};  // [0] = radius of the near pipe
        unsigned int max_num_landmarks_near{300};
        unsigned int max_num_landmarks_middle{300};
        unsigned int max_num_landmarks_far{300};
    };
    explicit LandmarkSparsificationSchemeVoxel(Parameters p) : params_(p) {}

    std::set<LandmarkId> getSelection(const LandmarkMap& landmarks, const KeyframeMap& keyframes) const override {
        std::set<LandmarkId> out;
        for (const auto& el : categorized(landmarks, keyframes)) out.insert(out.end(), el.first);
        return out;
    }
    std::vector<LandmarkId> getSelectionSorted(const LandmarkView& landmarks, const KeyframeMap& keyframes) const override {
        std::vector<LandmarkId> out;
        for (const auto& el : categorized(landmarks, keyframes)) out.push_back(el.first);
        return out;
    }
    std::map<LandmarkId, Category> getCategorizedSelection(const LandmarkMap& lms, const KeyframeMap& keyframes) const override {
        std::map<LandmarkId, Category> out;
        for (const auto& el : categorized(lms, keyframes)) out.insert(out.end(), el);
        return out;
    }
    std::vector<std::pair<LandmarkId, Category>> getCategorizedSelectionSorted(const LandmarkView& lms, const KeyframeMap& keyframes) const override {
        return categorized(lms, keyframes);
    }

    template <class Range>  // a LandmarkMap or a LandmarkView; returns (id, category) sorted by id
    std::vector<std::pair<LandmarkId, Category>> categorized(const Range& lms, const KeyframeMap& keyframes) const {
        std::vector<std::pair<LandmarkId, Category>> out;
        if (keyframes.empty()) return out;
        const auto newest = std::max_element(keyframes.cbegin(), keyframes.cend(), [](const auto& a, const auto& b) {
            return a.second->timestamp_ < b.second->timestamp_;
        });
        const EigenPose cur = newest->second->getEigenPose();  // newest keyframe <- origin
        std::vector<Vector3d> path;                            // keyframe positions in the newest keyframe's frame
        for (const auto& kf : keyframes) path.push_back(cur * kf.second->getEigenPose().inverse().translation());

        struct P {
            LandmarkId id;
            Vector3d p;
            double dist;
        };
        std::vector<P> pipe;
        std::vector<LandmarkId> ids_far;
        for (const auto& id_lm : lms) {
            const Vector3d p = cur * Vector3d(id_lm.second->pos.data());
            if (!(p[2] >= -20. && p[2] <= 100.)) continue;  // plausibility cut (voxel.cpp:150-155)
            const double d = landmark_helpers::distanceToPath(p, path);
            if (d < params_.roi_far_xyz[0])
                pipe.push_back({id_lm.first, p, d});
            else
                ids_far.push_back(id_lm.first);
        }
        // voxel grid over the pipe: one representative per occupied voxel.  (voxel key, pipe index) pairs sorted by key, the
        // members of a voxel in pipe order (= landmark id order, the order their coordinates are summed in) - instead of a
        // std::map of cells with a member vector each.
        // The three cell indices packed into ONE 64-bit key that sorts like the triple (21 bits each, offset 2^20: cells within
        // +-1e6 of the origin in every axis, i.e. 500 km at the finest 0.5 m voxel; outside that the triple itself is compared).
        std::vector<std::pair<uint64_t, size_t>> cells(pipe.size());
        bool packed = true;
        auto cell_of = [&](size_t i, long* c) {
            c[0] = (long)std::floor(pipe[i].p[0] / params_.voxel_size_xyz[0]);
            c[1] = (long)std::floor(pipe[i].p[1] / params_.voxel_size_xyz[1]);
            c[2] = (long)std::floor(pipe[i].p[2] / params_.voxel_size_xyz[2]);
        };
        for (size_t i = 0; i < pipe.size() && packed; ++i) {
            long c[3];
            cell_of(i, c);
            const long off = 1l << 20;
            packed = c[0] > -off && c[0] < off && c[1] > -off && c[1] < off && c[2] > -off && c[2] < off;
            cells[i] = {((uint64_t)(c[0] + off) << 42) | ((uint64_t)(c[1] + off) << 21) | (uint64_t)(c[2] + off), i};
        }
        if (packed) {
            std::sort(cells.begin(), cells.end());
        } else {  // (rank the triples instead: the key becomes the triple's position in sorted order)
            std::vector<std::pair<std::array<long, 3>, size_t>> wide(pipe.size());
            for (size_t i = 0; i < pipe.size(); ++i) {
                long c[3];
                cell_of(i, c);
                wide[i] = {{{c[0], c[1], c[2]}}, i};
            }
            std::sort(wide.begin(), wide.end());
            uint64_t key = 0;
            for (size_t q = 0; q < wide.size(); ++q) {
                if (q > 0 && wide[q].first != wide[q - 1].first) ++key;
                cells[q] = {key, wide[q].second};
            }
        }
        std::vector<LandmarkId> ids_near, ids_middle;
        for (size_t c0 = 0; c0 < cells.size();) {
            size_t c1 = c0;
            double sx = 0, sy = 0, sz = 0;
            while (c1 < cells.size() && cells[c1].first == cells[c0].first) {
                sx += pipe[cells[c1].second].p[0];
                sy += pipe[cells[c1].second].p[1];
                sz += pipe[cells[c1].second].p[2];
                ++c1;
            }
            const double n = (double)(c1 - c0);
            const Vector3d centroid(sx / n, sy / n, sz / n);
            size_t best = cells[c0].second;
            double bd = std::numeric_limits<double>::max();
            // nearest member to the centroid; members whose distances agree to 1e-9 relative count as equidistant and the
            // smaller id wins - the two members of a 2-member voxel are ALWAYS equidistant from their midpoint, and which
            // of the two computed distances comes out smaller is rounding noise
            for (size_t q = c0; q < c1; ++q) {
                const size_t i = cells[q].second;
                const double d = (pipe[i].p - centroid).norm();
                const bool tie = std::fabs(d - bd) <= 1e-9 * (d + bd);
                if ((!tie && d < bd) || (tie && pipe[i].id < pipe[best].id)) {
                    bd = d;
                    best = i;
                }
            }
            (pipe[best].dist < params_.roi_middle_xyz[0] ? ids_near : ids_middle).push_back(pipe[best].id);
            c0 = c1;
        }
        std::sort(ids_near.begin(), ids_near.end());  // (voxel order so far; the choice below does not depend on the order)
        const auto flow = landmark_helpers::calcFlowSorted(ids_near, keyframes, false);
        for (const auto& id : landmark_helpers::chooseNearLmIds(params_.max_num_landmarks_near, flow)) out.push_back({id, Category::NearField});
        for (const auto& id : landmark_helpers::chooseMiddleLmIds(params_.max_num_landmarks_middle, ids_middle, newest->second->timestamp_))
            out.push_back({id, Category::MiddleField});
        for (const auto& id : landmark_helpers::chooseFarLmIds(params_.max_num_landmarks_far, ids_far, keyframes)) out.push_back({id, Category::FarField});
        // (the three fields are disjoint; a std::map filled in this order kept the LAST category of an id - none repeats)
        std::stable_sort(out.begin(), out.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        return out;
    }

    static LandmarkSparsificationSchemeBase::ConstPtr createConst(Parameters p) { return std::make_shared<const LandmarkSparsificationSchemeVoxel>(p); }
    static LandmarkSparsificationSchemeBase::Ptr create(Parameters p) { return std::make_shared<LandmarkSparsificationSchemeVoxel>(p); }

private:
    Parameters params_;
};

class LandmarkSelectionSchemeAddDepth : public LandmarkSelectionSchemeBase {
public:
    using FrameIndex = int;       // 0 = oldest active keyframe
    using NumberLandmarks = int;  // how many to force in
    using Comparator = std::function<bool(const Landmark::ConstPtr&)>;               // which landmarks qualify
    using Sorter = std::function<float(const Measurement&, const Vector3d& local_lm)>;  // key; the SMALLEST win
    struct Parameters {
        std::vector<std::tuple<FrameIndex, NumberLandmarks, Comparator, Sorter>> params_per_keyframe;
    };
    explicit LandmarkSelectionSchemeAddDepth(Parameters p) : params_(p) {}

    std::set<LandmarkId> getSelection(const LandmarkMap& landmarks, const KeyframeMap& keyframes) const override {
        const std::vector<LandmarkId> v = picked(landmarks, keyframes);
        return std::set<LandmarkId>(v.begin(), v.end());
    }
    std::vector<LandmarkId> getSelectionSorted(const LandmarkView& landmarks, const KeyframeMap& keyframes) const override {
        return picked(landmarks, keyframes);
    }
    template <class Range>  // a LandmarkMap or a LandmarkView; returns the picked ids, sorted, each once
    std::vector<LandmarkId> picked(const Range& landmarks, const KeyframeMap& keyframes) const {
        std::vector<LandmarkId> out;
        std::vector<Keyframe::ConstPtr> kfs;  // active keyframes, oldest first
        for (const auto& kf : keyframes)
            if (kf.second->is_active_) kfs.push_back(kf.second);
        std::sort(kfs.begin(), kfs.end(), [](const auto& a, const auto& b) { return a->timestamp_ < b->timestamp_; });
        // the landmarks every configuration's predicate lets through, collected in ONE pass over the landmark map
        const size_t n_cfg = params_.params_per_keyframe.size();
        std::vector<std::vector<typename Range::const_iterator>> qualifies(n_cfg);
        for (auto it = landmarks.cbegin(); it != landmarks.cend(); ++it)
            for (size_t c = 0; c < n_cfg; ++c) {
                const FrameIndex ind = std::get<0>(params_.params_per_keyframe[c]);
                if (ind < 0 || ind > (int)kfs.size() - 1) continue;
                if (std::get<2>(params_.params_per_keyframe[c])(it->second)) qualifies[c].push_back(it);
            }
        for (size_t cfg = 0; cfg < n_cfg; ++cfg) {
            const auto& el = params_.params_per_keyframe[cfg];
            const FrameIndex ind = std::get<0>(el);
            if (ind < 0 || ind > (int)kfs.size() - 1) continue;
            const Keyframe& kf = *kfs[ind];
            std::vector<std::pair<LandmarkId, double>> keyed;
            const EigenPose T = kf.getEigenPose();
            // the landmarks that qualify (a fifth of them with the ground-plane predicate), then their measurements in this
            // keyframe - both in id order: a merge with the keyframe's measurement table (binary search ahead when the next id is far)
            Keyframe::MeasurementTableScope table_scope;  // (the table is trusted for the duration of this call: keyframe.hpp)
            const auto& rows = kf.measurementTable();
            size_t im = 0;
            for (const auto& lm_it : qualifies[cfg]) {
                const auto& lm = *lm_it;
                if (im < rows.size() && rows[im].id < lm.first)
                    im = std::lower_bound(rows.begin() + im, rows.end(), lm.first, [](const Keyframe::MeasurementRef& r, LandmarkId v) { return r.id < v; }) - rows.begin();
                if (im == rows.size()) break;
                if (rows[im].id != lm.first) continue;
                const Vector3d local = T * Vector3d(lm.second->pos.data());
                double worst = -std::numeric_limits<double>::max();  // largest key over the cameras that see it
                for (; im < rows.size() && rows[im].id == lm.first; ++im) worst = std::max(worst, (double)std::get<3>(el)(*rows[im].m, local));
                keyed.push_back({lm.first, worst});
            }
            std::sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) {
                return a.second < b.second || (a.second == b.second && a.first < b.first);
            });
            const int n = std::min(std::get<1>(el), (int)keyed.size());
            for (int i = 0; i < n; ++i) out.push_back(keyed[i].first);
        }
        std::sort(out.begin(), out.end());
        out.erase(std::unique(out.begin(), out.end()), out.end());
        return out;
    }
    static LandmarkSelectionSchemeBase::ConstPtr createConst(Parameters p) { return std::make_shared<const LandmarkSelectionSchemeAddDepth>(p); }
    static LandmarkSelectionSchemeBase::Ptr create(Parameters p) { return std::make_shared<LandmarkSelectionSchemeAddDepth>(p); }

private:
    Parameters params_;
};

// Fields by observability: |mean image flow| relative to the largest flow among the landmarks - at least
// bound_near_middle of it = near field, at most bound_middle_far = far field, middle in between.  Near and middle prefer
// landmarks with a measured depth (their budget is filled from those first); near takes the largest flows, middle a
// random subset, far the longest tracks.
class LandmarkSparsificationSchemeObservability : public LandmarkSparsificationSchemeBase, public LandmarkCategorizatonInterface {
public:
    struct BinParameters {
        unsigned int max_num_landmarks_near{300}, max_num_landmarks_middle{300}, max_num_landmarks_far{300};
        double bound_near_middle = 0.4;  // fractions of the largest flow
        double bound_middle_far = 0.2;
    };
    struct Parameters {
        int histogram_cache_size{500};  // kept for source compatibility; the bounds are fractions, no histogram is built
        BinParameters bin_params_;
    };
    explicit LandmarkSparsificationSchemeObservability(Parameters p) : params_(p) { identifier = "observability"; }

    std::set<LandmarkId> getSelection(const LandmarkMap& landmarks, const KeyframeMap& keyframes) const override {
        std::set<LandmarkId> out;
        for (const auto& el : getCategorizedSelection(landmarks, keyframes)) out.insert(el.first);
        return out;
    }

    std::map<LandmarkId, Category> getCategorizedSelection(const LandmarkMap& landmarks, const KeyframeMap& keyframes) const override {
        std::map<LandmarkId, Category> out;
        std::vector<LandmarkId> ids;
        for (const auto& el : landmarks) ids.push_back(el.first);
        std::map<LandmarkId, double> flow = landmark_helpers::calcFlow(ids, keyframes, true);
        if (flow.empty()) return out;
        double max_flow = 0.;
        for (auto& el : flow) {
            el.second = std::abs(el.second);
            max_flow = std::max(max_flow, el.second);
        }
        const BinParameters& b = params_.bin_params_;
        std::vector<LandmarkId> near_d, near_n, mid_d, mid_n, far;
        for (const auto& el : landmarks) {
            const auto it = flow.find(el.first);
            if (it == flow.end()) continue;  // never seen twice by one camera: no flow (the reference's map_data.at would throw)
            const bool depth = el.second->has_measured_depth;
            if (it->second >= b.bound_near_middle * max_flow)
                (depth ? near_d : near_n).push_back(el.first);
            else if (it->second > b.bound_middle_far * max_flow)
                (depth ? mid_d : mid_n).push_back(el.first);
            else
                far.push_back(el.first);
        }
        std::vector<LandmarkId> near = landmark_helpers::chooseNearLmIds(b.max_num_landmarks_near, near_d, flow);
        for (const auto& id : landmark_helpers::chooseNearLmIds(b.max_num_landmarks_near - near.size(), near_n, flow)) near.push_back(id);
        // (the random subset is drawn afresh per solve, like the reference's shuffle: the newest keyframe's stamp is the seed - with a
        // fixed seed the same ids would win for a whole drive)
        TimestampNSec newest = 0;
        for (const auto& kf : keyframes) newest = std::max(newest, kf.second->timestamp_);
        std::vector<LandmarkId> mid = landmark_helpers::chooseMiddleLmIds(b.max_num_landmarks_middle, mid_d, newest);
        for (const auto& id : landmark_helpers::chooseMiddleLmIds(b.max_num_landmarks_middle - mid.size(), mid_n, newest)) mid.push_back(id);
        for (const auto& id : near) out[id] = Category::NearField;
        for (const auto& id : mid) out[id] = Category::MiddleField;
        for (const auto& id : landmark_helpers::chooseFarLmIds(b.max_num_landmarks_far, far, keyframes)) out[id] = Category::FarField;
        return out;
    }

    static ConstPtr createConst(Parameters p) { return ConstPtr(new LandmarkSparsificationSchemeObservability(p)); }
    static Ptr create(Parameters p) { return Ptr(new LandmarkSparsificationSchemeObservability(p)); }

    Parameters params_;
};

// Keep the landmarks that lie inside an axis-aligned box in the frame of the NEWEST keyframe (largest id).
class LandmarkRejectionSchemeDimensionPlausibility : public LandmarkRejectionSchemeBase {
public:
    struct Params {
        // defaults as in the reference: numeric_limits<double>::min() is the smallest POSITIVE double, so an unset lower
        // bound still demands a positive coordinate
        double min_x{std::numeric_limits<double>::min()}, max_x{std::numeric_limits<double>::max()};
        double min_y{std::numeric_limits<double>::min()}, max_y{std::numeric_limits<double>::max()};
        double min_z{std::numeric_limits<double>::min()}, max_z{std::numeric_limits<double>::max()};
    };
    explicit LandmarkRejectionSchemeDimensionPlausibility(const Params& p) : params_(p) { identifier = "dimension plausibility"; }

    std::set<LandmarkId> getSelection(const LandmarkMap& landmarks, const KeyframeMap& keyframes) const override {
        std::set<LandmarkId> out;
        if (keyframes.empty()) return out;
        const EigenPose T = keyframes.rbegin()->second->getEigenPose();
        for (const auto& el : landmarks) {
            const Vector3d p = T * Vector3d(el.second->pos[0], el.second->pos[1], el.second->pos[2]);
            if (p[0] > params_.min_x && p[0] < params_.max_x && p[1] > params_.min_y && p[1] < params_.max_y && p[2] > params_.min_z &&
                p[2] < params_.max_z)
                out.insert(el.first);
        }
        return out;
    }

    static ConstPtr createConst(const Params& p) { return ConstPtr(new LandmarkRejectionSchemeDimensionPlausibility(p)); }
    static Ptr create(const Params& p) { return Ptr(new LandmarkRejectionSchemeDimensionPlausibility(p)); }

private:
    Params params_;
};

}  // namespace keyframe_bundle_adjustment
