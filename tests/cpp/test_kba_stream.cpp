// Streaming visual-odometry scenario on a synthetic sequence, through the keyframe_bundle_adjustment API of the shim
// (limo_amd/kba): the per-frame call order of the reference's ROS node, restated without ROS
//   keyframe_bundle_adjustment_ros_tool/src/mono_lidar/mono_lidar.cpp:186-260
//   build the frame with a motion prior -> adjustPoseOnly -> push -> deactivateKeyframes -> solve -> dump the pose
// (every frame is taken as a keyframe; SURVEY §8f-3's keyframe selection stays on the host and is not the subject).
// Checks the trajectory against the synthetic ground truth (absolute trajectory error) and the window bookkeeping.
// Linked against the emulated C-ABI in the CPU test tier and against liblimo_hip.so in the GPU tier.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "../../limo_amd/kba/bundle_adjuster_keyframes.hpp"
#include "../../limo_amd/kba/keyframe_selector.hpp"
#include "../../limo_amd/kba/landmark_selection_voxel.hpp"

using namespace keyframe_bundle_adjustment;
using matches_msg_types::FeaturePoint;
using matches_msg_types::Tracklets;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        ++g_checks;                                                                  \
        if (!(cond)) {                                                               \
            ++g_fail;                                                                \
            std::printf("CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
        }                                                                            \
    } while (0)

static std::mt19937_64 rng(7);
static double gauss(double s) {
    return s > 0 ? std::normal_distribution<double>(0., s)(rng) : 0.;
}
static double uni(double a, double b) {
    return std::uniform_real_distribution<double>(a, b)(rng);
}

// Unit checks of the keyframe schemes (keyframe_*_scheme_*.cpp) on hand-made frames.
static void test_keyframe_schemes() {
    auto frame = [](uint64_t stamp, float du, double yaw) {
        Tracklets ts;
        ts.stamps = {stamp};
        for (int i = 0; i < 20; ++i) {
            matches_msg_types::Tracklet tr;
            tr.id = i;
            tr.feature_points.push_back(FeaturePoint(100.f + 10.f * i + du, 50.f));
            ts.tracks.push_back(tr);
        }
        EigenPose p = EigenPose::Identity();
        p.rotate(yaw, Vector3d(0., 0., 1.));
        return std::make_shared<Keyframe>(stamp, ts, std::make_shared<Camera>(600., Vector2d(200., 100.), EigenPose::Identity()), p);
    };
    std::map<KeyframeId, Keyframe::Ptr> none, last{{0, frame(1000000000ull, 0.f, 0.)}};
    KeyframeRejectionSchemeFlow flow(5.);
    CHECK(flow.isUsable(frame(1100000000ull, 0.f, 0.), none));      // nothing to compare with
    CHECK(!flow.isUsable(frame(1100000000ull, 3.f, 0.), last));     // 3 px mean flow < 5 px
    CHECK(flow.isUsable(frame(1100000000ull, 8.f, 0.), last));
    KeyframeSelectionSchemePose pose(0.04);
    CHECK(!pose.isUsable(frame(1100000000ull, 0.f, 0.3), none));    // never selects without a previous keyframe
    CHECK(!pose.isUsable(frame(1100000000ull, 0.f, 0.02), last));
    CHECK(pose.isUsable(frame(1100000000ull, 0.f, 0.06), last));
    CHECK(std::fabs(calcQuaternionDiff(frame(0, 0.f, 0.25)->pose_, frame(0, 0.f, -0.15)->pose_) - 0.4) < 1e-12);
    KeyframeSparsificationSchemeTime time(0.3e9);
    CHECK(time.isUsable(frame(1100000000ull, 0.f, 0.), none));
    CHECK(!time.isUsable(frame(1200000000ull, 0.f, 0.), last));
    CHECK(time.isUsable(frame(1400000000ull, 0.f, 0.), last));
    // selector: rejected frames never pass; otherwise selection OR sparsification decides
    KeyframeSelector sel;
    sel.addScheme(KeyframeRejectionSchemeFlow::createConst(5.));
    sel.addScheme(KeyframeSelectionSchemePose::createConst(0.04));
    sel.addScheme(KeyframeSparsificationSchemeTime::createConst(0.3e9));
    CHECK(sel.select({frame(1400000000ull, 8.f, 0.)}, last).size() == 1);   // enough time
    CHECK(sel.select({frame(1100000000ull, 8.f, 0.)}, last).size() == 0);   // too early, no turn
    CHECK(sel.select({frame(1100000000ull, 8.f, 0.1)}, last).size() == 1);  // too early but turning
    CHECK(sel.select({frame(1400000000ull, 1.f, 0.1)}, last).size() == 0);  // no image motion: rejected whatever else
    CHECK(sel.select({frame(1100000000ull, 0.f, 0.)}, none).size() == 1);   // very first frame
}

// Unit checks of the voxel / add-depth landmark schemes and their helpers on a hand-made scene.
static void test_landmark_schemes() {
    using Cat = LandmarkCategorizatonInterface::Category;
    // two keyframes 2 m apart along x (identity rotations), one camera looking along +x of the vehicle frame
    auto cam = std::make_shared<Camera>(500., Vector2d(320., 240.), EigenPose::Identity());
    std::vector<Vector3d> pts;
    for (int i = 0; i < 40; ++i) pts.push_back(Vector3d(5.0 + 0.01 * i, 0.5, 1.0));  // 40 points inside ONE 1 x 1 x 0.5 voxel, 5 m from the path
    pts.push_back(Vector3d(3.0, 30.0, 1.0));    // id 40: 30 m from the path -> middle field (25 <= d < 50)
    pts.push_back(Vector3d(3.0, 70.0, 1.0));    // id 41: 70 m -> far field
    pts.push_back(Vector3d(3.0, 0.0, 150.0));   // id 42: z > 100 in the newest frame: implausible, dropped
    Tracklets t0, t1;
    t0.stamps = {100};
    t1.stamps = {200};
    for (size_t i = 0; i < pts.size(); ++i) {
        matches_msg_types::Tracklet a, b;
        a.id = b.id = i;
        a.feature_points.push_back(FeaturePoint(100.f, 100.f));
        b.feature_points.push_back(FeaturePoint(100.f + (float)i, 100.f));  // flow grows with the id
        t0.tracks.push_back(a);
        if (i != 41) t1.tracks.push_back(b);  // id 41 is seen once only
    }
    EigenPose p1 = EigenPose::Identity();
    p1.translate(Vector3d(-2., 0., 0.));  // keyframe <- origin of a vehicle 2 m further along x
    std::map<KeyframeId, Keyframe::ConstPtr> kfs{{100, std::make_shared<Keyframe>(100, t0, cam, EigenPose::Identity())},
                                                 {200, std::make_shared<Keyframe>(200, t1, cam, p1)}};
    std::map<LandmarkId, Landmark::ConstPtr> lms;
    for (size_t i = 0; i < pts.size(); ++i) lms[i] = std::make_shared<Landmark>(pts[i], i % 2 == 0);
    LandmarkSparsificationSchemeVoxel::Parameters vp;
    LandmarkSparsificationSchemeVoxel voxel(vp);
    const auto cat = voxel.getCategorizedSelection(lms, kfs);
    int n_near = 0;
    for (const auto& c : cat) n_near += c.second == Cat::NearField;
    CHECK(n_near == 1);                                        // the 40 co-located points collapse to one representative
    CHECK(cat.count(40) && cat.at(40) == Cat::MiddleField);
    CHECK(cat.count(41) && cat.at(41) == Cat::FarField);
    CHECK(!cat.count(42));
    CHECK(voxel.getSelection(lms, kfs).size() == cat.size());
    // helpers
    std::vector<LandmarkId> ids{1, 2, 3, 41};
    const auto flow = landmark_helpers::calcFlow(ids, kfs, false);
    CHECK(flow.size() == 3 && !flow.count(41));                // seen once: no flow
    CHECK(std::fabs(flow.at(3) - 3.0) < 1e-6);
    const auto near2 = landmark_helpers::chooseNearLmIds(2, ids, flow);
    CHECK(near2.size() == 2 && near2[0] == 3 && near2[1] == 2);  // largest flow first
    const auto far1 = landmark_helpers::chooseFarLmIds(1, {41, 5}, kfs);
    CHECK(far1.size() == 1 && far1[0] == 5);                   // two observations beat one
    CHECK(landmark_helpers::chooseMiddleLmIds(3, {7, 8, 9, 10, 11}).size() == 3);
    CHECK(std::fabs(landmark_helpers::distanceToPath(Vector3d(1., 3., 0.), {Vector3d(0., 0., 0.), Vector3d(2., 0., 0.)}) - 3.) < 1e-12);
    CHECK(std::fabs(landmark_helpers::distanceToPath(Vector3d(5., 4., 0.), {Vector3d(0., 0., 0.), Vector3d(2., 0., 0.)}) - 5.) < 1e-12);
    // add-depth: in the oldest keyframe force in the 3 landmarks with measured depth that have the smallest id-derived key
    LandmarkSelectionSchemeAddDepth::Parameters ap;
    ap.params_per_keyframe.push_back(std::make_tuple(0, 3, [](const Landmark::ConstPtr& lm) { return lm->has_measured_depth; },
                                                     [](const Measurement&, const Vector3d& local) { return (float)local.norm(); }));
    ap.params_per_keyframe.push_back(std::make_tuple(7, 3, [](const Landmark::ConstPtr&) { return true; },
                                                     [](const Measurement&, const Vector3d&) { return 0.f; }));  // no such keyframe
    LandmarkSelectionSchemeAddDepth add(ap);
    const auto forced = add.getSelection(lms, kfs);
    CHECK(forced.size() == 3 && forced.count(0) && forced.count(2) && forced.count(4));  // nearest even ids (has depth)
    // observability: flow of landmark i is i px (41 has none); largest = 42 -> near from 16.8 px, far up to 8.4 px
    LandmarkSparsificationSchemeObservability::Parameters op;
    op.bin_params_.max_num_landmarks_near = 4;
    op.bin_params_.max_num_landmarks_middle = 3;
    op.bin_params_.max_num_landmarks_far = 2;
    LandmarkSparsificationSchemeObservability obs(op);
    const auto ocat = obs.getCategorizedSelection(lms, kfs);
    int on = 0, om = 0, of = 0;
    for (const auto& c : ocat) {
        on += c.second == Cat::NearField;
        om += c.second == Cat::MiddleField;
        of += c.second == Cat::FarField;
        if (c.second == Cat::NearField) CHECK(c.first >= 17);
        if (c.second == Cat::MiddleField) CHECK(c.first > 8 && c.first < 17);
        if (c.second == Cat::FarField) CHECK(c.first <= 8);
    }
    CHECK(on == 4 && om == 3 && of == 2 && !ocat.count(41));
    CHECK(ocat.count(42) && ocat.count(40) && ocat.count(38) && ocat.count(36));  // near: largest flows among those WITH depth (even ids) first
    CHECK(obs.getSelection(lms, kfs).size() == 9);
    // dimension plausibility: box in the frame of the newest keyframe (2 m further along x)
    LandmarkRejectionSchemeDimensionPlausibility::Params dp;
    dp.max_x = 3.2;   // x_new = x - 2: the 40 clustered points sit at 3.0 .. 3.39
    dp.max_y = 10.;
    dp.max_z = 100.;
    LandmarkRejectionSchemeDimensionPlausibility dim(dp);
    const auto plausible = dim.getSelection(lms, kfs);
    CHECK(plausible.size() == 20 && plausible.count(0) && plausible.count(19) && !plausible.count(20) && !plausible.count(40) && !plausible.count(42));
}

// Three N(0,1) draws that depend only on `key` (counter-based: splitmix64 + Box-Muller), so a measurement is the same
// in every tracklet message that repeats it and costs no generator seeding.
static void hash_normals(uint64_t key, double* out) {
    auto next = [&key]() {
        key += 0x9e3779b97f4a7c15ull;
        uint64_t z = key;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z ^= z >> 31;
        return ((double)(z >> 11) + 0.5) * (1.0 / 9007199254740992.0);  // (0, 1)
    };
    const double two_pi = 6.283185307179586;
    const double r0 = std::sqrt(-2.0 * std::log(next())), a0 = two_pi * next();
    const double r1 = std::sqrt(-2.0 * std::log(next())), a1 = two_pi * next();
    out[0] = r0 * std::cos(a0);
    out[1] = r0 * std::sin(a0);
    out[2] = r1 * std::cos(a1);
}

int main(int argc, char** argv) {
    test_keyframe_schemes();
    test_landmark_schemes();
    const int n_frames = argc > 1 ? std::atoi(argv[1]) : 24;
    const int n_lm = argc > 2 ? std::atoi(argv[2]) : 1500;
    if (const char* sd = std::getenv("STREAM_SEED")) rng.seed((uint64_t)std::atoll(sd));  // other scenes than the default one
    const bool long_run = argc > 3;  // any third argument: report fps / ATE of a long drive, no accuracy thresholds
    const int window = 5, history = 10;
    // camera <- vehicle (vehicle x forward, y left, z up; camera z forward, x right, y down), KITTI-like intrinsics
    EigenPose cam_veh = EigenPose::Identity();
    {
        const double R[9] = {0, -1, 0, 0, 0, -1, 1, 0, 0};
        for (int i = 0; i < 9; ++i) cam_veh.R[i] = R[i];
        const Vector3d t_veh_cam(1.08, 0., 1.35);  // camera position in the vehicle frame
        const Vector3d t = cam_veh * Vector3d(-t_veh_cam[0], -t_veh_cam[1], -t_veh_cam[2]);
        cam_veh.t[0] = t[0] - cam_veh.t[0];
        cam_veh.t[1] = t[1] - cam_veh.t[1];
        cam_veh.t[2] = t[2] - cam_veh.t[2];
    }
    const double f = 718.856, cx = 607.1928, cy = 185.2157, W = 1241., H = 376.;
    Camera::Ptr cam = std::make_shared<Camera>(f, Vector2d(cx, cy), cam_veh);

    // ground truth: origin <- vehicle_t, forward motion with a slow turn; keyframe <- origin = inverse
    std::vector<EigenPose> origin_veh(n_frames);
    {
        EigenPose p = EigenPose::Identity();
        if (const char* y0 = std::getenv("STREAM_YAW0")) p.rotate(std::atof(y0), Vector3d(0., 0., 1.));  // start heading
        for (int t = 0; t < n_frames; ++t) {
            origin_veh[t] = p;
            p.translate(Vector3d(0.55, 0., 0.));
            p.rotate(0.006, Vector3d(0., 0., 1.));
        }
    }
    // landmarks in the origin frame along the route; 20 % on the ground plane (z = -0.31 under the vehicle origin)
    std::vector<Vector3d> lms(n_lm);
    std::vector<char> on_ground(n_lm);
    std::vector<int> anchor_of(n_lm);
    for (int i = 0; i < n_lm; ++i) {
        const int anchor = (int)uni(0, n_frames - 1);
        anchor_of[i] = anchor;
        on_ground[i] = uni(0, 1) < 0.2;
        const Vector3d local(uni(4., 45.), uni(-12., 12.), on_ground[i] ? -0.31 : uni(-0.2, 4.0));
        lms[i] = origin_veh[anchor] * local;
    }
    auto project = [&](int t, int i, double& u, double& v, double& z) {
        const Vector3d pc = cam_veh * (origin_veh[t].inverse() * lms[i]);
        z = pc[2];
        if (z < 1.0 || z > 60.) return false;
        u = f * pc[0] / z + cx;
        v = f * pc[1] / z + cy;
        return u >= 0 && u < W && v >= 0 && v < H;
    };
    // A landmark can only be in view within [-130, +110] frames of its anchor (4..45 m ahead of it, seen from 1..60 m
    // at 0.55 m per frame): frames look at that slice of the anchor-sorted list instead of at every landmark.
    std::vector<int> by_anchor(n_lm);
    for (int i = 0; i < n_lm; ++i) by_anchor[i] = i;
    std::stable_sort(by_anchor.begin(), by_anchor.end(), [&](int a, int b) { return anchor_of[a] < anchor_of[b]; });
    std::vector<int> first_with_anchor(n_frames + 1, n_lm);
    for (int j = n_lm - 1; j >= 0; --j) first_with_anchor[anchor_of[by_anchor[j]]] = j;
    for (int t = n_frames - 1; t >= 0; --t) first_with_anchor[t] = std::min(first_with_anchor[t], first_with_anchor[t + 1]);
    std::vector<char> has_depth(n_lm);
    for (int i = 0; i < n_lm; ++i) has_depth[i] = uni(0, 1) < 0.45;

    BundleAdjusterKeyframes ba;
    ba.set_solver_time(20.);
    // keyframe selection as the KITTI launch wires it (keyframe_ba_monolid.launch: time between keyframes, critical
    // rotation difference, minimum flow), scaled to this sequence's 0.05 s frame spacing
    KeyframeSelector selector;
    selector.addScheme(KeyframeRejectionSchemeFlow::createConst(3.));
    selector.addScheme(KeyframeSelectionSchemePose::createConst(0.1));
    selector.addScheme(KeyframeSparsificationSchemeTime::createConst(0.09e9));
    std::vector<char> is_kf(n_frames, 0);
    // landmark selection as the KITTI launch wires it (keyframe_ba_monolid.launch:36-38, mono_lidar.cpp:396-430):
    // voxel sparsification with 200 / 200 / 100 budgets on top of the default cheirality rejection, plus "always keep
    // the 20 nearest depth landmarks and 20 nearest ground landmarks of the oldest keyframe"
    {
        LandmarkSparsificationSchemeVoxel::Parameters vp;
        vp.max_num_landmarks_near = 200;
        vp.max_num_landmarks_middle = 200;
        vp.max_num_landmarks_far = 100;
        vp.roi_far_xyz = {{40., 40., 40.}};
        vp.roi_middle_xyz = {{15., 15., 15.}};
        if (!std::getenv("STREAM_NO_VOXEL")) ba.landmark_selector_->addScheme(LandmarkSparsificationSchemeVoxel::createConst(vp));
        LandmarkSelectionSchemeAddDepth::Parameters ap;
        ap.params_per_keyframe.push_back(std::make_tuple(0, 20, [](const Landmark::ConstPtr& lm) { return lm->has_measured_depth; },
                                                         [](const Measurement& m, const Vector3d&) { return m.d; }));
        ap.params_per_keyframe.push_back(std::make_tuple(0, 20, [](const Landmark::ConstPtr& lm) { return lm->is_ground_plane; },
                                                         [](const Measurement&, const Vector3d& local) { return (float)local.norm(); }));
        if (!std::getenv("STREAM_NO_ADDDEPTH")) ba.landmark_selector_->addScheme(LandmarkSelectionSchemeAddDepth::createConst(ap));
    }
    std::map<uint64_t, int> frame_of_stamp;
    std::vector<EigenPose> est(n_frames);  // keyframe <- origin estimates as dumped right after each solve
    EigenPose last_motion = EigenPose::Identity();
    double t_solve = 0., t_ba = 0.;
    int n_solves = 0;
    const int t_stop = std::getenv("STREAM_STOP") ? std::atoi(std::getenv("STREAM_STOP")) : n_frames;  // debugging aid
    for (int t = 0; t < n_frames && t < t_stop; ++t) {
        // tracklets of this frame: every landmark visible now, with its history over the consecutive frames it was seen
        Tracklets ts;
        for (int k = 0; k < history && t - k >= 0; ++k) ts.stamps.push_back((uint64_t)(t - k) * 50000000ull + 1000ull);
        frame_of_stamp[ts.stamps[0]] = t;
        const int j_lo = first_with_anchor[std::max(0, t - 130)], j_hi = first_with_anchor[std::min(n_frames, t + 111)];
        std::vector<int> in_range(by_anchor.begin() + j_lo, by_anchor.begin() + j_hi);
        std::sort(in_range.begin(), in_range.end());  // tracklets in landmark-id order, as before
        for (int i : in_range) {
            matches_msg_types::Tracklet tr;
            tr.id = i;
            for (int k = 0; k < (int)ts.stamps.size(); ++k) {
                double u, v, z;
                if (!project(t - k, i, u, v, z)) break;
                // deterministic per (frame, landmark) noise so that a measurement is the same in every tracklet message
                double n3[3];
                hash_normals((uint64_t)(t - k) * 1000003ull + (uint64_t)i, n3);
                u += 0.3 * n3[0];
                v += 0.3 * n3[1];
                const double d = z + 0.03 * n3[2];
                tr.feature_points.push_back(has_depth[i] ? FeaturePoint((float)u, (float)v, (float)d) : FeaturePoint((float)u, (float)v));
            }
            if (tr.feature_points.size() >= 1) {
                tr.age = tr.feature_points.size();
                tr.label = on_ground[i] ? 7 : 11;  // cityscapes: road / building (labels_ of the adjuster)
                ts.tracks.push_back(tr);
            }
        }
        // motion prior: constant velocity from the last two estimates, perturbed (mono_lidar.cpp:150-185)
        EigenPose prior = origin_veh[0].inverse();
        if (t == 1) prior = origin_veh[1].inverse();
        if (t >= 2) prior = last_motion * est[t - 1];
        if (t >= 2 && ba.keyframes_.size() < 3) prior = origin_veh[t].inverse();  // bootstrap: no motion-only refinement yet
        if (t >= 1) {
            prior.translate(Vector3d(gauss(0.05), gauss(0.03), gauss(0.02)));
            prior.rotate(gauss(0.004), Vector3d(0., 0., 1.));
        }
        {   // The prior is a product of estimated transforms: bring its rotation back onto SO(3).  Without this the
            // matrix -> quaternion -> matrix round trips of the loop (convert() does not normalise, like the
            // reference's definitions.cpp:14-28) let |q| drift; the trace <= 0 branch of the conversion amplifies the
            // drift, so a drive diverges a few frames after its heading passes 90 degrees.
            Pose q = convert(prior);
            const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            for (int i = 0; i < 4; ++i) q[i] /= n;
            prior = convert(q);
        }
        Plane gp;
        gp.distance = 0.31;  // height over ground (launch file), normal +z in the vehicle frame
        gp.direction = {{0., 0., 1.}};
        const auto t_frame0 = std::chrono::steady_clock::now();
        auto cur = std::make_shared<Keyframe>(ts.stamps[0], ts, cam, prior,
                                              t == 0 ? Keyframe::FixationStatus::Pose : Keyframe::FixationStatus::None, gp);
        if (ba.keyframes_.size() >= 3 && !std::getenv("STREAM_NO_POSEONLY")) {
            ba.adjustPoseOnly(*cur);
            CHECK(ba.last_report_.termination == 0 || ba.last_report_.termination == 1);
            if (ba.last_report_.termination != 0 && ba.last_report_.termination != 1)
                std::printf("frame %d adjustPoseOnly: termination %d, cost %.6e -> %.6e, %d iterations, %d solves\n", t, ba.last_report_.termination,
                            ba.last_report_.initial_cost, ba.last_report_.final_cost, ba.last_report_.iterations_total, ba.last_report_.num_solves);
        }
        const auto selected = selector.select({cur}, ba.getActiveKeyframePtrs());
        CHECK(selected.size() < 2);
        auto count_bad = [&](const char* where) {
            if (!std::getenv("STREAM_TRACE")) return;
            int bad = 0;
            unsigned long first = 0;
            for (const auto& kv : ba.landmarks_)
                if (!std::isfinite(kv.second->pos[0]) || !std::isfinite(kv.second->pos[1]) || !std::isfinite(kv.second->pos[2])) {
                    if (!bad) first = kv.first;
                    ++bad;
                }
            if (bad) std::printf("      [%s, frame %d] %d landmarks with non-finite position, first id %lu (depth %d, ground %d)\n", where, t, bad, first,
                                 (int)ba.landmarks_.at(first)->has_measured_depth, (int)ba.landmarks_.at(first)->is_ground_plane);
        };
        count_bad("before push");
        for (const auto& kf : selected) ba.push(*kf);
        count_bad("after push");
        is_kf[t] = !selected.empty();
        if (!selected.empty() && ba.keyframes_.size() > 2) {
            ba.deactivateKeyframes(3, 3, window);
            ba.updateLabels(ts, 0.9);
            CHECK((int)ba.active_keyframe_ids_.size() <= window);
            const auto t0 = std::chrono::steady_clock::now();
            const std::string summary = ba.solve();
            count_bad("after solve");
            t_solve += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            ++n_solves;
            CHECK(!summary.empty());
            CHECK(ba.last_report_.termination == 0 || ba.last_report_.termination == 1);
            if (ba.last_report_.termination != 0 && ba.last_report_.termination != 1)
                std::printf("frame %d solve: termination %d, cost %.6e -> %.6e, %d iterations, %d solves, %d trimmed\n", t, ba.last_report_.termination,
                            ba.last_report_.initial_cost, ba.last_report_.final_cost, ba.last_report_.iterations_total, ba.last_report_.num_solves,
                            ba.last_report_.n_trimmed_landmarks);
            CHECK(ba.last_report_.final_cost <= ba.last_report_.initial_cost || ba.last_report_.initial_cost < 0);
            CHECK(ba.selected_landmark_ids_.size() <= 200 + 200 + 100 + 40);  // the budgets of the voxel scheme + add-depth
        }
        // the node dumps the optimised pose when the frame became a keyframe, its prior otherwise (:281-294)
        est[t] = is_kf[t] ? ba.getKeyframe().getEigenPose() : cur->getEigenPose();
        t_ba += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_frame0).count();
        if (std::getenv("STREAM_TRACE")) {
            const Vector3d e = est[t].inverse().translation() - origin_veh[t].translation();
            std::printf("frame %d kf %d err %.4f m, tracks %zu, term %d, cost %.4g -> %.4g, its %d, selected %zu\n", t, (int)is_kf[t], e.norm(),
                        ts.tracks.size(), ba.last_report_.termination, ba.last_report_.initial_cost, ba.last_report_.final_cost,
                        ba.last_report_.iterations_total, ba.selected_landmark_ids_.size());
            // landmark errors: selected set, and all landmarks with depth / without
            std::vector<double> es, ed, en;
            for (const auto& kv : ba.landmarks_) {
                const auto& L = *kv.second;
                const Vector3d& g = lms[kv.first];
                const double e2 = std::sqrt((L.pos[0] - g[0]) * (L.pos[0] - g[0]) + (L.pos[1] - g[1]) * (L.pos[1] - g[1]) + (L.pos[2] - g[2]) * (L.pos[2] - g[2]));
                if (std::abs(anchor_of[kv.first] - t) > 40) continue;
                (L.has_measured_depth ? ed : en).push_back(e2);
                if (ba.selected_landmark_ids_.count(kv.first)) es.push_back(e2);
            }
            {
                const EigenPose D = est[t] * origin_veh[t];  // estimated vehicle <- true vehicle
                const double ang = std::acos(std::max(-1., std::min(1., (D.R[0] + D.R[4] + D.R[8] - 1.) / 2.)));
                std::printf("      true vehicle origin in the estimated vehicle frame (fwd, left, up): %.4f %.4f %.4f m, rotation error %.5f rad\n", D.t[0], D.t[1], D.t[2], ang);
            }
            {   // reprojection rms of this frame's measurements of the SELECTED landmarks (estimated positions) under the
                // ground-truth pose and under the estimated pose
                double s_gt = 0., s_est = 0.;
                int n = 0;
                const EigenPose Tg = origin_veh[t].inverse(), Te = est[t];
                for (const auto& tr : ts.tracks) {
                    if (!ba.selected_landmark_ids_.count(tr.id) || !ba.landmarks_.count(tr.id)) continue;
                    const auto& L = *ba.landmarks_.at(tr.id);
                    const Vector3d P(L.pos[0], L.pos[1], L.pos[2]);
                    const Vector3d a = cam_veh * (Tg * P), b = cam_veh * (Te * P);
                    const double ug = f * a[0] / a[2] + cx, vg = f * a[1] / a[2] + cy, ue = f * b[0] / b[2] + cx, ve = f * b[1] / b[2] + cy;
                    const double mu = tr.feature_points[0].u, mv = tr.feature_points[0].v;
                    s_gt += (ug - mu) * (ug - mu) + (vg - mv) * (vg - mv);
                    s_est += (ue - mu) * (ue - mu) + (ve - mv) * (ve - mv);
                    ++n;
                }
                std::printf("      reprojection rms over %d selected tracks: GT pose %.2f px, estimated pose %.2f px\n", n, std::sqrt(s_gt / std::max(1, n)), std::sqrt(s_est / std::max(1, n)));
            }
            auto med = [](std::vector<double>& v) { if (v.empty()) return -1.; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            std::printf("      landmark error medians: selected %.3f (n %zu), depth %.3f (n %zu), no depth %.3f (n %zu)\n", med(es), es.size(), med(ed), ed.size(), med(en), en.size());
        }
        if (t >= 1) last_motion = est[t] * est[t - 1].inverse();
    }
    // absolute trajectory error of the dumped poses (vehicle positions in the origin frame; first pose is fixed = GT)
    double se = 0., worst = 0.;
    for (int t = 0; t < n_frames; ++t) {
        const Vector3d e = est[t].inverse().translation() - origin_veh[t].translation();
        se += e.norm() * e.norm();
        worst = std::max(worst, e.norm());
    }
    const double ate = std::sqrt(se / n_frames);
    const double path = 0.55 * (n_frames - 1);
    int n_kf_total = 0;
    for (char c : is_kf) n_kf_total += c;
    CHECK(n_kf_total >= n_frames / 3 && n_kf_total <= (2 * n_frames) / 3 + 1);  // about every second frame
    std::printf("stream: %d frames, %d keyframes, %d landmarks, window %d: ATE rmse %.4f m (max %.4f m) over %.1f m; %d solves, %.1f ms per solve()\n",
                n_frames, n_kf_total, n_lm, window, ate, worst, path, n_solves, n_solves ? 1e3 * t_solve / n_solves : 0.);
    std::printf("stream: back end (keyframe construction, adjustPoseOnly, selection, push, window cut, solve) %.2f ms per frame -> %.1f frames/s\n",
                1e3 * t_ba / n_frames, n_frames / t_ba);
    if (!long_run) {
        CHECK(ate < 0.05);
        CHECK(worst < 0.12);
    }
    CHECK((int)ba.keyframes_.size() == n_kf_total);
    std::printf("%d checks, %d failed\n", g_checks, g_fail);
    return g_fail ? 1 : 0;
}
