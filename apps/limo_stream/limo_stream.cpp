// limo_stream — ROS-free streaming visual odometry on a synthetic KITTI-00-shaped drive (BASELINE.json configs[4]):
//   per frame: 64-beam sweep + tracked features -> StreamDriver (limo_depth_estimate -> adjustPoseOnly -> keyframe
//   selection -> push -> deactivateKeyframes -> solve) -> KITTI pose row.
// The replacement of the reference's demo application (demo_keyframe_bundle_adjustment_meta/apps/main_program/
// main_program.cpp:39-216 reads KITTI data from disk and plays it through the ROS nodes; there is no dataset here, so the
// drive is synthesised by synth_world.hpp) and of the node's callback (limo_amd/kba/stream_driver.hpp).
//
//   limo_stream [--frames N] [--features N] [--az N] [--seed S] [--window K] [--poses out.txt] [--gt-poses gt.txt]
//               [--dump-velodyne DIR] [--velodyne DIR] [--no-depth] [--depth-ahead thread|stream|none] [--quiet]
// --dump-velodyne writes every synthetic sweep as a KITTI velodyne scan (DIR/NNNNNN.bin); --velodyne replays scans from
// such a directory instead of ray-casting them (the scans of a real KITTI sequence have the same format; the tracked
// features of a real sequence come from the feature tracker, which is outside this path).
// The sweeps are received into page-locked buffers (limo_host_alloc) and the input runs one frame ahead of the pipeline: the driver
// assigns the depths of frame t+1 while the pose refinement and the solve of frame t run (StreamDriver::announceNextFrame) - the
// reference's depth estimator is a process of its own beside the BA node.  --depth-ahead thread (default): a thread of the driver does
// it; stream: the calling thread starts it on the driver's own HIP stream and collects it at the next frame (limo_depth_estimate_begin
// / _end); none (= --no-prefetch): every frame's depths are assigned inside its own process() call.  Same pose rows, bit for bit.
// Prints one summary line per run and `key value` lines for scripts: fps of the pipeline (input synthesis excluded and
// reported separately), ATE against the ground truth, share of features that received a LiDAR depth.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>

#include "../../limo_amd/kba/kitti_io.hpp"
#include "../../limo_amd/kba/stream_driver.hpp"
#include "synth_world.hpp"

using namespace keyframe_bundle_adjustment;

int main(int argc, char** argv) {
    int n_frames = 200, n_feat = 1500, n_az = 2000, window = 5, misannounce_every = 0;
    uint64_t seed = 7;
    std::string poses_path, gt_path, dump_dir, replay_dir;
    bool use_depth = true, quiet = false, five_point_prior = false;
    StreamParams::DepthAhead depth_ahead = StreamParams::DepthAhead::Thread;
    double min_flow = -1., time_between_keyframes = -1.;
    for (int i = 1; i < argc; ++i) {
        auto arg = [&](const char* name) { return !std::strcmp(argv[i], name) && i + 1 < argc; };
        if (arg("--frames")) n_frames = std::atoi(argv[++i]);
        else if (arg("--features")) n_feat = std::atoi(argv[++i]);
        else if (arg("--az")) n_az = std::atoi(argv[++i]);
        else if (arg("--seed")) seed = std::strtoull(argv[++i], nullptr, 10);
        else if (arg("--window")) window = std::atoi(argv[++i]);
        else if (arg("--misannounce-every")) misannounce_every = std::atoi(argv[++i]);  // (testing aid, see the frame loop)
        else if (arg("--min-flow")) min_flow = std::atof(argv[++i]);
        else if (arg("--time-between-keyframes")) time_between_keyframes = std::atof(argv[++i]);
        else if (arg("--poses")) poses_path = argv[++i];
        else if (arg("--gt-poses")) gt_path = argv[++i];
        else if (arg("--dump-velodyne")) dump_dir = argv[++i];
        else if (arg("--velodyne")) replay_dir = argv[++i];
        else if (!std::strcmp(argv[i], "--no-depth")) use_depth = false;
        else if (!std::strcmp(argv[i], "--no-prefetch")) depth_ahead = StreamParams::DepthAhead::None;
        else if (arg("--depth-ahead")) {
            const std::string m = argv[++i];
            if (m != "thread" && m != "stream" && m != "none") {
                std::fprintf(stderr, "limo_stream: --depth-ahead thread|stream|none\n");
                return 2;
            }
            depth_ahead = m == "thread" ? StreamParams::DepthAhead::Thread : m == "stream" ? StreamParams::DepthAhead::Stream : StreamParams::DepthAhead::None;
        }
        else if (!std::strcmp(argv[i], "--quiet")) quiet = true;
        else if (!std::strcmp(argv[i], "--five-point-prior")) five_point_prior = true;  // the node's prior without tf (mono_lidar.cpp:157-186)
        else {
            std::fprintf(stderr, "usage: limo_stream [--frames N] [--features N] [--az N] [--seed S] [--window K] [--min-flow px] [--time-between-keyframes sec] [--poses file] [--gt-poses file] [--dump-velodyne dir] [--velodyne dir] [--no-depth] [--depth-ahead thread|stream|none] [--five-point-prior] [--quiet]\n");
            return 2;
        }
    }
    using clk = std::chrono::steady_clock;
    synth_world::World world(n_frames, seed);
    world.n_az = n_az;
    Camera::Ptr cam = std::make_shared<Camera>(world.f, Vector2d(world.cx, world.cy), world.cam_veh);
    StreamParams sp;
    sp.max_size_optimization_window = window;
    sp.assign_depth = use_depth;
    sp.depth_ahead = depth_ahead;
    if (min_flow >= 0.) sp.min_median_flow = min_flow;
    if (time_between_keyframes > 0.) sp.time_between_keyframes_sec = time_between_keyframes;
    if (five_point_prior) sp.motion_prior = StreamParams::MotionPrior::FivePoint;
    sp.solver_time_sec = -1.;  // no wall-clock cap: the drive is reproducible
    sp.image_width = (int)world.W;
    sp.image_height = (int)world.H;
    sp.height_over_ground = synth_world::World::kHeightOverGround;
    StreamDriver driver(sp, cam, world.cam_lidar);

    std::mt19937_64 rng(seed * 31 + 1);
    const int history = 10;
    std::vector<int> live;  // landmarks tracked in the previous frame
    std::map<int, int> first_seen;
    std::vector<float> cloud;
    std::vector<uint8_t> cloud_ground;
    double sec_synth = 0., sec_pipeline = 0.;
    size_t n_points = 0;
    // a frame as the sensor drivers deliver it: the tracker's message and the sweep in a page-locked buffer
    struct Frame {
        Tracklets ts;
        float* scan = nullptr;
        size_t cap = 0, n_pts = 0, n_tracks = 0;
    } frames[2];
    auto release = [&]() {
        driver.cancelPrefetch();
        for (Frame& fr : frames)
            if (fr.scan) limo_host_free(fr.scan);
    };
    auto synthesize = [&](int t, Frame& out) -> bool {
        world.sweep(t, cloud, cloud_ground);  // (also labels the returns that hit the ground: the features' class labels come from it)
        if (!replay_dir.empty()) {
            std::vector<float> scan;
            if (!kitti_io::readVelodyneBin(kitti_io::velodynePath(replay_dir, t), scan) || scan.size() != cloud.size()) {
                std::fprintf(stderr, "limo_stream: cannot replay %s\n", kitti_io::velodynePath(replay_dir, t).c_str());
                return false;
            }
            cloud.swap(scan);
        }
        if (!dump_dir.empty() && !kitti_io::writeVelodyneBin(kitti_io::velodynePath(dump_dir, t), cloud.data(), cloud.size() / 4)) {
            std::fprintf(stderr, "limo_stream: cannot write %s\n", kitti_io::velodynePath(dump_dir, t).c_str());
            return false;
        }
        n_points += cloud.size() / 4;
        if (cloud.size() > out.cap) {
            if (out.scan) limo_host_free(out.scan);
            out.cap = cloud.size() + cloud.size() / 8;
            out.scan = static_cast<float*>(limo_host_alloc(out.cap * sizeof(float)));
            if (!out.scan) {
                std::fprintf(stderr, "limo_stream: limo_host_alloc failed\n");
                return false;
            }
        }
        std::memcpy(out.scan, cloud.data(), cloud.size() * sizeof(float));
        out.n_pts = cloud.size() / 4;
        // tracks that survive into this frame (still visible, not occluded), then new ones up to n_feat
        const Vector3d here = world.origin_veh[t].translation();
        const std::vector<int> near = world.boxes_near(here, 70.);
        std::vector<int> now;
        for (int id : live) {
            double u, v, z;
            if (world.visible(t, world.lms[id], near, u, v, z)) now.push_back(id);
        }
        if ((int)now.size() < n_feat) {
            const size_t before = world.lms.size();
            world.spawn(t, cloud, cloud_ground, n_feat - (int)now.size(), rng);
            for (size_t id = before; id < world.lms.size(); ++id) {
                now.push_back((int)id);
                first_seen[(int)id] = t;
            }
        }
        Tracklets ts;
        for (int k = 0; k < history && t - k >= 0; ++k) ts.stamps.push_back((uint64_t)(t - k) * 50000000ull + 1000ull);
        for (int id : now) {
            Tracklet tr;
            tr.id = id;
            const int age = std::min(history, t - first_seen[id] + 1);
            for (int k = 0; k < age; ++k) {
                double u, v, z;
                if (!world.project(t - k, world.lms[id].p, u, v, z)) break;
                double n3[3];
                synth_world::hash_normals((uint64_t)(t - k) * 1000003ull + (uint64_t)id, n3);
                tr.feature_points.push_back(FeaturePoint((float)(u + 0.3 * n3[0]), (float)(v + 0.3 * n3[1])));  // d < 0: the driver fills it
            }
            if (tr.feature_points.empty()) continue;
            tr.age = tr.feature_points.size();
            tr.label = world.lms[id].ground ? 7 : 11;  // cityscapes road / building
            ts.tracks.push_back(tr);
        }
        live = now;
        out.n_tracks = ts.tracks.size();
        out.ts = std::move(ts);
        return true;
    };
    {
        const auto t0 = clk::now();
        if (n_frames > 0 && !synthesize(0, frames[0])) {
            release();
            return 1;
        }
        sec_synth += std::chrono::duration<double>(clk::now() - t0).count();
    }
    Tracklets stale;
    for (int t = 0; t < n_frames; ++t) {
        Frame& cur = frames[t & 1];
        Frame& next = frames[(t + 1) & 1];
        const auto t0 = clk::now();
        if (t + 1 < n_frames && !synthesize(t + 1, next)) {  // the input runs one frame ahead
            release();
            return 1;
        }
        const auto t1 = clk::now();
        // --misannounce-every N (testing aid): every N-th frame the driver is told that the CURRENT frame comes next - what it prepares
        // then does not belong to the frame it gets, must be dropped, and the pose rows must not change
        if (misannounce_every > 0 && t % misannounce_every == misannounce_every - 1) {
            stale = cur.ts;
            driver.announceNextFrame(stale, cur.scan, cur.n_pts);
        } else if (t + 1 < n_frames)
            driver.announceNextFrame(next.ts, next.scan, next.n_pts);
        driver.process(std::move(cur.ts), cur.scan, cur.n_pts);  // (the tracker's message is handed over: the driver fills in depths)
        driver.waitForDepthAhead();  // (the depth thread's work on frame t+1 belongs to this timed region, not to the synthesis below)
        const auto t2 = clk::now();
        sec_synth += std::chrono::duration<double>(t1 - t0).count();
        sec_pipeline += std::chrono::duration<double>(t2 - t1).count();
        if (!quiet && (t % 100 == 0 || t == n_frames - 1)) {
            const Vector3d e = driver.poses().back().inverse().translation() - world.origin_veh[t].translation();
            std::printf("frame %d: %zu tracks, %zu points, position error %.3f m, %d keyframes, %d solves\n", t, cur.n_tracks, cur.n_pts, e.norm(),
                        driver.stats().keyframes, driver.stats().solves);
            std::fflush(stdout);
        }
    }
    release();
    // absolute trajectory error of the dumped poses (vehicle positions in the origin frame)
    double se = 0., worst = 0.;
    for (int t = 0; t < n_frames; ++t) {
        const Vector3d e = driver.poses()[t].inverse().translation() - world.origin_veh[t].translation();
        se += e.norm() * e.norm();
        worst = std::max(worst, e.norm());
    }
    const double ate = std::sqrt(se / n_frames);
    if (!poses_path.empty()) {
        std::ofstream f(poses_path);
        driver.writeKittiTrajectory(f);
    }
    // devkit-style relative errors on the KITTI pose rows (camera frame)
    std::vector<EigenPose> gt_rows, est_rows;
    {
        const EigenPose cv = cam->getEigenPose();
        for (int t = 0; t < n_frames; ++t) {
            gt_rows.push_back(cv * world.origin_veh[t] * cv.inverse());
            est_rows.push_back(cv * driver.poses()[t].inverse() * cv.inverse());
        }
        if (!gt_path.empty()) {
            std::ofstream f(gt_path);
            for (const auto& p : gt_rows) kitti_io::writePoseRow(f, p);
        }
    }
    const kitti_io::TrajectoryError te = kitti_io::evaluateTrajectory(gt_rows, est_rows);
    const auto& st = driver.stats();
    std::printf("limo_stream: %d frames (%d keyframes, %d solves, window %d), %.0f points and %.0f features per frame, %.1f %% of the features with a LiDAR depth\n",
                n_frames, st.keyframes, st.solves, window, (double)n_points / n_frames, (double)st.features / n_frames,
                100. * st.features_with_depth / std::max(1, st.features));
    std::printf("limo_stream: pipeline %.2f ms per frame -> %.1f frames/s (depth %.2f, pose-only %.2f, push %.2f, solve %.2f ms per frame; %.2f ms per solve()); input synthesis %.1f ms per frame\n",
                1e3 * sec_pipeline / n_frames, n_frames / sec_pipeline, 1e3 * st.sec_depth / n_frames, 1e3 * st.sec_pose_only / n_frames,
                1e3 * st.sec_push / n_frames, 1e3 * st.sec_solve / n_frames, st.solves ? 1e3 * st.sec_solve / st.solves : 0., 1e3 * sec_synth / n_frames);
    std::printf("limo_stream: host side per frame: Keyframe object %.2f, keyframe selection %.2f, window cut + labels %.2f ms; inside the C-ABI: adjustPoseOnly %.2f of %.2f, solve %.2f of %.2f ms per frame\n",
                1e3 * st.sec_keyframe / n_frames, 1e3 * st.sec_select / n_frames, 1e3 * st.sec_window / n_frames, 1e3 * st.sec_abi_pose_only / n_frames,
                1e3 * st.sec_pose_only / n_frames, 1e3 * st.sec_abi_solve / n_frames, 1e3 * st.sec_solve / n_frames);
    std::printf("limo_stream: depth assignment one frame ahead (%s) for %d frames: %.2f ms per frame on the depth thread; per-track depth history %.3f ms per frame\n",
                depth_ahead == StreamParams::DepthAhead::Thread ? "thread" : depth_ahead == StreamParams::DepthAhead::Stream ? "stream" : "none", st.depth_prefetched,
                1e3 * st.sec_depth_thread / n_frames, 1e3 * st.sec_depth_history / n_frames);
    std::printf("limo_stream: ATE rmse %.4f m (max %.4f m) over %.1f m\n", ate, worst, 0.55 * (n_frames - 1));
    if (te.rel_samples)
        std::printf("limo_stream: relative errors over 100..800 m sub-paths (KITTI devkit measure, %d samples): translation %.3f %%, rotation %.5f deg/m\n",
                    te.rel_samples, 100. * te.rel_trans, te.rel_rot * 180. / M_PI);
    std::printf("frames %d\nfps %.3f\nate_rmse %.6f\nate_max %.6f\ndepth_fraction %.4f\nkeyframes %d\nsolves %d\nsolves_on_non_keyframes %d\ndepth_prefetched %d\n", n_frames,
                n_frames / sec_pipeline, ate, worst, (double)st.features_with_depth / std::max(1, st.features), st.keyframes, st.solves, st.solves_on_non_keyframes,
                st.depth_prefetched);
    return 0;
}
