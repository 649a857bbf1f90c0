// slam_run -- BASELINE.json configs[3]: a synthetic KITTI-00-shaped sequence through the pgslam facade
// (pgslam::PoseGraphSlam<float>: scan-to-local-map ICP on the GPU, keyframe graph, loop closing, pose-graph solve on
// the host), as a C++ pgslam user would run it: host clouds in, poses out.  Driver of `bench.py --workload slam` and of
// tests/test_slam_replay.py; not part of the library.
//
//   slam_run SEQUENCE [--record N FILE] [--limit S] [--mt] [--filters identity|sensor]
//
// --filters sensor: the input filters a range sensor's driver would configure (Localizer.hpp:77,103): RemoveNaN, a range cut
// just inside the sensor's reach, the box of the vehicle itself -- they drop next to nothing of the synthetic scans, but every
// scan goes through the localizer's input stage at full size (one device pass: pgicp_filter_cloud).
//
// --mt: the multi-thread flavour (pgslam::PoseGraphSlamMT): scans are queued as fast as they are read, the three workers
// run freely (the loop closer drains its queue into device batches); poses are compared at the end only.
//
// SEQUENCE (written by bench.py / the test from pgslam_amd/synth.py): int32 magic 'PGSQ', n_scans, n_pts; per scan
// 16 doubles T_world_robot (truth, row-major), 16 doubles odometry pose, n_pts*3 floats xyz (robot frame), n_pts*3
// floats normals.  --record: every ICP call whose ordinal is a multiple of (calls so far / N) -- in practice an even
// sample of N scan-to-map calls plus every loop-closure ICP up to N -- is written to FILE for replay through the CPU
// oracle: reading, reference, initial guess, result, iterations.  Prints one JSON object on stdout.
#include <pgslam_amd/slam.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using T = float;
IMPORT_PGSLAM_TYPES(T)

static const char *kIcpYaml =
    "matcher:\n  KDTreeMatcher:\n    knn: 1\n    epsilon: 0\n    maxDist: 2.0\n"
    "outlierFilters:\n  - TrimmedDistOutlierFilter:\n      ratio: 0.85\n"
    "errorMinimizer:\n  PointToPlaneWithCovErrorMinimizer:\n    sensorStdDev: 0.01\n"
    "transformationCheckers:\n  - CounterTransformationChecker:\n      maxIterationCount: 30\n"
    "  - DifferentialTransformationChecker:\n      minDiffRotErr: 0.001\n      minDiffTransErr: 0.01\n      smoothLength: 3\n"
    "inspector:\n  NullInspector\nlogger:\n  NullLogger\n";

static const char *kSensorFilters =
    "- RemoveNaNDataPointsFilter\n"
    "- MaxDistDataPointsFilter:\n    maxDist: 79.9\n"
    "- BoundingBoxDataPointsFilter:\n    xMin: -1.2\n    xMax: 1.2\n    yMin: -0.9\n    yMax: 0.9\n    zMin: -2.0\n    zMax: 0.5\n    removeInside: 1\n";
static const char *g_filters = "- IdentityDataPointsFilter\n";

static Matrix from_rows(const double *r)
{
    Matrix m(4, 4);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m(i, j) = (T)r[4 * i + j];
    return m;
}

struct Recorder {
    FILE *f = nullptr;
    int want = 0, written = 0, written_kind[2] = {0, 0}, scan = 0, every = 1;
    long long calls = 0;
    void open(const char *path, int n, int expected_calls)
    {
        f = std::fopen(path, "wb");
        want = n;
        every = std::max(1, expected_calls / std::max(1, n));
        const int head[4] = {0x50524750 /* 'PGRP' */, 1, 0, 0};
        if (f) std::fwrite(head, sizeof head, 1, f);
    }
    // kind 0: scan-to-local-map ICP (every `every`-th call, at most `want`); kind 1: loop-closure ICP (the first `want`)
    void operator()(int k, const DP &reading, const DP &reference, const Matrix &Ti, const Matrix &To, const pgicp_stats &st)
    {
        if (!f || written_kind[k] >= want) return;
        const int n = (int)reading.getNbPoints(), m = (int)reference.getNbPoints();
        const int head[8] = {k, scan, n, m, st.iterations, st.converged, st.status, st.max_iter_reached};
        std::fwrite(head, sizeof head, 1, f);
        double t[32];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { t[4 * i + j] = (double)Ti(i, j); t[16 + 4 * i + j] = (double)To(i, j); }
        std::fwrite(t, sizeof t, 1, f);
        const double ov = st.overlap;
        std::fwrite(&ov, sizeof ov, 1, f);
        std::vector<float> buf;
        auto dump = [&](const T *p, int stride, int cnt) {
            buf.resize((size_t)cnt * 3);
            for (int i = 0; i < cnt; i++) for (int a = 0; a < 3; a++) buf[(size_t)3 * i + a] = (float)p[(size_t)i * stride + a];
            std::fwrite(buf.data(), sizeof(float), buf.size(), f);
        };
        dump(reading.xyzPtr(), reading.xyzStride(), n);
        dump(reference.xyzPtr(), reference.xyzStride(), m);
        dump(reference.normalsPtr(), reference.normalsStride(), m);
        written++;
        written_kind[k]++;
    }
    //! asked before the observer's arguments are made (ICPChainBase::onAlignWanted): a local map that lives in device memory
    //! is downloaded only for the calls that are written
    bool wanted(int k)
    {
        if (!f || written_kind[k] >= want) return false;
        return k != 0 || (calls++ % every) == 0;
    }
    void close()
    {
        if (!f) return;
        std::fseek(f, 8, SEEK_SET);
        std::fwrite(&written, sizeof written, 1, f);
        std::fclose(f);
    }
};

// How many keyframes lie within `thr` metres of a keyframe made at least `gap` keyframes earlier (a revisit the loop
// closer's geometric test, LoopCloser.hpp:17, can see), by the given positions.
template <typename P>
static int count_revisits(size_t n, P pos, double thr, size_t gap)
{
    int c = 0;
    for (size_t v = gap; v < n; v++) {
        bool hit = false;
        for (size_t u = 0; u + gap <= v && !hit; u++) {
            const double dx = pos(v, 0) - pos(u, 0), dy = pos(v, 1) - pos(u, 1), dz = pos(v, 2) - pos(u, 2);
            hit = dx * dx + dy * dy + dz * dz <= thr * thr;
        }
        c += hit ? 1 : 0;
    }
    return c;
}

// the multi-thread flavour: queue everything, wait for the workers, compare the keyframes with the truth
static int run_mt(FILE *f, int S, int N, bool report, std::vector<double> &walls)
{
    pgslam::PoseGraphSlamMT<T> slam;
    slam.SetIcpConfigFromStrings(g_filters, kIcpYaml, kIcpYaml);
    slam.Run();
    std::vector<double> Tt(16), To(16);
    std::vector<float> xyz((size_t)N * 3), nrm((size_t)N * 3);
    std::vector<Matrix> truth;
    const Matrix I4 = Matrix::Identity(4, 4);
    const auto t_begin = std::chrono::steady_clock::now();
    for (int s = 0; s < S; s++) {
        if (std::fread(Tt.data(), sizeof(double), 16, f) != 16 || std::fread(To.data(), sizeof(double), 16, f) != 16 ||
            std::fread(xyz.data(), sizeof(float), xyz.size(), f) != xyz.size() || std::fread(nrm.data(), sizeof(float), nrm.size(), f) != nrm.size()) {
            std::fprintf(stderr, "sequence file truncated at scan %d\n", s);
            return 2;
        }
        truth.push_back(from_rows(Tt.data()));
        slam.AddData((unsigned long long)s, "world", from_rows(To.data()), I4, std::make_shared<DP>(DP::fromXYZ(xyz.data(), N, nrm.data())));
        // a bounded input queue, as a sensor driver would keep: at most 8 scans ahead of the localizer
        while ((size_t)s > slam.localizer().processed() + 8) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    slam.WaitIdle();
    const double this_wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    std::fclose(f);
    walls.push_back(this_wall);
    if (!report) return 0;
    // the figure reported: this pass alone, or -- with --passes P > 1 -- the median of the passes after the first
    double wall = this_wall;
    std::string passes_json;
    if (walls.size() > 1) {
        std::vector<double> timed(walls.begin() + 1, walls.end());
        std::sort(timed.begin(), timed.end());
        wall = timed[timed.size() / 2];
        passes_json = ", \"passes\": " + std::to_string(walls.size()) + ", \"pass_wall_s\": [";
        for (size_t k = 0; k < walls.size(); k++) { char b[64]; std::snprintf(b, sizeof b, "%s%.6f", k ? ", " : "", walls[k]); passes_json += b; }
        passes_json += "], \"wall_s_is\": \"median of the passes after the first (the process's warm-up)\"";
    }
    auto lock = slam.map_manager().GetGraphLock();
    auto &g = slam.map_manager().GetGraph();
    int loops = 0;
    for (size_t e = 0; e < g.NumEdges(); e++) loops += g.Edge(e).c.type == Constraint::kLoopConstraint;
    const Matrix d = truth.back().inverse() * slam.localizer().T_world_robot();
    const double e_last = std::sqrt((double)(d(0, 3) * d(0, 3) + d(1, 3) * d(1, 3) + d(2, 3) * d(2, 3)));
    std::printf("{\"mode\": \"mt\", \"scans\": %d, \"points_per_scan\": %d, \"wall_s\": %.6f, \"slam_s\": %.6f, \"scans_per_s\": %.3f, "
                "\"keyframes\": %zu, \"loop_edges\": %d, \"loop_candidates_tried\": %d, \"loops_closed\": %d, \"loop_batches\": %d, "
                "\"largest_loop_batch\": %d, \"optimizer_runs\": %d, \"optimizer_iterations\": %d, \"optimizer_host_s\": %.6f, "
                "\"map_rebuilds\": %d, \"tracking_error_last_m\": %.5f, \"keyframes_revisiting_within_3m_by_estimate\": %d, "
                "\"clouds_uploaded_one_scan_ahead\": %zu, \"loop_batches_on_device\": %zu, \"loop_candidates_assembled_on_device\": %zu, "
                "\"keyframes_resident\": %zu, \"keyframe_uploads\": %zu, \"keyframe_evictions\": %zu, "
                "\"input_stage_thread_s\": %.4f, \"localizer_thread_s\": {\"waiting_for_input_stage\": %.4f, \"icp\": %.4f, \"after_icp\": %.4f, \"overlap_probe\": %.4f, \"rebuilds\": %.4f, \"new_keyframes\": %.4f}%s}\n",
                S, N, wall, wall, (S - 1) / wall, g.NumVertices(), loops, slam.loop_closer().candidates_tried(), slam.loop_closer().loops_closed(),
                slam.loop_closer().batches(), slam.loop_closer().largest_batch(), slam.optimizer().runs(), slam.optimizer().total_iterations(),
                slam.optimizer().total_seconds(), slam.localizer().rebuilds(), e_last,
                count_revisits(g.NumVertices(), [&](size_t v, int a) { return (double)g[v].optimized_T_world_kf(a, 3); }, 3.0, 4),
                slam.localizer().prefetches(), slam.loop_closer().device_batches(), slam.loop_closer().device_candidates(),
                slam.map_manager().resident_keyframes(), slam.map_manager().device_uploads(), slam.map_manager().device_evictions(),
                slam.localizer().input_stage_seconds(), slam.localizer().waited_for_input_stage(), slam.localizer().phase_seconds()[1], slam.localizer().phase_seconds()[2],
                slam.localizer().after_icp_seconds()[1], slam.localizer().after_icp_seconds()[2], slam.localizer().after_icp_seconds()[3], passes_json.c_str());
    return 0;
}

struct StResult {
    double slam_s = 0, icp_first_s = 0, icp_p50_s = 0, icp_p99_s = 0, icp_max_s = 0;
    int icp_max_scan = 0;
    long long alloc[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // pgicp_debug_alloc_stats of this pass: calls / ns of hipMalloc, hipFree, hipHostMalloc, hipHostFree
    std::string slowest;                                // JSON list: the five slowest ICP calls (scan, seconds, iterations, map rebuilt by the scan before)
    std::string json;
};

// one pass of the single-thread flavour over the sequence: a fresh facade (graph, chains, contexts), every scan through AddData
static int run_st(const char *seq_path, int limit, const char *rec_path, int rec_n, StResult &out)
{
    FILE *f = std::fopen(seq_path, "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", seq_path); return 2; }
    int head[3];
    if (std::fread(head, sizeof head, 1, f) != 1 || head[0] != 0x51534750 /* 'PGSQ' */) { std::fprintf(stderr, "bad sequence file\n"); return 2; }
    const int S = std::min(head[1], limit), N = head[2];
    pgslam::PoseGraphSlam<T> slam;
    slam.SetIcpConfigFromStrings(g_filters, kIcpYaml, kIcpYaml);
    Recorder rec;
    if (rec_path) {
        rec.open(rec_path, rec_n, S);
        slam.localizer().icp().onAlign = [&](const DP &r, const DP &m, const Matrix &a, const Matrix &b, const pgicp_stats &s) { rec(0, r, m, a, b, s); };
        slam.loop_closer().icp().onAlign = [&](const DP &r, const DP &m, const Matrix &a, const Matrix &b, const pgicp_stats &s) { rec(1, r, m, a, b, s); };
        slam.localizer().icp().onAlignWanted = [&]() { return rec.wanted(0); };
        slam.loop_closer().icp().onAlignWanted = [&]() { return rec.wanted(1); };
    }
    std::vector<double> Tt(16), To(16);
    std::vector<float> xyz((size_t)N * 3), nrm((size_t)N * 3);
    std::vector<Matrix> truth;
    std::vector<size_t> kf_scan;                    // scan every keyframe (graph vertex) was made from
    Matrix last_odom = Matrix::Identity(4, 4);
    std::vector<double> err_track;
    double t_icp_loop = 0.0, t_io = 0.0;
    double icp_s_before = 0.0;
    std::vector<int> icp_scan_iters, icp_scan_after_rebuild;
    int rebuilds_before = 0;
    bool rebuilt_last_scan = false;
    long long alloc0[8];
    (void)pgicp_debug_alloc_stats(alloc0);
    std::vector<double> icp_scan_s;                 // the ICP call of every scan (localizer phase 1), to tell a cold start from a slow box
    long long icp_iterations = 0;
    int not_converged = 0;
    const Matrix I4 = Matrix::Identity(4, 4);
    unsigned last_cloud_points = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    for (int s = 0; s < S; s++) {
        const auto t0 = std::chrono::steady_clock::now();
        if (std::fread(Tt.data(), sizeof(double), 16, f) != 16 || std::fread(To.data(), sizeof(double), 16, f) != 16 ||
            std::fread(xyz.data(), sizeof(float), xyz.size(), f) != xyz.size() || std::fread(nrm.data(), sizeof(float), nrm.size(), f) != nrm.size()) {
            std::fprintf(stderr, "sequence file truncated at scan %d\n", s);
            return 2;
        }
        auto cloud = std::make_shared<DP>(DP::fromXYZ(xyz.data(), N, nrm.data()));
        const auto t1 = std::chrono::steady_clock::now();
        t_io += std::chrono::duration<double>(t1 - t0).count();
        rec.scan = s;
        slam.AddData((unsigned long long)s, "world", from_rows(To.data()), I4, cloud);
        t_icp_loop += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        {
            const double so_far = slam.localizer().phase_seconds()[1];
            icp_scan_s.push_back(so_far - icp_s_before);
            icp_s_before = so_far;
            icp_scan_iters.push_back(s > 0 ? slam.localizer().icp().lastStats.iterations : 0);
            icp_scan_after_rebuild.push_back(rebuilt_last_scan ? 1 : 0);
            rebuilt_last_scan = slam.localizer().rebuilds() != rebuilds_before;
            rebuilds_before = slam.localizer().rebuilds();
        }
        last_cloud_points = cloud->getNbPoints();
        truth.push_back(from_rows(Tt.data()));
        last_odom = from_rows(To.data());
        while (kf_scan.size() < slam.map_manager().GetGraph().NumVertices()) kf_scan.push_back((size_t)s);
        if (s > 0) {
            const auto &st = slam.localizer().icp().lastStats;
            icp_iterations += st.iterations;
            not_converged += st.converged ? 0 : 1;
        }
        // tracking error: the live pose against the truth, in the frame of the first pose (odometry starts at the truth)
        const Matrix d = truth[s].inverse() * slam.localizer().T_world_robot();
        err_track.push_back(std::sqrt((double)(d(0, 3) * d(0, 3) + d(1, 3) * d(1, 3) + d(2, 3) * d(2, 3))));
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    std::fclose(f);
    rec.close();
    auto &g = slam.map_manager().GetGraph();
    int loops = 0;
    for (size_t e = 0; e < g.NumEdges(); e++) loops += g.Edge(e).c.type == Constraint::kLoopConstraint;
    // keyframe poses after the last optimisation against the truth of the scans they were made from
    double kf_max = 0, kf_sum2 = 0;
    for (size_t v = 0; v < g.NumVertices() && v < kf_scan.size(); v++) {
        const Matrix d = truth[kf_scan[v]].inverse() * g[v].optimized_T_world_kf;
        const double e = std::sqrt((double)(d(0, 3) * d(0, 3) + d(1, 3) * d(1, 3) + d(2, 3) * d(2, 3)));
        kf_max = std::max(kf_max, e);
        kf_sum2 += e * e;
    }
    double e_max = 0, e_sum2 = 0, e_last = err_track.empty() ? 0 : err_track.back();
    for (double e : err_track) { e_max = std::max(e_max, e); e_sum2 += e * e; }
    // odometry alone, for comparison: where the last odometry pose ends up against the truth
    const Matrix d_odo = truth.back().inverse() * last_odom;
    const double odo_last = std::sqrt((double)(d_odo(0, 3) * d_odo(0, 3) + d_odo(1, 3) * d_odo(1, 3) + d_odo(2, 3) * d_odo(2, 3)));
    // the fast matcher over every context of the process (all zero unless PGICP_PROFILE_ALL=1 made the contexts profile)
    long long kp_l = 0, kp_u = 0, kp_p = 0, kp_m = 0;
    double kp_ms = 0;
    (void)pgicp_profile_process(PGICP_PROF_KNN_GRID, &kp_l, &kp_ms, &kp_u, &kp_p, &kp_m);
    {
        long long a1[8];
        (void)pgicp_debug_alloc_stats(a1);
        for (int k = 0; k < 8; k++) out.alloc[k] = a1[k] - alloc0[k];
        std::vector<int> order(icp_scan_s.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return icp_scan_s[(size_t)a] > icp_scan_s[(size_t)b]; });
        out.slowest = "[";
        for (size_t k = 0; k < std::min<size_t>(5, order.size()); k++) {
            char b[160];
            const size_t i = (size_t)order[k];
            std::snprintf(b, sizeof b, "%s{\"scan\": %zu, \"s\": %.6f, \"iterations\": %d, \"map_rebuilt_by_previous_scan\": %d}", k ? ", " : "", i, icp_scan_s[i],
                          icp_scan_iters[i], icp_scan_after_rebuild[i]);
            out.slowest += b;
        }
        out.slowest += "]";
    }
    std::vector<double> sorted_icp(icp_scan_s.begin() + std::min<size_t>(1, icp_scan_s.size()), icp_scan_s.end());   // (scan 0 makes the first keyframe: no ICP)
    if (sorted_icp.empty()) sorted_icp.push_back(0.0);
    out.icp_first_s = sorted_icp[0];
    out.icp_max_scan = 1 + (int)(std::max_element(sorted_icp.begin(), sorted_icp.end()) - sorted_icp.begin());
    std::sort(sorted_icp.begin(), sorted_icp.end());
    out.icp_p50_s = sorted_icp[sorted_icp.size() / 2];
    out.icp_p99_s = sorted_icp[std::min(sorted_icp.size() - 1, sorted_icp.size() * 99 / 100)];
    out.icp_max_s = sorted_icp.back();
    out.slam_s = t_icp_loop;
    char buf[4096];
    std::snprintf(buf, sizeof buf, "{\"scans\": %d, \"points_per_scan\": %d, \"wall_s\": %.6f, \"slam_s\": %.6f, \"io_s\": %.6f, \"scans_per_s\": %.3f, "
                "\"keyframes\": %zu, \"loop_edges\": %d, \"loop_candidates_tried\": %d, \"loops_closed\": %d, "
                "\"optimizer_runs\": %d, \"optimizer_iterations\": %d, \"optimizer_host_s\": %.6f, \"map_rebuilds\": %d, "
                "\"mean_icp_iterations\": %.3f, \"scans_not_converged\": %d, \"tracking_error_rms_m\": %.5f, "
                "\"tracking_error_max_m\": %.5f, \"tracking_error_last_m\": %.5f, \"odometry_error_last_m\": %.5f, "
                "\"keyframe_error_rms_m\": %.5f, \"keyframe_error_max_m\": %.5f, \"recorded_calls\": %d, \"recorded_loop_calls\": %d, "
                "\"localizer_host_s\": {\"filters_and_sensor_transform\": %.4f, \"icp\": %.4f, \"after_icp\": %.4f, \"after_icp_parts\": {\"neighbour_search\": %.4f, \"overlap_probe\": %.4f, \"rebuilds\": %.4f, \"new_keyframes_incl_loop_closing\": %.4f}}, "
                "\"input_filters\": \"%s\", \"device_input_stages\": %zu, \"device_readings_used\": %zu, \"device_map_rebuilds\": %zu, \"overlap_probes_seeded\": %zu, \"points_after_filters_last_scan\": %u, "
                "\"loop_candidates_assembled_on_device\": %zu, \"keyframes_resident\": %zu, \"keyframe_uploads\": %zu, \"keyframe_evictions\": %zu, "
                "\"keyframes_revisiting_within_3m_by_truth\": %d, \"keyframes_revisiting_within_3m_by_estimate\": %d, "
                "\"knn_profile\": {\"launches\": %lld, \"total_ms\": %.4f, \"reading_points\": %lld, \"problems\": %lld, \"map_points\": %lld}",
                S, N, wall, t_icp_loop, t_io, (S - 1) / t_icp_loop, g.NumVertices(), loops, slam.loop_closer().candidates_tried(),
                slam.loop_closer().loops_closed(), slam.optimizer().runs(), slam.optimizer().total_iterations(), slam.optimizer().total_seconds(),
                slam.localizer().rebuilds(), S > 1 ? (double)icp_iterations / (S - 1) : 0.0, not_converged,
                std::sqrt(e_sum2 / std::max<size_t>(1, err_track.size())), e_max, e_last, odo_last,
                std::sqrt(kf_sum2 / std::max<size_t>(1, kf_scan.size())), kf_max, rec.written, rec.written_kind[1],
                slam.localizer().phase_seconds()[0], slam.localizer().phase_seconds()[1], slam.localizer().phase_seconds()[2],
                slam.localizer().after_icp_seconds()[0], slam.localizer().after_icp_seconds()[1], slam.localizer().after_icp_seconds()[2], slam.localizer().after_icp_seconds()[3],
                g_filters == kSensorFilters ? "RemoveNaN, MaxDist 79.9, BoundingBox (vehicle)" : "Identity", slam.localizer().device_input_stages(),
                slam.localizer().device_readings_used(), slam.localizer().device_rebuilds(), slam.localizer().probes_seeded(), last_cloud_points,
                slam.loop_closer().device_candidates(), slam.map_manager().resident_keyframes(), slam.map_manager().device_uploads(), slam.map_manager().device_evictions(),
                count_revisits(std::min(g.NumVertices(), kf_scan.size()), [&](size_t v, int a) { return (double)truth[kf_scan[v]](a, 3); }, 3.0, 4),
                count_revisits(g.NumVertices(), [&](size_t v, int a) { return (double)g[v].optimized_T_world_kf(a, 3); }, 3.0, 4),
                kp_l, kp_ms, kp_u, kp_p, kp_m);
    out.json = buf;
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: slam_run SEQUENCE [--record N FILE] [--limit S] [--mt] [--filters identity|sensor] [--passes P]\n"); return 2; }
    const char *rec_path = nullptr;
    int rec_n = 0, limit = 1 << 30, passes = 1;
    bool mt = false;
    for (int a = 2; a < argc; a++) {
        if (!std::strcmp(argv[a], "--record") && a + 2 < argc) { rec_n = std::atoi(argv[a + 1]); rec_path = argv[a + 2]; a += 2; }
        else if (!std::strcmp(argv[a], "--limit") && a + 1 < argc) { limit = std::atoi(argv[a + 1]); a += 1; }
        else if (!std::strcmp(argv[a], "--passes") && a + 1 < argc) { passes = std::max(1, std::atoi(argv[a + 1])); a += 1; }
        else if (!std::strcmp(argv[a], "--mt")) mt = true;
        else if (!std::strcmp(argv[a], "--filters") && a + 1 < argc) { g_filters = !std::strcmp(argv[a + 1], "sensor") ? kSensorFilters : "- IdentityDataPointsFilter\n"; a += 1; }
    }
    if (mt) {
        // --passes P > 1, as for the single-thread flavour: the first pass warms the process up (reported, not counted), every
        // pass runs a fresh facade with fresh worker threads and contexts; the statistics printed are the last pass's, its
        // wall and rate those of the median timed pass
        std::vector<double> walls;
        for (int p = 0; p < passes; p++) {
            FILE *f = std::fopen(argv[1], "rb");
            if (!f) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
            int head[3];
            if (std::fread(head, sizeof head, 1, f) != 1 || head[0] != 0x51534750 /* 'PGSQ' */) { std::fprintf(stderr, "bad sequence file\n"); return 2; }
            const int rc = run_mt(f, std::min(head[1], limit), head[2], p + 1 == passes, walls);
            if (rc) return rc;
        }
        return 0;
    }
    // --passes P > 1: the FIRST pass is the process's warm-up (code-object load, the first allocations of every pool, the page
    // cache of the sequence file) and is reported but not counted; the P - 1 that follow are timed, each with a fresh facade,
    // and the line carries every pass and the median of the timed ones (`slam_s` = that median, so a caller dividing scans by
    // `slam_s` gets the median rate).  --record applies to the last pass only.
    std::vector<StResult> res((size_t)passes);
    for (int p = 0; p < passes; p++) {
        const bool last = p + 1 == passes;
        const int rc = run_st(argv[1], limit, last ? rec_path : nullptr, rec_n, res[(size_t)p]);
        if (rc) return rc;
    }
    std::vector<double> timed;
    for (int p = passes > 1 ? 1 : 0; p < passes; p++) timed.push_back(res[(size_t)p].slam_s);
    std::sort(timed.begin(), timed.end());
    const double median = timed[timed.size() / 2];
    std::string passes_json = "[";
    for (int p = 0; p < passes; p++) { char b[64]; std::snprintf(b, sizeof b, "%s%.6f", p ? ", " : "", res[(size_t)p].slam_s); passes_json += b; }
    passes_json += "]";
    const StResult &r = res.back();
    auto alloc_json = [](const StResult &q) {
        char b[256];
        std::snprintf(b, sizeof b, "{\"hipMalloc\": [%lld, %.4f], \"hipFree\": [%lld, %.4f], \"hipHostMalloc\": [%lld, %.4f], \"hipHostFree\": [%lld, %.4f]}",
                      q.alloc[0], q.alloc[1] * 1e-9, q.alloc[2], q.alloc[3] * 1e-9, q.alloc[4], q.alloc[5] * 1e-9, q.alloc[6], q.alloc[7] * 1e-9);
        return std::string(b);
    };
    std::printf("%s, \"passes\": %d, \"pass_slam_s\": %s, \"slam_s_last_pass\": %.6f, \"slam_s_median_timed\": %.6f, "
                "\"icp_call_s\": {\"first\": %.6f, \"p50\": %.6f, \"p99\": %.6f, \"max\": %.6f, \"max_at_scan\": %d, "
                "\"first_of_pass0\": %.6f, \"max_of_pass0\": %.6f}, "
                "\"alloc_calls_and_seconds_last_pass\": %s, \"alloc_calls_and_seconds_pass0\": %s, \"slowest_icp_calls_last_pass\": %s, "
                "\"slowest_icp_calls_pass0\": %s}\n",
                r.json.c_str(), passes, passes_json.c_str(), r.slam_s, median, r.icp_first_s, r.icp_p50_s, r.icp_p99_s, r.icp_max_s,
                r.icp_max_scan, res[0].icp_first_s, res[0].icp_max_s, alloc_json(r).c_str(), alloc_json(res[0]).c_str(), r.slowest.c_str(),
                res[0].slowest.c_str());
    return 0;
}
