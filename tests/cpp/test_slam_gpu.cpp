// PoseGraphSlam facade end to end on the GPU: a robot drives a closed loop through the room-corner scene;
// every scan becomes a keyframe (overlap threshold above what the trimmed filter can report), the loop
// closer finds the first keyframes again, the loop ICP is accepted and the pose graph is optimised.
#include "common.hpp"
#include <pgslam_amd/slam.hpp>

// The single-thread flavour's deferred host compaction (round 5): with the sensor at the robot's origin the input stage leaves the
// dropped points in the host cloud while the ICP runs on the (complete) device copy, and a worker thread closes the gaps meanwhile.
// Same poses and the same clouds as the synchronous stage, bit for bit.
template <typename T>
void run_deferred_compaction(const char *name)
{
    IMPORT_PGSLAM_TYPES(T)
    TransformationPtr rigid = PM::get().REG(Transformation).create("RigidTransformation");
    const int S = 10;
    // (a vehicle-box filter and a range cut that bites rarely: a few hundred of a scan's 9 750 points go -- the deferred form lists up to 4 096)
    const char *filters = "- MaxDistDataPointsFilter:\n    maxDist: 3.4\n- BoundingBoxDataPointsFilter:\n    xMin: -0.6\n    xMax: 0.6\n    yMin: -0.6\n    yMax: 0.6\n    zMin: -1\n    zMax: 1\n    removeInside: 1\n";
    std::vector<Matrix> truth, odom, sync_poses;
    for (int s = 0; s < S; s++) truth.push_back(pose<T>(1.2 + 0.06 * s, 1.4 + 0.03 * s, 0.0, 0.02 * s));
    odom.push_back(truth[0]);
    for (int s = 1; s < S; s++) odom.push_back(odom[s - 1] * (truth[s - 1].inverse() * truth[s]) * pose<T>(0.010, -0.008, 0.0, 0.004));
    std::vector<unsigned> sync_counts;
    std::vector<T> sync_sum;
    for (int deferred = 0; deferred < 2; deferred++) {
        if (deferred) unsetenv("PGSLAM_SYNC_HOST_COMPACTION"); else setenv("PGSLAM_SYNC_HOST_COMPACTION", "1", 1);
        pgslam::PoseGraphSlam<T> slam;
        slam.SetIcpConfigFromStrings(filters, kIcpYaml, kIcpYaml);
        slam.localizer().SetOverlapThreshold(T(0.9));
        for (int s = 0; s < S; s++) {
            auto cloud = std::make_shared<DP>(rigid->compute(make_corner<T>(3000, 370 + s, 0.004), truth[s].inverse()));
            const unsigned n_raw = cloud->getNbPoints();
            slam.AddData((unsigned long long)s, "world", odom[s], Matrix::Identity(4, 4), cloud);
            T sum = 0;
            for (unsigned i = 0; i < cloud->getNbPoints(); i++) sum += cloud->features(0, (int)i) + T(2) * cloud->features(1, (int)i) + T(3) * cloud->features(2, (int)i) + cloud->normalsPtr()[(size_t)i * cloud->normalsStride()];
            if (!deferred) { sync_poses.push_back(slam.localizer().T_world_robot()); sync_counts.push_back(cloud->getNbPoints()); sync_sum.push_back(sum); CHECK(cloud->getNbPoints() < n_raw); }
            else {
                CHECK(pose_diff(slam.localizer().T_world_robot(), sync_poses[s]) == 0.0);
                CHECK(cloud->getNbPoints() == sync_counts[s] && sum == sync_sum[s]);
            }
        }
        CHECK(slam.localizer().device_input_stages() == (size_t)S);
        std::printf("  (%s: %zu of %d scans compacted on the worker thread)\n", deferred ? "deferred" : "synchronous", slam.localizer().deferred_compactions(), S);
        CHECK(deferred ? slam.localizer().deferred_compactions() >= (size_t)S / 2 : slam.localizer().deferred_compactions() == 0);
    }
    unsetenv("PGSLAM_SYNC_HOST_COMPACTION");
    std::printf("%s: ok  (%d scans, gaps closed on the worker thread during the ICP; poses and clouds equal the synchronous stage's)\n", name, S);
}

template <typename T>
void run(const char *name)
{
    IMPORT_PGSLAM_TYPES(T)
    pgslam::PoseGraphSlam<T> slam;
    slam.SetIcpConfigFromStrings("- IdentityDataPointsFilter\n", kIcpYaml, kIcpYaml);
    slam.localizer().SetOverlapThreshold(T(0.9));
    slam.loop_closer().SetTopologicalDistanceThreshold(T(1.0));
    slam.loop_closer().SetGeometricalDistanceThreshold(T(0.3));
    TransformationPtr rigid = PM::get().REG(Transformation).create("RigidTransformation");
    const int S = 15;
    std::vector<Matrix> truth, odom;
    for (int s = 0; s < S; s++) {
        const double a = 2 * M_PI * s / (S - 1);                        // last pose = first pose
        truth.push_back(pose<T>(1.5 + 0.5 * std::cos(a), 1.5 + 0.5 * std::sin(a), 0.0, a * 0.2));
    }
    odom.push_back(truth[0]);
    for (int s = 1; s < S; s++) odom.push_back(odom[s - 1] * (truth[s - 1].inverse() * truth[s]) * pose<T>(0.012, -0.009, 0.0, 0.005));
    for (int s = 0; s < S; s++) {
        auto cloud = std::make_shared<DP>(rigid->compute(make_corner<T>(2000, 70 + s, 0.004), truth[s].inverse()));
        slam.AddData((unsigned long long)s, "world", odom[s], Matrix::Identity(4, 4), cloud);
        CHECK(pose_diff(slam.localizer().T_world_robot(), truth[s]) < 3e-2);
    }
    auto &g = slam.map_manager().GetGraph();
    CHECK(g.NumVertices() == (size_t)S);                                // one keyframe per scan
    int loops = 0;
    for (size_t e = 0; e < g.NumEdges(); e++) loops += g.Edge(e).c.type == Constraint::kLoopConstraint;
    CHECK(loops >= 1 && loops == slam.loop_closer().loops_closed());
    CHECK(slam.optimizer().last_iterations() >= 1 && slam.optimizer().last_final_error() <= slam.optimizer().last_initial_error());
    // the optimised keyframe poses sit on the true trajectory (odometry alone ends ~10 cm off)
    double worst = 0;
    for (size_t v = 0; v < g.NumVertices(); v++) worst = std::max(worst, pose_diff(g[v].optimized_T_world_kf, truth[v]));
    CHECK(worst < 3e-2);
    CHECK(pose_diff(odom[S - 1], truth[S - 1]) > 5e-2);
    slam.WriteGraphviz("/tmp/pgslam_amd_graph.dot");
    std::ifstream dot("/tmp/pgslam_amd_graph.dot");
    std::string text((std::istreambuf_iterator<char>(dot)), std::istreambuf_iterator<char>());
    CHECK(text.find("graph G") != std::string::npos && text.find("--") != std::string::npos);
    std::printf("%s: ok  (%zu keyframes, %d loop edges, %d optimiser iterations, worst keyframe error %.2e)\n", name, g.NumVertices(), loops,
                slam.optimizer().last_iterations(), worst);
}

// The multi-thread flavour (PoseGraphSlamMT.hpp:21-26) on the same drive.  First in lock step (WaitIdle after every scan:
// the three workers then do exactly what the single-thread facade does), then free running (all scans queued at once:
// loop closing and optimisation land while the localizer is already further along).
template <typename T>
void run_mt(const char *name)
{
    IMPORT_PGSLAM_TYPES(T)
    TransformationPtr rigid = PM::get().REG(Transformation).create("RigidTransformation");
    const int S = 15;
    std::vector<Matrix> truth, odom;
    for (int s = 0; s < S; s++) {
        const double a = 2 * M_PI * s / (S - 1);
        truth.push_back(pose<T>(1.5 + 0.5 * std::cos(a), 1.5 + 0.5 * std::sin(a), 0.0, a * 0.2));
    }
    odom.push_back(truth[0]);
    for (int s = 1; s < S; s++) odom.push_back(odom[s - 1] * (truth[s - 1].inverse() * truth[s]) * pose<T>(0.012, -0.009, 0.0, 0.005));
    for (int free_running = 0; free_running < 3; free_running++) {
        pgslam::PoseGraphSlamMT<T> slam;
        slam.SetIcpConfigFromStrings("- IdentityDataPointsFilter\n", kIcpYaml, kIcpYaml);
        slam.localizer().SetOverlapThreshold(T(0.9));
        slam.loop_closer().SetTopologicalDistanceThreshold(T(1.0));
        slam.loop_closer().SetGeometricalDistanceThreshold(T(0.3));
        const bool input_thread = free_running == 2;                    // (third run: free running with the input stage on its own thread)
        slam.localizer().SetInputThread(input_thread);
        slam.Run();
        for (int s = 0; s < S; s++) {
            auto cloud = std::make_shared<DP>(rigid->compute(make_corner<T>(2000, 70 + s, 0.004), truth[s].inverse()));
            slam.AddData((unsigned long long)s, "world", odom[s], Matrix::Identity(4, 4), cloud);
            if (!free_running) {
                slam.WaitIdle();
                CHECK(pose_diff(slam.localizer().T_world_robot(), truth[s]) < 3e-2);
            }
        }
        slam.WaitIdle();
        CHECK(slam.localizer().processed() == (size_t)S);
        // with the input thread every scan's input stage runs there, up to two scans ahead of the one being aligned
        // (LocalizerMT.hpp:27-40: the queue holds them by then); without it (the default) on the localizer's own thread
        CHECK(slam.localizer().prefetches() == (input_thread ? (size_t)S : 0));
        auto lock = slam.map_manager().GetGraphLock();
        auto &g = slam.map_manager().GetGraph();
        CHECK(g.NumVertices() == (size_t)S);
        int loops = 0;
        for (size_t e = 0; e < g.NumEdges(); e++) loops += g.Edge(e).c.type == Constraint::kLoopConstraint;
        CHECK(loops >= 1 && loops == slam.loop_closer().loops_closed() && slam.optimizer().runs() >= 1);
        double worst = 0;
        for (size_t v = 0; v < g.NumVertices(); v++) worst = std::max(worst, pose_diff(g[v].optimized_T_world_kf, truth[v]));
        CHECK(worst < (free_running ? 6e-2 : 3e-2));
        CHECK(pose_diff(slam.localizer().T_world_robot(), truth[S - 1]) < (free_running ? 6e-2 : 3e-2));
        std::printf("%s, %s: ok  (%zu keyframes, %d loop edges in %d device batch(es), largest %d; %d optimiser run(s); worst keyframe error %.2e)\n",
                    name, free_running == 2 ? "free running, input thread" : free_running ? "free running" : "lock step", g.NumVertices(), loops, slam.loop_closer().batches(),
                    slam.loop_closer().largest_batch(), slam.optimizer().runs(), worst);
    }
}

// A sensor that is NOT at the robot's origin and a sampling input filter (Localizer.hpp:103-106: filters in place, then
// sensor -> robot), single thread against free-running MT: in the MT flavour the next scan is pre-processed and uploaded
// while the current one aligns (LocalizerMT.hpp:27-40), and every cloud must go through the filters and the sensor
// transform exactly ONCE -- a second pass would halve the cloud again and move it by T_robot_sensor twice.
template <typename T>
void run_mt_sensor_pose(const char *name)
{
    IMPORT_PGSLAM_TYPES(T)
    TransformationPtr rigid = PM::get().REG(Transformation).create("RigidTransformation");
    const int S = 12;
    const Matrix T_robot_sensor = pose<T>(0.30, -0.10, 0.20, 0.10, 0.02, -0.03);
    const char *filters = "- FixStepSamplingDataPointsFilter:\n    startStep: 2\n";
    std::vector<Matrix> truth, odom;
    for (int s = 0; s < S; s++) truth.push_back(pose<T>(1.2 + 0.06 * s, 1.4 + 0.03 * s, 0.0, 0.02 * s));
    odom.push_back(truth[0]);
    for (int s = 1; s < S; s++) odom.push_back(odom[s - 1] * (truth[s - 1].inverse() * truth[s]) * pose<T>(0.010, -0.008, 0.0, 0.004));
    auto sensor_cloud = [&](int s) {
        // the scene as the SENSOR sees it: world -> robot -> sensor
        return std::make_shared<DP>(rigid->compute(make_corner<T>(2000, 170 + s, 0.004), (truth[s] * T_robot_sensor).inverse()));
    };
    const unsigned n_raw = sensor_cloud(0)->getNbPoints();
    std::vector<Matrix> st_poses;
    for (int device_stage = 0; device_stage < 2; device_stage++) {
        // the input stage (filters + sensor transform) on the host, filter by filter, and as ONE device pass
        // (pgicp_filter_cloud): the same kept points, the same arithmetic -- the same poses, bit for bit
        pgslam::PoseGraphSlam<T> slam;
        slam.SetIcpConfigFromStrings(filters, kIcpYaml, kIcpYaml);
        slam.localizer().SetOverlapThreshold(T(0.9));
        slam.localizer().SetDeviceInputStage(device_stage != 0);
        for (int s = 0; s < S; s++) {
            auto cloud = sensor_cloud(s);
            slam.AddData((unsigned long long)s, "world", odom[s], T_robot_sensor, cloud);
            CHECK(cloud->getNbPoints() == (n_raw + 1) / 2);
            if (device_stage == 0) st_poses.push_back(slam.localizer().T_world_robot());
            else CHECK(pose_diff(slam.localizer().T_world_robot(), st_poses[s]) == 0.0);
            CHECK(pose_diff(slam.localizer().T_world_robot(), truth[s]) < 3e-2);
        }
        CHECK(slam.localizer().device_input_stages() == (device_stage ? (size_t)S : 0));
    }
    pgslam::PoseGraphSlamMT<T> slam;
    slam.SetIcpConfigFromStrings(filters, kIcpYaml, kIcpYaml);
    slam.localizer().SetOverlapThreshold(T(0.9));
    // (the loop closer and the optimiser find nothing to do on this short straight drive: the localizer's poses are
    // then a function of the scans alone, and the two flavours must agree)
    slam.loop_closer().SetGeometricalDistanceThreshold(T(0.0));
    std::vector<DPPtr> clouds;
    for (int s = 0; s < S; s++) clouds.push_back(sensor_cloud(s));
    slam.Run();
    for (int s = 0; s < S; s++) slam.AddData((unsigned long long)s, "world", odom[s], T_robot_sensor, clouds[s]);
    slam.WaitIdle();
    CHECK(slam.localizer().processed() == (size_t)S);
    CHECK(slam.localizer().prefetches() == (size_t)S);                                   // (the input-stage thread, on by default, handed every scan over pre-processed)
    CHECK(slam.localizer().device_readings_used() == (size_t)S - 1);                     // every ICP ran on the device copy its input stage left (scan 0 has no ICP)
    CHECK(slam.localizer().device_input_stages() == (size_t)S);                          // (the input stage of every scan ran on the device)
    for (int s = 0; s < S; s++) CHECK(clouds[s]->getNbPoints() == (n_raw + 1) / 2);      // filtered once, in place
    CHECK(pose_diff(slam.localizer().T_world_robot(), truth[S - 1]) < 3e-2);
    CHECK(pose_diff(slam.localizer().T_world_robot(), st_poses[S - 1]) < 1e-4);
    std::printf("%s: ok  (%zu of %d scans aligned on a device copy uploaded ahead; final pose %.1e from the single-thread flavour's)\n", name,
                slam.localizer().device_readings_used(), S, pose_diff(slam.localizer().T_world_robot(), st_poses[S - 1]));
}

// Local maps assembled from keyframe clouds that STAY in device memory (Localizer::Rebuild, OverlapWith: one upload per
// keyframe, pgicp_build_local_map + pgicp_map_create on device pointers) against the host flow (LocalMap.hpp:209-224 through the
// host at every rebuild): the same kernels on the same values -- poses, keyframes, loop edges and the map itself bit for bit.
template <typename T>
void run_device_local_map(const char *name)
{
    IMPORT_PGSLAM_TYPES(T)
    TransformationPtr rigid = PM::get().REG(Transformation).create("RigidTransformation");
    const int S = 15;
    std::vector<Matrix> truth, odom;
    for (int s = 0; s < S; s++) {
        const double a = 2 * M_PI * s / (S - 1);
        truth.push_back(pose<T>(1.5 + 0.5 * std::cos(a), 1.5 + 0.5 * std::sin(a), 0.0, a * 0.2));
    }
    odom.push_back(truth[0]);
    for (int s = 1; s < S; s++) odom.push_back(odom[s - 1] * (truth[s - 1].inverse() * truth[s]) * pose<T>(0.012, -0.009, 0.0, 0.005));
    std::vector<Matrix> host_poses, host_kf;
    DP host_map;
    int host_loops = 0, host_rebuilds = 0;
    for (int on_device = 0; on_device < 2; on_device++) {
        pgslam::PoseGraphSlam<T> slam;
        slam.SetIcpConfigFromStrings("- IdentityDataPointsFilter\n", kIcpYaml, kIcpYaml);
        // (threshold 0.8: some scans join the current map, some become keyframes, and the neighbour-composition probe runs)
        slam.localizer().SetOverlapThreshold(T(0.8));
        slam.localizer().SetDeviceLocalMap(on_device != 0);
        slam.loop_closer().SetDeviceCandidates(on_device != 0);          // (the loop closer's candidate maps: assembled in HBM / through the host)
        slam.loop_closer().SetTopologicalDistanceThreshold(T(1.0));
        slam.loop_closer().SetGeometricalDistanceThreshold(T(0.3));
        for (int s = 0; s < S; s++) {
            auto cloud = std::make_shared<DP>(rigid->compute(make_corner<T>(2000, 70 + s, 0.004), truth[s].inverse()));
            slam.AddData((unsigned long long)s, "world", odom[s], Matrix::Identity(4, 4), cloud);
            if (!on_device) host_poses.push_back(slam.localizer().T_world_robot());
            else CHECK(pose_diff(slam.localizer().T_world_robot(), host_poses[s]) == 0.0);
        }
        auto &g = slam.map_manager().GetGraph();
        int loops = 0;
        for (size_t e = 0; e < g.NumEdges(); e++) loops += g.Edge(e).c.type == Constraint::kLoopConstraint;
        const DP &map = slam.localizer().icp().getPrefilteredMap();          // (device flow: the host copy is made here, on demand)
        if (!on_device) {
            for (size_t v = 0; v < g.NumVertices(); v++) host_kf.push_back(g[v].optimized_T_world_kf);
            host_loops = loops; host_rebuilds = slam.localizer().rebuilds(); host_map = map;
            CHECK(slam.localizer().device_rebuilds() == 0 && slam.loop_closer().device_candidates() == 0);
        } else {
            CHECK(slam.loop_closer().device_candidates() == (size_t)slam.loop_closer().candidates_tried());
            CHECK(g.NumVertices() == host_kf.size() && loops == host_loops && slam.localizer().rebuilds() == host_rebuilds);
            for (size_t v = 0; v < g.NumVertices(); v++) CHECK(pose_diff(g[v].optimized_T_world_kf, host_kf[v]) == 0.0);
            CHECK(slam.localizer().device_rebuilds() == (size_t)host_rebuilds);
            CHECK(map.getNbPoints() == host_map.getNbPoints() && map.getNbPoints() > 0);
            bool same = true;
            for (unsigned i = 0; i < map.getNbPoints() && same; i++)
                for (int a = 0; a < 3; a++)
                    same = same && map.features(a, (int)i) == host_map.features(a, (int)i) && map.normalsPtr()[(size_t)i * map.normalsStride() + a] == host_map.normalsPtr()[(size_t)i * host_map.normalsStride() + a];
            CHECK(same);
            std::printf("%s: ok  (%zu keyframes, %d rebuilds all in device memory, %d loop edges; poses, keyframes and the map equal the host flow's bit for bit)\n",
                        name, g.NumVertices(), host_rebuilds, loops);
        }
    }
}

// Keyframe clouds in device memory live under a budget (MapManager::EnsureKeyframeResident): with room for only a few of them the least
// recently used copies are dropped and made again when a composition names the keyframe once more -- the same poses, keyframes and
// loop edges as with every keyframe resident.
template <typename T>
void run_keyframe_residency(const char *name)
{
    IMPORT_PGSLAM_TYPES(T)
    TransformationPtr rigid = PM::get().REG(Transformation).create("RigidTransformation");
    const int S = 40;
    std::vector<Matrix> truth, odom, ref_poses;
    for (int s = 0; s < S; s++) {
        const double a = 2 * M_PI * s / (S - 1);
        truth.push_back(pose<T>(1.5 + 0.5 * std::cos(a), 1.5 + 0.5 * std::sin(a), 0.0, a * 0.2));
    }
    odom.push_back(truth[0]);
    for (int s = 1; s < S; s++) odom.push_back(odom[s - 1] * (truth[s - 1].inverse() * truth[s]) * pose<T>(0.004, -0.003, 0.0, 0.002));
    size_t loops_ref = 0;
    for (int tight = 0; tight < 2; tight++) {
        pgslam::PoseGraphSlam<T> slam;
        if (tight) slam.map_manager().SetDeviceKeyframeBudgetMB(0.5);    // (a keyframe of this drive holds ~0.17 MB: the 16 most recently used stay)
        slam.SetIcpConfigFromStrings("- IdentityDataPointsFilter\n", kIcpYaml, kIcpYaml);
        slam.localizer().SetOverlapThreshold(T(0.9));                    // every scan a keyframe
        slam.loop_closer().SetTopologicalDistanceThreshold(T(1.0));
        slam.loop_closer().SetGeometricalDistanceThreshold(T(0.3));
        for (int s = 0; s < S; s++) {
            auto cloud = std::make_shared<DP>(rigid->compute(make_corner<T>(1500, 570 + s, 0.004), truth[s].inverse()));
            slam.AddData((unsigned long long)s, "world", odom[s], Matrix::Identity(4, 4), cloud);
            if (!tight) ref_poses.push_back(slam.localizer().T_world_robot());
            else CHECK(pose_diff(slam.localizer().T_world_robot(), ref_poses[s]) == 0.0);
        }
        auto &mm = slam.map_manager();
        if (!tight) { loops_ref = (size_t)slam.loop_closer().loops_closed(); CHECK(mm.device_evictions() == 0 && mm.resident_keyframes() == (size_t)S); }
        else {
            CHECK(mm.device_evictions() > 0 && mm.resident_keyframes() <= 17 && mm.device_uploads() > (size_t)S);
            CHECK((size_t)slam.loop_closer().loops_closed() == loops_ref);
            std::printf("%s: ok  (%d keyframes, %zu resident, %zu uploads, %zu evictions; poses and loops as with all of them resident)\n", name, S,
                        mm.resident_keyframes(), mm.device_uploads(), mm.device_evictions());
        }
    }
}

// The batched dispatcher INSIDE the facade (LoopCloserMT.hpp:26-34 queues vertices, :45-67 pops them one at a time; here the
// worker drains its queue into one device batch): the loop closer is held back while the localizer makes fifteen keyframes, then
// let go -- every waiting vertex becomes a candidate of ONE batch (LoopClosureBatch over device-resident clouds) -- against the same
// drive with the dispatcher cut to one vertex per batch, upstream's way.  The optimiser is held back in both runs, so that both
// see the same graph: the edges must be the same, bit for bit.
template <typename T>
void run_mt_batched_dispatcher(const char *name)
{
    IMPORT_PGSLAM_TYPES(T)
    TransformationPtr rigid = PM::get().REG(Transformation).create("RigidTransformation");
    const int S = 15;
    std::vector<Matrix> truth, odom;
    for (int s = 0; s < S; s++) {
        const double a = 2 * M_PI * s / (S - 1);
        truth.push_back(pose<T>(1.5 + 0.5 * std::cos(a), 1.5 + 0.5 * std::sin(a), 0.0, a * 0.2));
    }
    odom.push_back(truth[0]);
    for (int s = 1; s < S; s++) odom.push_back(odom[s - 1] * (truth[s - 1].inverse() * truth[s]) * pose<T>(0.012, -0.009, 0.0, 0.005));
    std::vector<pgicp_edge> edges[2];
    int largest[2] = {0, 0}, batches[2] = {0, 0};
    for (int one_at_a_time = 0; one_at_a_time < 2; one_at_a_time++) {
        pgslam::PoseGraphSlamMT<T> slam;
        slam.SetIcpConfigFromStrings("- IdentityDataPointsFilter\n", kIcpYaml, kIcpYaml);
        slam.localizer().SetOverlapThreshold(T(0.9));
        slam.loop_closer().SetTopologicalDistanceThreshold(T(1.0));
        slam.loop_closer().SetGeometricalDistanceThreshold(T(0.6));
        slam.loop_closer().Pause();
        slam.optimizer().Pause();
        if (one_at_a_time) slam.loop_closer().SetMaxBatch(1);
        slam.Run();
        for (int s = 0; s < S; s++) {
            auto cloud = std::make_shared<DP>(rigid->compute(make_corner<T>(2000, 70 + s, 0.004), truth[s].inverse()));
            slam.AddData((unsigned long long)s, "world", odom[s], Matrix::Identity(4, 4), cloud);
        }
        slam.WaitIdle();                                                 // (a paused worker counts as resting)
        CHECK(slam.localizer().processed() == (size_t)S && slam.loop_closer().queued() == (size_t)S - 1 && slam.loop_closer().batches() == 0);
        slam.loop_closer().Resume();
        while (slam.loop_closer().queued() > 0 || !slam.loop_closer().Idle()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        slam.RethrowWorkerError();
        edges[one_at_a_time] = slam.loop_closer().edges();
        largest[one_at_a_time] = slam.loop_closer().largest_batch();
        batches[one_at_a_time] = slam.loop_closer().batches();
        CHECK(slam.loop_closer().device_batches() == (size_t)batches[one_at_a_time]);      // candidates assembled and aligned in device memory
        slam.optimizer().Resume();
        slam.WaitIdle();
        CHECK(slam.optimizer().runs() == (slam.loop_closer().loops_closed() > 0 ? 1 : 0)); // one solve drains every accepted edge (OptimizerMT.hpp:59-65)
    }
    CHECK(batches[0] == 1 && largest[0] >= 3);                            // the batch path closed the loops: several candidates in one device batch
    CHECK(batches[1] == largest[0] && largest[1] == 1);
    CHECK(edges[0].size() == edges[1].size() && !edges[0].empty());
    int accepted = 0;
    for (size_t k = 0; k < edges[0].size() && k < edges[1].size(); k++) {
        const pgicp_edge &a = edges[0][k], &b = edges[1][k];
        CHECK(a.from_id == b.from_id && a.to_id == b.to_id && a.accepted == b.accepted && a.iterations == b.iterations && a.status == b.status);
        CHECK(std::memcmp(a.T_from_to, b.T_from_to, sizeof a.T_from_to) == 0 && a.residual == b.residual && a.overlap == b.overlap);
        CHECK(std::memcmp(a.cov, b.cov, sizeof a.cov) == 0);
        accepted += a.accepted;
    }
    CHECK(accepted >= 1);
    std::printf("%s: ok  (%zu candidates as ONE device batch == the same %zu one at a time, bit for bit; %d accepted)\n", name, edges[0].size(), edges[1].size(), accepted);
}

// Clouds that carry `simpleSensorNoise` (the user's input filters add it): getOverlap() takes its sensor-noise branch, which reads
// the ICP's last error elements -- the MT loop closer sends such candidates one at a time (PairLoopCloser) instead of through the
// device batch, whose fused residual pass replaces those elements.  Same candidates, same poses; another overlap figure.
template <typename T>
void run_mt_sensor_noise(const char *name)
{
    IMPORT_PGSLAM_TYPES(T)
    TransformationPtr rigid = PM::get().REG(Transformation).create("RigidTransformation");
    const int S = 15;
    std::vector<Matrix> truth, odom;
    for (int s = 0; s < S; s++) {
        const double a = 2 * M_PI * s / (S - 1);
        truth.push_back(pose<T>(1.5 + 0.5 * std::cos(a), 1.5 + 0.5 * std::sin(a), 0.0, a * 0.2));
    }
    odom.push_back(truth[0]);
    for (int s = 1; s < S; s++) odom.push_back(odom[s - 1] * (truth[s - 1].inverse() * truth[s]) * pose<T>(0.012, -0.009, 0.0, 0.005));
    std::vector<pgicp_edge> edges[2];
    for (int noisy = 0; noisy < 2; noisy++) {
        pgslam::PoseGraphSlamMT<T> slam;
        slam.SetIcpConfigFromStrings(noisy ? "- SimpleSensorNoiseDataPointsFilter:\n    sensorType: 0\n    gain: 1\n" : "- IdentityDataPointsFilter\n", kIcpYaml, kIcpYaml);
        slam.localizer().SetOverlapThreshold(T(0.9999));             // (every scan a keyframe, whichever overlap figure the drive reads)
        slam.loop_closer().SetTopologicalDistanceThreshold(T(1.0));
        slam.loop_closer().SetGeometricalDistanceThreshold(T(0.6));
        slam.loop_closer().SetOverlapThreshold(T(0.3));
        slam.loop_closer().Pause();
        slam.optimizer().Pause();
        slam.Run();
        for (int s = 0; s < S; s++) {
            auto cloud = std::make_shared<DP>(rigid->compute(make_corner<T>(2000, 70 + s, 0.004), truth[s].inverse()));
            slam.AddData((unsigned long long)s, "world", odom[s], Matrix::Identity(4, 4), cloud);
        }
        slam.WaitIdle();
        slam.loop_closer().Resume();
        while (slam.loop_closer().queued() > 0 || !slam.loop_closer().Idle()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        slam.RethrowWorkerError();
        edges[noisy] = slam.loop_closer().edges();
        CHECK(slam.loop_closer().batches() >= 1);
        CHECK(noisy ? slam.loop_closer().device_batches() == 0 : slam.loop_closer().device_batches() >= 1);
        slam.optimizer().Resume();
        slam.WaitIdle();
    }
    CHECK(!edges[0].empty());
    // (the keyframes of the two drives may differ -- the localizer's own getOverlap() decides on new keyframes -- so the candidates are
    // compared where both drives made the same pair)
    int same_pairs = 0, other_overlap = 0;
    for (const pgicp_edge &a : edges[0])
        for (const pgicp_edge &b : edges[1])
            if (a.from_id == b.from_id && a.to_id == b.to_id) { same_pairs++; other_overlap += a.overlap != b.overlap; }
    CHECK(!edges[1].empty());
    for (const pgicp_edge &b : edges[1]) CHECK(b.overlap > 0.0 && b.overlap <= 1.0);
    std::printf("%s: ok  (%zu candidates through the device batch, %zu one at a time with the sensor-noise overlap; %d pairs in both, %d with another overlap)\n",
                name, edges[0].size(), edges[1].size(), same_pairs, other_overlap);
}

int main()
{
    run_mt_sensor_noise<float>("PoseGraphSlamMT<float>, clouds with simpleSensorNoise");
    run_keyframe_residency<float>("keyframe clouds under a device-memory budget, PoseGraphSlam<float>");
    run_deferred_compaction<float>("deferred host compaction, PoseGraphSlam<float>");
    run_deferred_compaction<double>("deferred host compaction, PoseGraphSlam<double>");
    run_mt_batched_dispatcher<float>("batched dispatcher inside PoseGraphSlamMT<float>");
    run_device_local_map<float>("local maps from device-resident keyframes, PoseGraphSlam<float>");
    run_device_local_map<double>("local maps from device-resident keyframes, PoseGraphSlam<double>");
    run_mt_sensor_pose<float>("PoseGraphSlamMT<float>, sensor off the robot's origin + sampling input filter");
    run_mt<float>("PoseGraphSlamMT<float>");
    run<float>("PoseGraphSlam<float>");
    run<double>("PoseGraphSlam<double>");
    std::puts("slam gpu tests ok");
    return 0;
}
