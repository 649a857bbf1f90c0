// GPU checks of the C++ drop-in layer: the same call sequences pgslam makes
// (reference src/pgslam/Localizer.hpp:126,148,282-348; LoopCloser.hpp:83-110,
// 308-365; LocalMap.hpp:209-224), on a synthetic room-corner scene with a known
// answer.  Mirrors tests/instantiation.cpp by running for float and double.
#include "common.hpp"

template <typename T>
void run(const char *name)
{
    IMPORT_PGSLAM_TYPES(T)
    const DP map = make_corner<T>(6000, 11, 0.004);
    const DP scan_world = make_corner<T>(2500, 12, 0.004);
    // the scan is observed from a robot pose P: reading = P^-1 * scan_world
    const Matrix P = pose<T>(0.40, -0.25, 0.10, 0.06, 0.01, -0.015);
    TransformationPtr rigid = PM::get().REG(Transformation).create("RigidTransformation");
    const DP reading = rigid->compute(scan_world, P.inverse());
    CHECK(reading.descriptorExists("normals") && reading.getNbPoints() == scan_world.getNbPoints());
    const Matrix guess = P * pose<T>(0.08, -0.05, 0.03, 0.02, 0.0, 0.0);

    // --- ICP::operator()(reading, reference, T_init)  (LoopCloser.hpp:98)
    ICP icp;
    { std::istringstream iss(kIcpYaml); icp.loadFromYaml(iss); }
    const Matrix T1 = icp(reading, map, guess);
    CHECK(pose_diff(T1, P) < 5e-3);
    CHECK(!icp.getMaxNumIterationsReached());
    const T overlap = icp.errorMinimizer->getOverlap();
    CHECK(overlap > T(0.84) && overlap <= T(1));
    const Matrix cov = icp.errorMinimizer->getCovariance();
    CHECK(cov.rows() == 6 && cov(0, 0) > 0 && cov(5, 5) > 0 && std::fabs((double)cov(0, 1) - (double)cov(1, 0)) < 1e-9);

    // --- ICPSequence: setMap once, then operator()(cloud, T_init)  (Localizer.hpp:126,148)
    ICPSequence seq;
    { std::istringstream iss(kIcpYaml); seq.loadFromYaml(iss); }
    CHECK(!seq.hasMap());
    CHECK(seq.setMap(map) && seq.hasMap());
    const Matrix T2 = seq(reading, guess);
    CHECK(pose_diff(T1, T2) == 0.0);                                   // same chain, same arithmetic
    ICPSequence first;                                                  // no map yet: first cloud becomes the map
    CHECK(pose_diff(first(map, guess), Matrix::Identity(4, 4)) == 0.0 && first.hasMap());

    // --- hand-driven partial chain, exactly as Localizer::ComputeOverlapWith does (Localizer.hpp:309-347)
    ICP temp;
    { std::istringstream iss(kIcpYaml); temp.loadFromYaml(iss); }
    DP reference(map);
    temp.referenceDataPointsFilters.init(); temp.referenceDataPointsFilters.apply(reference);
    temp.matcher->init(reference);
    DP moved = rigid->compute(reading, T1);
    const typename PM::Matches matches(temp.matcher->findClosests(moved));
    const typename PM::OutlierWeights w(temp.outlierFilters.compute(moved, reference, matches));
    typename PM::ErrorMinimizer::ErrorElements matched(moved, reference, w, matches);
    CHECK(matches.ids.cols() == (int)moved.getNbPoints());
    CHECK(std::fabs((double)matched.weightedPointUsedRatio - 0.85) < 0.01);
    const T residual = temp.errorMinimizer->getResidualError(moved, reference, w, matches);
    CHECK(residual >= 0 && residual < T(1.0));

    // --- pgslam::Localizer hot path
    pgslam::ScanLocalizer<T> loc;
    loc.SetIcpConfigFromString(kIcpYaml);
    loc.SetLocalMap(map, Matrix::Identity(4, 4));
    auto cloud_ptr = std::make_shared<DP>(reading);
    // first call: T_refkf_robot_ = I, odometry says the robot is at `guess`
    // (last_input = I), so the ICP starts from `guess`
    const Matrix T3 = loc.ProcessData(guess, Matrix::Identity(4, 4), cloud_ptr);
    CHECK(pose_diff(T3, T1) == 0.0);
    CHECK(loc.ComputeCurrentOverlap() == overlap && loc.IsOverlapEnough(overlap));
    const T ov2 = loc.ComputeOverlapWith(map);                         // candidate map in world frame (here refkf == world)
    CHECK(std::fabs((double)ov2 - (double)matched.weightedPointUsedRatio) < 1e-6);

    // --- pgslam::LoopCloser hot path
    pgslam::PairLoopCloser<T> lc;
    lc.SetIcpConfigFromString(kIcpYaml);
    auto r = lc.ProcessCandidate(reading, map, guess);
    CHECK(pose_diff(r.T_refkf_kf, T1) == 0.0 && r.accepted && !r.max_iterations_reached);
    CHECK(std::fabs((double)r.residual - (double)residual) <= 1e-3 * (double)residual + 1e-6);
    lc.SetResidualErrorThreshold(T(0));
    CHECK(!lc.CheckIcpResult(r) || r.residual == T(0));

    // --- batch dispatcher: 3 candidates, 2 ranks, union == all, results == single
    pgslam::LoopClosureBatch<T> batch;
    batch.SetIcpConfigFromString(kIcpYaml);
    auto rp = std::make_shared<DP>(reading), mp = std::make_shared<DP>(map);
    for (int k = 0; k < 3; k++) batch.Add({100 + k, 200 + k, rp, mp, guess});
    auto s0 = batch.Shard(2, 0), s1 = batch.Shard(2, 1);
    CHECK(s0.size() + s1.size() == 3);
    auto e0 = batch.Run(s0), e1 = batch.Run(s1);
    for (auto &e : e0) {
        CHECK(e.accepted == 1 && e.status == 0 && e.from_id >= 100);
        CHECK(pose_diff(pgslam_amd::from_row_major16<T>(e.T_from_to), T1) == 0.0);
    }
    CHECK(sizeof(pgicp_edge) == 512 && e1.size() == s1.size());
    {   // a reading that carries `simpleSensorNoise` (and `normals`): getOverlap() takes its sensor-noise branch over the ICP's last
        // error elements -- the pair path computes it, the batch (whose fused residual pass replaces those elements) refuses the candidate
        DP noisy(reading);
        typename PM::SimpleSensorNoiseDataPointsFilter(0, T(1)).inPlaceFilter(noisy);
        CHECK(noisy.descriptorExists("simpleSensorNoise") && noisy.descriptorExists("normals"));
        pgslam::PairLoopCloser<T> pl;
        pl.SetIcpConfigFromString(kIcpYaml);
        const auto rn = pl.ProcessCandidate(noisy, map, guess);
        CHECK(pose_diff(rn.T_refkf_kf, T1) == 0.0);                          // the descriptor changes no pose
        CHECK(pl.icp().errorMinimizer->getWeightedPointUsedRatio() == r.overlap);
        CHECK(rn.overlap != r.overlap && rn.overlap > T(0.3) && rn.overlap <= T(1));
        pgslam::LoopClosureBatch<T> nb2;
        nb2.SetIcpConfigFromString(kIcpYaml);
        nb2.Add({7, 8, std::make_shared<DP>(noisy), mp, guess});
        bool refused_noise = false;
        try { nb2.Run(nb2.Shard(1, 0)); } catch (const std::runtime_error &) { refused_noise = true; }
        CHECK(refused_noise);
    }
    {   // the collective of the C ABI on the real RCCL path, one rank: the gathered list is the queue in order
        char uid[PGICP_UNIQUE_ID_BYTES];
        CHECK(pgicp_comm_unique_id(uid) == PGICP_OK);
        pgicp_ctx *cctx = nullptr;
        CHECK(pgicp_ctx_create(0, &cctx) == PGICP_OK);
        pgicp_comm *comm = nullptr;
        CHECK(pgicp_comm_create(cctx, 1, 0, uid, &comm) == PGICP_OK);
        const auto all = batch.Shard(1, 0);
        const auto ea = batch.Run(all);
        const auto g = batch.Gather(comm, all, ea);
        CHECK(g.size() == 3);
        for (int k = 0; k < 3; k++) {
            CHECK(g[k].from_id == 100 + k && g[k].to_id == 200 + k && g[k].accepted == 1);
            CHECK(std::memcmp(g[k].T_from_to, ea[k].T_from_to, sizeof ea[k].T_from_to) == 0);
        }
        pgicp_comm_destroy(comm);
        pgicp_ctx_destroy(cctx);
    }
    {   // a chain WITH data-point filters gives the batch what the one-at-a-time loop closer gives (both apply them)
        const std::string yaml = std::string("readingDataPointsFilters:\n  - BoundingBoxDataPointsFilter:\n      xMin: -100\n      xMax: 100\n      yMin: -100\n      yMax: 100\n      zMin: -100\n      zMax: 0.9\n      removeInside: 0\n"
                                             "referenceDataPointsFilters:\n  - BoundingBoxDataPointsFilter:\n      xMin: -100\n      xMax: 100\n      yMin: -100\n      yMax: 100\n      zMin: -100\n      zMax: 0.9\n      removeInside: 0\n") + kIcpYamlTail;
        pgslam::PairLoopCloser<T> one;
        one.SetIcpConfigFromString(yaml);
        const auto r1 = one.ProcessCandidate(reading, map, guess);
        pgslam::LoopClosureBatch<T> fb;
        fb.SetIcpConfigFromString(yaml);
        fb.Add({1, 2, rp, mp, guess});
        const auto ef = fb.Run(fb.Shard(1, 0));
        CHECK(ef.size() == 1 && ef[0].status == 0);
        CHECK(pose_diff(pgslam_amd::from_row_major16<T>(ef[0].T_from_to), r1.T_refkf_kf) == 0.0);
    }

    {   // a chain WITH a SurfaceNormalOutlierFilter is never run without it: the batch carries the readings' normals and
        // gives what the one-at-a-time loop closer gives; the residual / overlap probes run stage by stage; a device reading is refused
        const std::string yaml = std::string("matcher:\n  KDTreeMatcher:\n    maxDist: 2.0\noutlierFilters:\n  - TrimmedDistOutlierFilter:\n      ratio: 0.85\n"
                                             "  - SurfaceNormalOutlierFilter:\n      maxAngle: 0.6\nerrorMinimizer:\n  PointToPlaneWithCovErrorMinimizer:\n    sensorStdDev: 0.01\n"
                                             "transformationCheckers:\n  - CounterTransformationChecker:\n      maxIterationCount: 30\n"
                                             "  - DifferentialTransformationChecker:\n      minDiffRotErr: 0.001\n      minDiffTransErr: 0.01\n      smoothLength: 3\n");
        // every seventh reading normal turned away: pairs the filter must drop
        DP turned(reading);
        const int rn = turned.getDescriptorStartingRow("normals");
        for (int j = 0; j < (int)turned.getNbPoints(); j += 7) { const T a = turned.descriptors(rn, j); turned.descriptors(rn, j) = turned.descriptors(rn + 2, j); turned.descriptors(rn + 2, j) = -a; }
        pgslam::PairLoopCloser<T> one, plain;
        one.SetIcpConfigFromString(yaml);
        plain.SetIcpConfigFromString(kIcpYaml);
        const auto r1 = one.ProcessCandidate(turned, map, guess), r0 = plain.ProcessCandidate(turned, map, guess);
        CHECK(r1.overlap < r0.overlap - T(0.05));                          // the filter acted in the ICP ...
        CHECK(r1.residual < r0.residual);                                   // ... and in ComputeResidualError's chain (stage by stage)
        pgslam::LoopClosureBatch<T> nb;
        nb.SetIcpConfigFromString(yaml);
        auto tp = std::make_shared<DP>(turned);
        nb.Add({1, 2, tp, mp, guess});
        nb.Add({3, 4, tp, mp, guess});
        const auto en = nb.Run(nb.Shard(1, 0));
        CHECK(en.size() == 2 && en[0].status == 0);
        CHECK(pose_diff(pgslam_amd::from_row_major16<T>(en[0].T_from_to), r1.T_refkf_kf) == 0.0 && (T)en[0].overlap == r1.overlap);
        CHECK(std::memcmp(en[0].T_from_to, en[1].T_from_to, sizeof en[0].T_from_to) == 0);
        // the overlap probe of the localizer: the same chain, stage by stage
        pgslam::ScanLocalizer<T> ln, lp;
        ln.SetIcpConfigFromString(yaml);
        lp.SetIcpConfigFromString(kIcpYaml);
        const T on = ln.ComputeOverlapOf(turned, r1.T_refkf_kf, map), op = lp.ComputeOverlapOf(turned, r1.T_refkf_kf, map);
        CHECK(on < op - T(0.05) && on > T(0.5));
        // a device reading cannot carry normals: refused for this chain, never run without the filter
        ICPSequence seqn;
        { std::istringstream iss(yaml); seqn.loadFromYaml(iss); }
        seqn.setMap(map);
        CHECK(!seqn.deviceReadingEquivalent());
        auto up = seqn.uploadReading(turned);
        bool refused = false;
        try { seqn(up, guess); } catch (const std::logic_error &) { refused = true; }
        CHECK(refused);
        CHECK(pose_diff(seqn(turned, guess), r1.T_refkf_kf) == 0.0);         // (the host cloud: with the filter, as ICP::operator() above)
    }

    // --- LocalMap::BuildCloudFromData  (LocalMap.hpp:209-224)
    std::vector<Keyframe> kfs(2);
    kfs[0].id = 0; kfs[0].cloud_ptr = mp; kfs[0].optimized_T_world_kf = pose<T>(1, 0, 0, 0.1);
    kfs[1].id = 1; kfs[1].cloud_ptr = rp; kfs[1].optimized_T_world_kf = pose<T>(1.5, 0.2, 0, 0.15);
    const DP local = pgslam::BuildLocalMapCloud<T>(kfs);
    CHECK(local.getNbPoints() == map.getNbPoints() + reading.getNbPoints() && local.descriptorExists("normals"));
    CHECK(local.features(0, 5) == map.features(0, 5) && local.features(3, (int)local.getNbPoints() - 1) == T(1));
    const DP expect = rigid->compute(reading, kfs[0].optimized_T_world_kf.inverse() * kfs[1].optimized_T_world_kf);
    const int off = (int)map.getNbPoints();
    for (int j : {0, 17, (int)reading.getNbPoints() - 1})
        for (int a = 0; a < 3; a++) {
            CHECK(local.features(a, off + j) == expect.features(a, j));
            CHECK(local.descriptors(a, off + j) == expect.descriptors(a, j));
        }

    // --- SurfaceNormalDataPointsFilter from YAML (a reference filter, Localizer.hpp:314-315): a map that
    //     arrives without normals gets them on the device and point-to-plane ICP lands on the same pose
    {
        DP bare(map.features, map.featureLabels);
        CHECK(!bare.descriptorExists("normals"));
        std::istringstream fs("- SurfaceNormalDataPointsFilter:\n    knn: 12\n    maxDist: 1.0\n    keepEigenValues: 1\n");
        typename PM::DataPointsFilters nf(fs);
        nf.init(); nf.apply(bare);
        CHECK(bare.descriptorExists("normals") && bare.descriptorExists("eigValues") && bare.descriptors.rows() == 6);
        int agree = 0;
        const int n = (int)bare.getNbPoints(), r0 = bare.getDescriptorStartingRow("normals"), t0 = map.getDescriptorStartingRow("normals");
        for (int j = 0; j < n; j++) {
            double dot = 0, nn = 0;
            for (int k = 0; k < 3; k++) { dot += (double)bare.descriptors(r0 + k, j) * (double)map.descriptors(t0 + k, j); nn += (double)bare.descriptors(r0 + k, j) * (double)bare.descriptors(r0 + k, j); }
            CHECK(std::fabs(nn - 1.0) < 1e-4);
            if (std::fabs(dot) > 0.95) agree++;
        }
        CHECK(agree > 0.9 * n);                                      // edges of the corner scene are the exceptions
        ICP icp2;
        { std::istringstream iss(kIcpYaml); icp2.loadFromYaml(iss); }
        CHECK(pose_diff(icp2(reading, bare, guess), P) < 1e-2);
    }

    // --- StreamingLocalizer: ProcessData + graph-free UpdateAfterIcp on a sliding LocalMap
    {
        // the robot walks through the corner scene; every scan is the scene seen from its true pose, the
        // odometry reports the true increments with a small error
        std::vector<Matrix> truth;
        for (int s = 0; s < 7; s++) truth.push_back(pose<T>(0.25 * s, 0.1 * s, 0.0, 0.03 * s));
        auto scan_at = [&](int s) { return std::make_shared<DP>(rigid->compute(make_corner<T>(2500, 40 + s, 0.004), truth[s].inverse())); };
        std::vector<Matrix> odom(truth);
        for (int s = 1; s < 7; s++) odom[s] = odom[s - 1] * (truth[s - 1].inverse() * truth[s]) * pose<T>(0.01, -0.008, 0.0, 0.004);
        // threshold above what the trimmed filter can ever report (0.85): every scan becomes a keyframe
        pgslam::StreamingLocalizer<T> every(3);
        every.SetIcpConfigFromString(kIcpYaml);
        every.SetOverlapThreshold(T(0.9));
        for (int s = 0; s < 7; s++) {
            const Matrix Tw = every.ProcessData(odom[s], Matrix::Identity(4, 4), scan_at(s));
            CHECK(pose_diff(Tw, truth[s]) < 2e-2);                          // odometry alone would be ~6 cm off by the end
        }
        CHECK(every.local_map().Data().size() == 3 && every.rebuilds() == 7);
        CHECK(every.local_map().Data()[0].id == 4 && every.local_map().ReferenceKeyframe().id == 6);
        CHECK(every.local_map().Cloud().getNbPoints() == 3 * scan_at(0)->getNbPoints());
        // threshold the filter always meets: one keyframe, no rebuild after the first, pose still tracked
        pgslam::StreamingLocalizer<T> never(3);
        never.SetIcpConfigFromString(kIcpYaml);
        never.SetOverlapThreshold(T(0.5));
        for (int s = 0; s < 7; s++) CHECK(pose_diff(never.ProcessData(odom[s], Matrix::Identity(4, 4), scan_at(s)), truth[s]) < 2e-2);
        CHECK(never.local_map().Data().size() == 1 && never.rebuilds() == 1 && never.last_overlap() >= T(0.84));
        // case #2: with two keyframes in the window, walking back towards the older one makes it the reference
        pgslam::StreamingLocalizer<T> back(3);
        back.SetIcpConfigFromString(kIcpYaml);
        back.SetOverlapThreshold(T(0.9));
        back.ProcessData(truth[0], Matrix::Identity(4, 4), scan_at(0));
        back.ProcessData(truth[4], Matrix::Identity(4, 4), scan_at(4));      // second keyframe (ids 0, 1), reference = 1
        back.SetOverlapThreshold(T(0.5));
        back.ProcessData(truth[3], Matrix::Identity(4, 4), scan_at(3));
        CHECK(back.local_map().ReferenceKeyframe().id == 1);
        const Matrix Tb = back.ProcessData(truth[1], Matrix::Identity(4, 4), scan_at(1));
        CHECK(back.local_map().ReferenceKeyframe().id == 0 && back.local_map().Data().size() == 2);
        CHECK(pose_diff(Tb, truth[1]) < 2e-2);
    }

    // --- the chain's other modules from YAML (round 4): knn > 1, PointToPointErrorMinimizer, SurfaceNormalOutlierFilter,
    //     BoundTransformationChecker -- any of them may stand in a pgslam user's config (Localizer.hpp:70, LoopCloser.hpp:73)
    {
        auto chain = [](const char *matcher, const char *outliers, const char *minimizer, const char *checkers) {
            return std::string("matcher:\n  KDTreeMatcher:\n    maxDist: 2.0\n") + matcher + "outlierFilters:\n  - TrimmedDistOutlierFilter:\n      ratio: 0.85\n" +
                   outliers + "errorMinimizer:\n  " + minimizer + "\ntransformationCheckers:\n  - CounterTransformationChecker:\n      maxIterationCount: 30\n"
                   "  - DifferentialTransformationChecker:\n      minDiffRotErr: 0.001\n      minDiffTransErr: 0.01\n      smoothLength: 3\n" + checkers;
        };
        ICP k3;
        { std::istringstream iss(chain("    knn: 3\n", "", "PointToPlaneWithCovErrorMinimizer", "")); k3.loadFromYaml(iss); }
        CHECK(pose_diff(k3(reading, map, guess), P) < 1e-2 && k3.matcher->knn == 3);
        CHECK(k3.errorMinimizer->getOverlap() > T(0.84) && k3.errorMinimizer->getOverlap() <= T(0.8501));
        // ... and through the hand-driven partial chain: knn x N matches, weights, error elements
        k3.matcher->init(map);
        const typename PM::Matches m3(k3.matcher->findClosests(moved));
        CHECK(m3.ids.rows() == 3 && m3.ids.cols() == (int)moved.getNbPoints());
        for (int j : {0, 7, 1000}) CHECK(m3.dists(0, j) <= m3.dists(1, j) && m3.dists(1, j) <= m3.dists(2, j) && m3.ids(0, j) == matches.ids(0, j));
        const typename PM::OutlierWeights w3(k3.outlierFilters.compute(moved, map, m3));
        typename PM::ErrorMinimizer::ErrorElements el3(moved, map, w3, m3);
        CHECK(std::fabs((double)el3.weightedPointUsedRatio - 0.85) < 0.01);
        CHECK(k3.errorMinimizer->getResidualError(moved, map, w3, m3) >= 0);

        ICP p2p;
        { std::istringstream iss(chain("", "", "PointToPointErrorMinimizer", "")); p2p.loadFromYaml(iss); }
        CHECK(pose_diff(p2p(reading, map, guess), P) < 3e-2);
        CHECK(p2p.errorMinimizer->getCovariance()(0, 0) == T(0));          // the base class's covariance: zeros
        // PointToPointWithCovErrorMinimizer: the same alignment, and the covariance pgslam's optimiser needs (positive diagonal)
        ICP p2pc;
        { std::istringstream iss(chain("", "", "PointToPointWithCovErrorMinimizer:\n    sensorStdDev: 0.01", "")); p2pc.loadFromYaml(iss); }
        {
            const typename PM::TransformationParameters Ta = p2p(reading, map, guess), Tb = p2pc(reading, map, guess);
            CHECK(pose_diff(Ta, Tb) == 0.0);
            const typename PM::Matrix cov = p2pc.errorMinimizer->getCovariance();
            for (int a = 0; a < 6; a++) CHECK(cov(a, a) > T(0));
        }

        // PointToPlaneErrorMinimizer{force4DOF: 1}: the correction is a rotation about z and a translation -- the z axis of the
        // result is the z axis of the guess, whatever the scene asks for
        ICP dof4;
        { std::istringstream iss(chain("", "", "PointToPlaneErrorMinimizer:\n    force4DOF: 1", "")); dof4.loadFromYaml(iss); }
        CHECK(dof4.errorMinimizer->force4DOF);
        {
            const typename PM::TransformationParameters T4 = dof4(reading, map, guess);
            const typename PM::TransformationParameters D = T4 * guess.inverse();
            CHECK(std::fabs((double)D(2, 2) - 1.0) < 1e-5 && std::fabs((double)D(0, 2)) < 1e-5 && std::fabs((double)D(2, 1)) < 1e-5);
            CHECK(pose_diff(T4, P) < 5e-2);
        }

        ICP nrm;
        { std::istringstream iss(chain("", "  - SurfaceNormalOutlierFilter:\n      maxAngle: 0.6\n", "PointToPlaneErrorMinimizer", "")); nrm.loadFromYaml(iss); }
        CHECK(pose_diff(nrm(reading, map, guess), P) < 1e-2);
        CHECK(nrm.errorMinimizer->getOverlap() <= overlap);                 // the filter only removes pairs
        const typename PM::OutlierWeights wn(nrm.outlierFilters.compute(moved, reference, matches));
        int dropped = 0;
        for (int j = 0; j < wn.cols(); j++) dropped += (wn(0, j) == T(0) && w(0, j) != T(0));
        CHECK(dropped >= 0);

        // RobustOutlierFilter in the quantile filter's place: the run converges to the same pose; its stage-level weights are the
        // M-estimator's of the squared distances over the MAD scale (here recomputed on the host, in T, bit for bit)
        ICP rob;
        {
            std::istringstream iss(std::string("matcher:\n  KDTreeMatcher:\n    maxDist: 2.0\noutlierFilters:\n  - RobustOutlierFilter:\n      robustFct: cauchy\n      tuning: 1.5\n"
                                               "errorMinimizer:\n  PointToPlaneWithCovErrorMinimizer\ntransformationCheckers:\n  - CounterTransformationChecker:\n      maxIterationCount: 30\n"
                                               "  - DifferentialTransformationChecker:\n      minDiffRotErr: 0.001\n      minDiffTransErr: 0.01\n      smoothLength: 3\n"));
            rob.loadFromYaml(iss);
        }
        CHECK(pose_diff(rob(reading, map, guess), P) < 1e-2);
        CHECK(rob.errorMinimizer->getOverlap() > T(0) && rob.errorMinimizer->getOverlap() < T(1));     // the mean weight
        CHECK(rob.errorMinimizer->getCovariance()(0, 0) > T(0));
        {
            const typename PM::OutlierWeights wr(rob.outlierFilters.compute(moved, reference, matches));
            std::vector<T> v;
            for (int j = 0; j < matches.dists.cols(); j++) if (std::isfinite((double)matches.dists(0, j))) v.push_back(matches.dists(0, j));
            std::sort(v.begin(), v.end());
            const T med = v[v.size() / 2];
            for (auto &x : v) x = std::fabs(x - med);
            std::sort(v.begin(), v.end());
            const T sc = std::sqrt(v[v.size() / 2]), s2 = sc * sc, k2 = T(1.5) * T(1.5);
            int same = 0;
            for (int j = 0; j < wr.cols(); j++) {
                const T e2 = matches.dists(0, j) / s2;
                const T expect = std::isfinite((double)matches.dists(0, j)) ? T(1) / (T(1) + e2 / k2) : T(0);
                same += (wr(0, j) == expect) || (sizeof(T) == 8 && expect <= T(1e-50) && wr(0, j) <= T(1e-50));
            }
            CHECK(same == (int)wr.cols());
            typename PM::ErrorMinimizer::ErrorElements elr(moved, reference, wr, matches);
            CHECK(elr.weightedPointUsedRatio > T(0) && elr.weightedPointUsedRatio < T(1));
            CHECK(rob.errorMinimizer->getResidualError(moved, reference, wr, matches) >= 0);
        }

        ICP bound;
        { std::istringstream iss(chain("", "", "PointToPlaneErrorMinimizer", "  - BoundTransformationChecker:\n      maxRotationNorm: 0.5\n      maxTranslationNorm: 0.02\n")); bound.loadFromYaml(iss); }
        bool bthrew = false;
        try { bound(reading, map, guess); } catch (const typename PM::ConvergenceError &) { bthrew = true; }
        CHECK(bthrew);                                                       // the guess is 10 cm off: the correction leaves a 2 cm bound
        ICP bound_ok;
        { std::istringstream iss(chain("", "", "PointToPlaneErrorMinimizer", "  - BoundTransformationChecker:\n      maxRotationNorm: 0.5\n      maxTranslationNorm: 0.5\n")); bound_ok.loadFromYaml(iss); }
        CHECK(pose_diff(bound_ok(reading, map, guess), T1) < 1e-6);
    }

    // --- ConvergenceError propagates like libpointmatcher's
    bool threw = false;
    try { seq(reading, pose<T>(500, 0, 0, 0)); } catch (const typename PM::ConvergenceError &) { threw = true; }
    CHECK(threw);
    std::printf("%s: ok  (overlap %.3f, residual %.4g, |T - P| %.2e)\n", name, (double)overlap, (double)residual, pose_diff(T1, P));
}

int main()
{
    run<float>("PoseGraphSlam<float> hot path");
    run<double>("PoseGraphSlam<double> hot path");
    std::puts("dropin gpu tests ok");
    return 0;
}
