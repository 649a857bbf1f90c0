// CPU-only checks of the C++ drop-in layer (no GPU in the build container):
// type spelling instantiates for float and double (mirrors the reference's only
// test, tests/instantiation.cpp), DataPoints / YAML / matrix semantics, and the
// "no CPU fallback" rule: constructing an ICP object without a GPU must throw.
#include "common.hpp"
#include <limits>

template <typename T>
void type_spelling()
{
    IMPORT_PGSLAM_TYPES(T)
    Keyframe kf;
    kf.id = 3; kf.T_world_kf = Matrix::Identity(4, 4); kf.optimized_T_world_kf = Matrix::Identity(4, 4);
    kf.cloud_ptr = std::make_shared<DP>();
    Constraint c;
    c.type = Constraint::kLoopConstraint; c.T_from_to = Matrix::Identity(4, 4); c.cov_from_to = CovMatrix::Identity(6, 6); c.weight = T(1);
    CHECK(c.cov_from_to.rows() == 6 && kf.cloud_ptr->getNbPoints() == 0);
    TransformationPtr t = PM::get().REG(Transformation).create("RigidTransformation");
    CHECK(t->checkParameters(Matrix::Identity(4, 4)));
    Matrix bad = Matrix::Identity(4, 4); bad(0, 0) = T(2);
    CHECK(!t->checkParameters(bad));
    DataPointsFilters none;
    DP empty;
    none.init(); none.apply(empty);
}

template <typename T>
void datapoints_semantics()
{
    using PM = PointMatcher<T>;
    using DP = typename PM::DataPoints;
    const T a_xyz[] = {0, 0, 0, 1, 0, 0}, a_n[] = {0, 0, 1, 0, 0, 1};
    const T b_xyz[] = {2, 2, 2}, b_n[] = {1, 0, 0};
    DP a = DP::fromXYZ(a_xyz, 2, a_n), b = DP::fromXYZ(b_xyz, 1, b_n);
    typename PM::Matrix extra(1, 2); extra(0, 0) = 5; extra(0, 1) = 6;
    a.addDescriptor("intensity", extra);                    // only in a -> dropped by concatenate
    CHECK(a.features.rows() == 4 && a.features(3, 1) == T(1) && a.xyzStride() == 4 && a.normalsStride() == 4);
    a.concatenate(b);
    CHECK(a.getNbPoints() == 3 && a.descriptors.rows() == 3 && a.descriptorExists("normals") && !a.descriptorExists("intensity"));
    CHECK(a.features(0, 2) == T(2) && a.getDescriptorViewByName("normals")(0, 2) == T(1));
    DP e;
    e.concatenate(b);
    CHECK(e.getNbPoints() == 1);
    // ErrorElements compaction
    typename PM::Matches m(1, 3); typename PM::OutlierWeights w(1, 3);
    m.ids(0, 0) = 2; m.ids(0, 1) = 0; m.ids(0, 2) = -1; m.dists(0, 0) = 1; m.dists(0, 1) = 2; m.dists(0, 2) = PM::Matches::InvalidDist();
    w(0, 0) = 1; w(0, 1) = 0; w(0, 2) = 0;
    typename PM::ErrorMinimizer::ErrorElements ee(a, a, w, m);
    CHECK(ee.reading.getNbPoints() == 1 && ee.reference.features(0, 0) == T(2));
    CHECK(std::fabs((double)ee.weightedPointUsedRatio - 1.0 / 3.0) < 1e-6 && ee.nbRejectedMatches == 2);
    w(0, 0) = 0;
    bool threw = false;
    try { typename PM::ErrorMinimizer::ErrorElements bad(a, a, w, m); } catch (const typename PM::ConvergenceError &) { threw = true; }
    CHECK(threw);
}

void yaml_and_matrix()
{
    auto y = pgslam_amd::yaml_lite::parse_string(kIcpYaml);
    CHECK(y.sections.at("matcher")[0].name == "KDTreeMatcher" && y.sections.at("matcher")[0].params.at("maxDist") == "2.0");
    CHECK(y.sections.at("transformationCheckers").size() == 2 && y.sections.at("transformationCheckers")[1].params.at("smoothLength") == "3");
    CHECK(y.sections.at("outlierFilters")[0].params.at("ratio") == "0.85" && y.sections.at("inspector")[0].name == "NullInspector");
    CHECK(y.sections.at("readingDataPointsFilters")[0].name == "IdentityDataPointsFilter");
    bool threw = false;
    try { pgslam_amd::yaml_lite::parse_string("matcher:\n\tKDTreeMatcher:\n"); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    auto P = pose<double>(1, 2, 3, 0.3, -0.1, 0.05);
    CHECK(pose_diff(P * P.inverse(), pgslam_amd::Mat<double>::Identity(4, 4)) < 1e-12);
    double rm[16];
    pgslam_amd::to_row_major16(P, rm);
    CHECK(rm[3] == 1.0 && rm[7] == 2.0 && rm[11] == 3.0 && pose_diff(pgslam_amd::from_row_major16<double>(rm), P) == 0.0);
    // filters from YAML (Localizer::SetInputFiltersConfig)
    std::istringstream fs("- MaxDistDataPointsFilter:\n    maxDist: 1.5\n- IdentityDataPointsFilter\n");
    PointMatcher<float>::DataPointsFilters f(fs);
    const float xyz[] = {1, 0, 0, 2, 0, 0, 0, 1, 0};
    auto dp = PointMatcher<float>::DataPoints::fromXYZ(xyz, 3);
    f.init(); f.apply(dp);
    CHECK(f.size() == 2 && dp.getNbPoints() == 2 && dp.features(1, 1) == 1.0f);
    {   // the deterministic host-side filters: bounding box, NaN removal, observation directions, normal orientation
        std::istringstream bs("- BoundingBoxDataPointsFilter:\n    xMin: -0.5\n    xMax: 0.5\n    yMin: -0.5\n    yMax: 0.5\n    zMin: -0.5\n    zMax: 0.5\n    removeInside: 1\n"
                              "- RemoveNaNDataPointsFilter\n"
                              "- ObservationDirectionDataPointsFilter:\n    x: 0\n    y: 0\n    z: 2\n"
                              "- OrientNormalsDataPointsFilter:\n    towardCenter: 1\n");
        PointMatcher<float>::DataPointsFilters g(bs);
        CHECK(g.size() == 4);
        const float nan = std::numeric_limits<float>::quiet_NaN();
        const float pts[] = {0.1f, 0.2f, 0.3f,  3, 0, 0,  nan, 1, 1,  0, 3, 0,  0.5f, 0, 0};
        auto c = PointMatcher<float>::DataPoints::fromXYZ(pts, 5);
        PointMatcher<float>::Matrix nrm(3, 5);
        for (int j = 0; j < 5; j++) { nrm(0, j) = 0; nrm(1, j) = 0; nrm(2, j) = j % 2 ? 1.0f : -1.0f; }
        c.addDescriptor("normals", nrm);
        g.apply(c);
        // point 0 is inside the box (removed), point 2 has a NaN (removed), point 4 sits ON the box (kept: strict bounds)
        CHECK(c.getNbPoints() == 3 && c.features(0, 0) == 3.0f && c.features(1, 1) == 3.0f && c.features(0, 2) == 0.5f);
        const int ro = c.getDescriptorStartingRow("observationDirections"), rn = c.getDescriptorStartingRow("normals");
        CHECK(c.descriptors(ro, 0) == -3.0f && c.descriptors(ro + 2, 0) == 2.0f && c.descriptors(ro + 1, 1) == -3.0f);
        for (int j = 0; j < 3; j++) CHECK(c.descriptors(rn + 2, j) == 1.0f);       // all normals now face the sensor at z = 2
        std::istringstream ks("- BoundingBoxDataPointsFilter:\n    removeInside: 0\n");
        PointMatcher<float>::DataPointsFilters keepIn(ks);
        auto c2 = PointMatcher<float>::DataPoints::fromXYZ(pts, 5);
        keepIn.apply(c2);
        CHECK(c2.getNbPoints() == 2 && c2.features(0, 0) == 0.1f && c2.features(0, 1) == 0.5f);   // default box is (-1, 1)^3
        std::istringstream os("- OrientNormalsDataPointsFilter\n");
        PointMatcher<float>::DataPointsFilters orient(os);
        threw = false;
        try { orient.apply(c2); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);                                                          // no normals: refused like upstream
    }
    {   // ShadowDataPointsFilter{eps}: a surface seen at a grazing angle goes -- |n^ . p^| > sin(eps) keeps
        std::istringstream ss("- ShadowDataPointsFilter:\n    eps: 0.2\n");
        PointMatcher<float>::DataPointsFilters sh(ss);
        // points on the x axis; normals: along the ray (kept), perpendicular (dropped), 0.1 rad off perpendicular (dropped:
        // sin 0.1 < sin 0.2), 0.3 rad off (kept), unnormalised and pointing away (kept: absolute value, normalised)
        const float pts[] = {2, 0, 0,  2, 0, 0,  3, 0, 0,  3, 0, 0,  5, 0, 0};
        auto c = PointMatcher<float>::DataPoints::fromXYZ(pts, 5);
        PointMatcher<float>::Matrix nrm(3, 5);
        const float nn[5][3] = {{1, 0, 0}, {0, 1, 0}, {std::sin(0.1f), std::cos(0.1f), 0}, {std::sin(0.3f), 0, std::cos(0.3f)}, {-7, 0, 1}};
        for (int j = 0; j < 5; j++) for (int a = 0; a < 3; a++) nrm(a, j) = nn[j][a];
        c.addDescriptor("normals", nrm);
        sh.apply(c);
        CHECK(c.getNbPoints() == 3 && c.features(0, 0) == 2.0f && c.features(0, 1) == 3.0f && c.features(0, 2) == 5.0f);
        CHECK(c.descriptors(c.getDescriptorStartingRow("normals") + 2, 1) == std::cos(0.3f));     // descriptors travel with their points
        auto bare = PointMatcher<float>::DataPoints::fromXYZ(pts, 5);
        bool threw2 = false;
        try { sh.apply(bare); } catch (const std::runtime_error &) { threw2 = true; }
        CHECK(threw2);                                                         // no normals: refused like upstream
    }
    {   // the sampling filters: FixStep keeps every step-th point; RandomSampling is seeded (reproducible, a different
        // sample per seed, about prob of the points) -- upstream draws from rand(): same distribution, no bit parity
        auto cloud_of = [&](int n) { std::vector<float> xyz(3 * n); for (int i = 0; i < n; i++) { xyz[3 * i] = (float)i; xyz[3 * i + 1] = 1.f; xyz[3 * i + 2] = 2.f; } return PointMatcher<float>::DataPoints::fromXYZ(xyz.data(), n, nullptr); };
        std::istringstream fs("- FixStepSamplingDataPointsFilter:\n    startStep: 4\n");
        PointMatcher<float>::DataPointsFilters fix(fs);
        auto a = cloud_of(10);
        fix.apply(a);
        CHECK(a.getNbPoints() == 3 && a.features(0, 1) == 4.f && a.features(0, 2) == 8.f);
        std::istringstream r1("- RandomSamplingDataPointsFilter:\n    prob: 0.5\n"), r2("- RandomSamplingDataPointsFilter:\n    prob: 0.5\n"),
            r3("- RandomSamplingDataPointsFilter:\n    prob: 0.5\n    seed: 7\n");
        PointMatcher<float>::DataPointsFilters f1(r1), f2(r2), f3(r3);
        auto b1 = cloud_of(4000), b2 = cloud_of(4000), b3 = cloud_of(4000);
        f1.apply(b1); f2.apply(b2); f3.apply(b3);
        CHECK(b1.getNbPoints() == b2.getNbPoints() && b1.getNbPoints() > 1800 && b1.getNbPoints() < 2200);
        for (int j = 0; j < (int)b1.getNbPoints(); j++) CHECK(b1.features(0, j) == b2.features(0, j));
        bool differs = b3.getNbPoints() != b1.getNbPoints();
        for (int j = 0; !differs && j < (int)b1.getNbPoints(); j++) differs = b1.features(0, j) != b3.features(0, j);
        CHECK(differs);
        // a varying step ([EXT] FixStepSampling.cpp: step *= stepMult after every cloud, clamped at endStep; init() starts over)
        std::istringstream vs("- FixStepSamplingDataPointsFilter:\n    startStep: 8\n    endStep: 2\n    stepMult: 0.5\n");
        PointMatcher<float>::DataPointsFilters v(vs);
        const int want[5] = {13, 25, 50, 50, 50};                              // steps 8, 4, 2, 2, 2 over 100 points
        for (int k = 0; k < 5; k++) { auto c = cloud_of(100); v.apply(c); CHECK((int)c.getNbPoints() == want[k]); }
        v.init();
        { auto c = cloud_of(100); v.apply(c); CHECK(c.getNbPoints() == 13 && c.features(0, 1) == 8.f); }
        std::istringstream bad_step("- FixStepSamplingDataPointsFilter:\n    startStep: 0\n");
        threw = false;
        try { PointMatcher<float>::DataPointsFilters w(bad_step); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
        // MaxPointCount: nothing happens up to maxCount points; beyond, about maxCount survive (the seeded sampler, prob = maxCount / N)
        std::istringstream ms("- MaxPointCountDataPointsFilter:\n    maxCount: 1000\n");
        PointMatcher<float>::DataPointsFilters mc(ms);
        auto few = cloud_of(1000), many = cloud_of(8000);
        mc.apply(few); mc.apply(many);
        CHECK(few.getNbPoints() == 1000 && many.getNbPoints() > 850 && many.getNbPoints() < 1150);
        // Min / MaxDist along one axis ([EXT] MaxDist.cpp: features(dim, i) < maxDist) and by radius (the norm in T, strict)
        std::istringstream ds("- MaxDistDataPointsFilter:\n    maxDist: 5\n    dim: 0\n- MinDistDataPointsFilter:\n    minDist: 2\n    dim: 0\n");
        PointMatcher<float>::DataPointsFilters dl(ds);
        auto line = cloud_of(10);
        dl.apply(line);
        CHECK(line.getNbPoints() == 2 && line.features(0, 0) == 3.f && line.features(0, 1) == 4.f);
        const float tri[6] = {3.f, 4.f, 0.f, 3.f, 4.f, 0.01f};              // |(3,4,0)| = 5 exactly: strict comparisons drop it both ways
        std::istringstream rs("- MaxDistDataPointsFilter:\n    maxDist: 5\n"), rs2("- MinDistDataPointsFilter:\n    minDist: 5\n");
        PointMatcher<float>::DataPointsFilters rmax(rs), rmin(rs2);
        auto t1 = PointMatcher<float>::DataPoints::fromXYZ(tri, 2, nullptr), t2 = PointMatcher<float>::DataPoints::fromXYZ(tri, 2, nullptr);
        rmax.apply(t1); rmin.apply(t2);
        CHECK(t1.getNbPoints() == 0 && t2.getNbPoints() == 1 && t2.features(2, 0) == 0.01f);
    }
    // an epsilon > 0 (libnabo's allowance for an approximate search, common in libpointmatcher's example configurations) is
    // accepted: the exact search meets it
    { std::istringstream ok("- SurfaceNormalDataPointsFilter:\n    knn: 10\n    epsilon: 3.16\n"); PointMatcher<float>::DataPointsFilters g(ok); CHECK(g.size() == 1); }
    // anything outside the supported set is refused at load time, never ignored
    for (const char *txt : {"- VoxelGridDataPointsFilter:\n    vSizeX: 0.1\n",
                            "- MaxDensityDataPointsFilter:\n    maxDensity: -1\n",
                            "- SamplingSurfaceNormalDataPointsFilter:\n    ratio: 1.5\n",
                            "- SamplingSurfaceNormalDataPointsFilter:\n    keepDensities: 1\n",
                            "- MaxDistDataPointsFilter:\n    maxDist: 5\n    dim: 3\n",
                            "- SurfaceNormalDataPointsFilter:\n    knn: 10\n    epsilon: -1\n",
                            "- SurfaceNormalDataPointsFilter:\n    knn: 64\n"}) {
        std::istringstream bad(txt);
        threw = false;
        try { PointMatcher<float>::DataPointsFilters g(bad); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
    }
}

int main()
{
    type_spelling<float>();
    type_spelling<double>();
    datapoints_semantics<float>();
    datapoints_semantics<double>();
    yaml_and_matrix();
    if (pgicp_device_count() == 0) {
        // construction touches no device (contexts are made at first use: tests/instantiation.cpp of the reference only
        // constructs); the first COMPUTE call refuses -- there is no CPU fallback
        PointMatcher<float>::ICP icp;
        std::vector<float> xyz = {0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
        auto dp = PointMatcher<float>::DataPoints::fromXYZ(xyz.data(), 3, nullptr);
        bool threw = false;
        try { icp.matcher->init(dp); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
        std::puts("no GPU: the first compute call refuses, as required");
    }
    {   // a chain whose YAML puts the random sampler into the per-iteration slot is refused (upstream resamples every
        // iteration: one fixed subsample would change overlap and covariance), the same filter as a reading filter loads
        PointMatcher<float>::ICP icp;
        std::istringstream bad("readingStepDataPointsFilters:\n  - RandomSamplingDataPointsFilter:\n      prob: 0.5\n"
                               "matcher:\n  KDTreeMatcher:\n    knn: 1\n");
        bool threw = false;
        try { icp.loadFromYaml(bad); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
        std::istringstream good("readingDataPointsFilters:\n  - RandomSamplingDataPointsFilter:\n      prob: 0.5\n"
                                "readingStepDataPointsFilters:\n  - FixStepSamplingDataPointsFilter:\n      startStep: 2\n"
                                "matcher:\n  KDTreeMatcher:\n    knn: 1\n");
        icp.loadFromYaml(good);
        CHECK(icp.readingDataPointsFilters.size() == 1 && icp.readingStepDataPointsFilters.size() == 1);
        // the quantile filter of the chain may be the median filter; two quantile filters are refused
        std::istringstream med("matcher:\n  KDTreeMatcher:\n    knn: 1\noutlierFilters:\n  - MedianDistOutlierFilter:\n      factor: 2.5\n  - MaxDistOutlierFilter:\n      maxDist: 1.0\n");
        icp.loadFromYaml(med);
        CHECK(icp.outlierFilters.size() == 2);
        CHECK(std::dynamic_pointer_cast<PointMatcher<float>::MedianDistOutlierFilter>(icp.outlierFilters[0])->factor == 2.5f);
        std::istringstream two("matcher:\n  KDTreeMatcher:\n    knn: 1\noutlierFilters:\n  - MedianDistOutlierFilter:\n      factor: 2.5\n  - TrimmedDistOutlierFilter:\n      ratio: 0.8\n");
        threw = false;
        try { icp.loadFromYaml(two); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
        // RobustOutlierFilter takes the quantile filter's place; what the device chain does not implement is refused by name
        std::istringstream rob("matcher:\n  KDTreeMatcher:\n    knn: 1\noutlierFilters:\n  - RobustOutlierFilter:\n      robustFct: huber\n      tuning: 2.0\n      approximation: 0.5\n");
        icp.loadFromYaml(rob);
        auto rb = std::dynamic_pointer_cast<PointMatcher<float>::RobustOutlierFilter>(icp.outlierFilters[0]);
        CHECK(rb && rb->fctCode() == PGICP_ROBUST_HUBER && rb->scaleCode() == PGICP_ROBUST_SCALE_MAD && rb->tuning == 2.0f && rb->approximation == 0.5f);
        for (const char *badp : {"      robustFct: lorentz\n", "      distanceType: point2plane\n", "      nbIterationForScale: 3\n", "      scaleEstimator: berg\n", "      tuning: 0\n"}) {
            std::istringstream bad2(std::string("matcher:\n  KDTreeMatcher:\n    knn: 1\noutlierFilters:\n  - RobustOutlierFilter:\n") + badp);
            threw = false;
            try { icp.loadFromYaml(bad2); } catch (const std::runtime_error &) { threw = true; }
            CHECK(threw);
        }
        std::istringstream rob2("matcher:\n  KDTreeMatcher:\n    knn: 1\noutlierFilters:\n  - RobustOutlierFilter:\n      robustFct: cauchy\n  - TrimmedDistOutlierFilter:\n      ratio: 0.8\n");
        threw = false;
        try { icp.loadFromYaml(rob2); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
    }
    {   // [EXT] ICPChaineBase::setDefault: the default data-point filters are upstream's -- RandomSampling (prob 0.75) on the reading,
        // SamplingSurfaceNormal (ratio 0.5, knn 7) on the reference, where the default point-to-plane minimiser gets its normals
        PointMatcher<float>::ICP icp;
        icp.setDefault();
        CHECK(icp.readingDataPointsFilters.size() == 1 && icp.referenceDataPointsFilters.size() == 1 && icp.readingStepDataPointsFilters.empty());
        CHECK(dynamic_cast<PointMatcher<float>::RandomSamplingDataPointsFilter *>(icp.readingDataPointsFilters[0].get()) != nullptr);
        auto *ssn = dynamic_cast<PointMatcher<float>::SamplingSurfaceNormalDataPointsFilter *>(icp.referenceDataPointsFilters[0].get());
        CHECK(ssn != nullptr && ssn->knn == 7 && ssn->ratio == 0.5f && ssn->samplingMethod == 0 && ssn->keepNormals);
        // a flat patch with a little noise: every kept point carries the patch's normal, +-z
        std::vector<float> xyz;
        unsigned lcg = 12345u;
        for (int i = 0; i < 40; i++) for (int j = 0; j < 40; j++) {
            lcg = lcg * 1664525u + 1013904223u;
            xyz.push_back(0.05f * i); xyz.push_back(0.05f * j); xyz.push_back(1.0f + 1e-4f * (float)((lcg >> 8) & 0xFF) / 255.f);
        }
        auto ref = PointMatcher<float>::DataPoints::fromXYZ(xyz.data(), 1600, nullptr);
        icp.referenceDataPointsFilters.apply(ref);
        CHECK(ref.getNbPoints() > 600 && ref.getNbPoints() < 1000 && ref.descriptorExists("normals"));
        const int rn = ref.getDescriptorStartingRow("normals");
        int good = 0;
        for (size_t j = 0; j < ref.getNbPoints(); j++) good += std::fabs(ref.descriptors(rn + 2, (int)j)) > 0.999f;
        CHECK(good == (int)ref.getNbPoints());
        // MaxDensity without the densities descriptor: refused like upstream
        std::istringstream md("- MaxDensityDataPointsFilter:\n    maxDensity: 100\n");
        PointMatcher<float>::DataPointsFilters mdf(md);
        bool refused = false;
        try { mdf.apply(ref); } catch (const std::runtime_error &) { refused = true; }
        CHECK(refused);
        std::puts("setDefault filters ok");
    }
    std::puts("dropin cpu tests ok");
    return 0;
}
