// pm_ref_harness.cpp -- TEST INFRASTRUCTURE, never part of the product: runs the benchmark's ICP chain through a REAL,
// installed libpointmatcher (BASELINE.md section 2's probe: found at configure time -> cpu_baseline.kind "reference").
// It is compiled by oracle/ref_probe.py only when the library, libnabo, Eigen, Boost and yaml-cpp are all present on the
// box; this image has none of them, so this file has never been compiled here (ref_probe reports a failed build as
// such and the bench falls back to the port).  It calls the same public API pgslam calls: PM::ICP::loadFromYaml
// (reference src/pgslam/LoopCloser.hpp:73), PM::ICP::operator()(reading, reference, T_init) (LoopCloser.hpp:98),
// errorMinimizer->getOverlap() / getCovariance() (LoopCloser.hpp:108,331).
//
// usage: pm_ref PROBLEM.bin [repetitions]   PROBLEM.bin = int32 n, int32 m, float64 T_init[16] (row-major),
//        float32 reading[n][3], float32 ref[m][3], float32 ref_normals[m][3]
// prints one JSON line: {"T": [16 row-major], "overlap": .., "seconds": .., "cov": [36]}
#include <pointmatcher/PointMatcher.h>

#include <chrono>
#include <cstdio>
#include <sstream>
#include <vector>

typedef PointMatcher<float> PM;
typedef PM::DataPoints DP;

static const char *kChain =
    "matcher:\n  KDTreeMatcher:\n    knn: 1\n    epsilon: 0\n    maxDist: 2.0\n"
    "outlierFilters:\n  - TrimmedDistOutlierFilter:\n      ratio: 0.85\n"
    "errorMinimizer:\n  PointToPlaneWithCovErrorMinimizer:\n    sensorStdDev: 0.01\n"
    "transformationCheckers:\n  - CounterTransformationChecker:\n      maxIterationCount: 30\n"
    "  - DifferentialTransformationChecker:\n      minDiffRotErr: 0.001\n      minDiffTransErr: 0.01\n      smoothLength: 3\n"
    "inspector:\n  NullInspector\n"
    "logger:\n  NullLogger\n";

static DP cloud(const std::vector<float> &xyz, const std::vector<float> *nrm)
{
    const int n = (int)(xyz.size() / 3);
    DP::Labels fl;
    fl.push_back(DP::Label("x", 1)); fl.push_back(DP::Label("y", 1)); fl.push_back(DP::Label("z", 1)); fl.push_back(DP::Label("pad", 1));
    PM::Matrix f(4, n);
    for (int i = 0; i < n; i++) { f(0, i) = xyz[3 * i]; f(1, i) = xyz[3 * i + 1]; f(2, i) = xyz[3 * i + 2]; f(3, i) = 1.f; }
    if (!nrm) return DP(f, fl);
    DP::Labels dl;
    dl.push_back(DP::Label("normals", 3));
    PM::Matrix d(3, n);
    for (int i = 0; i < n; i++) for (int a = 0; a < 3; a++) d(a, i) = (*nrm)[3 * i + a];
    return DP(f, fl, d, dl);
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: pm_ref PROBLEM.bin [repetitions]\n"); return 2; }
    const int reps = argc > 2 ? std::atoi(argv[2]) : 1;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int nm[2];
    double Ti[16];
    if (std::fread(nm, sizeof nm, 1, f) != 1 || std::fread(Ti, sizeof Ti, 1, f) != 1) return 2;
    std::vector<float> rd((size_t)nm[0] * 3), rx((size_t)nm[1] * 3), rn((size_t)nm[1] * 3);
    if (std::fread(rd.data(), 4, rd.size(), f) != rd.size() || std::fread(rx.data(), 4, rx.size(), f) != rx.size() ||
        std::fread(rn.data(), 4, rn.size(), f) != rn.size()) return 2;
    std::fclose(f);
    const DP reading = cloud(rd, nullptr), reference = cloud(rx, &rn);
    PM::TransformationParameters T0 = PM::TransformationParameters::Identity(4, 4);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) T0(i, j) = (float)Ti[4 * i + j];
    PM::ICP icp;
    std::istringstream yaml(kChain);
    icp.loadFromYaml(yaml);
    PM::TransformationParameters T = T0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) T = icp(reading, reference, T0);
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    const PM::Matrix cov = icp.errorMinimizer->getCovariance();
    std::printf("{\"T\": [");
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) std::printf("%s%.9g", i + j ? ", " : "", (double)T(i, j));
    std::printf("], \"overlap\": %.9g, \"seconds\": %.6f, \"cov\": [", (double)icp.errorMinimizer->getOverlap(), s);
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) std::printf("%s%.9g", i + j ? ", " : "", (double)cov(i, j));
    std::printf("]}\n");
    return 0;
}
