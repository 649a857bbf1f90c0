// The drop-in's own four-way instantiation test, spelled like pgslam user code: what the reference's
// tests/instantiation.cpp:4-19 does (float / double x single / multi thread), through the forwarding headers, plus the
// three-string constructors' and SetIcpConfig's error path.  Runs WITHOUT a GPU: device contexts are made at first use,
// so constructing and configuring the facades touches no device.
#include "pgslam/PoseGraphSlam.h"
#include "pgslam/PoseGraphSlamMT.h"
#include <pointmatcher/PointMatcher.h>

#include <cstdio>
#include <fstream>
#include <stdexcept>

static const char *kIcpYaml =
    "matcher:\n  KDTreeMatcher:\n    knn: 1\n    maxDist: 2.0\n"
    "outlierFilters:\n  - TrimmedDistOutlierFilter:\n      ratio: 0.85\n"
    "errorMinimizer:\n  PointToPlaneWithCovErrorMinimizer:\n    sensorStdDev: 0.01\n"
    "transformationCheckers:\n  - CounterTransformationChecker:\n      maxIterationCount: 30\n"
    "  - DifferentialTransformationChecker:\n      minDiffRotErr: 0.001\n      minDiffTransErr: 0.01\n      smoothLength: 3\n";
static const char *kFiltersYaml = "- MinDistDataPointsFilter:\n    minDist: 1.0\n";

template <typename SLAM>
static void three_ways(const std::string &f, const std::string &a, const std::string &b)
{
    { SLAM slam; }                                   // default construction (tests/instantiation.cpp)
    { SLAM slam(f, a, b); }                          // the three-string constructor
    { SLAM slam; slam.SetIcpConfig(f, a, b); }       // configured after construction
    bool threw = false;
    try { SLAM slam("/nonexistent/filters.yaml", a, b); } catch (const std::runtime_error &) { threw = true; }
    if (!threw) throw std::logic_error("a missing config file must throw");
}

int main()
{
    const std::string dir = "/tmp/pgslam_amd_inst_";
    const std::string f = dir + "filters.yaml", a = dir + "icp.yaml", b = dir + "lc.yaml";
    { std::ofstream(f) << kFiltersYaml; std::ofstream(a) << kIcpYaml; std::ofstream(b) << kIcpYaml; }
    three_ways<pgslam::PoseGraphSlam<float>>(f, a, b);
    three_ways<pgslam::PoseGraphSlam<double>>(f, a, b);
    three_ways<pgslam::PoseGraphSlamMT<float>>(f, a, b);
    three_ways<pgslam::PoseGraphSlamMT<double>>(f, a, b);
    {   // the MT flavour's workers start and stop without a device as long as nothing is fed
        pgslam::PoseGraphSlamMT<double> slam(f, a, b);
        slam.Run();
        slam.WaitIdle();
    }
    {   // an ICP chain by itself: construction and YAML loading need no device either
        PointMatcher<float>::ICP icp;
        PointMatcher<double>::ICPSequence seq;
        icp.setDefault();
    }
    std::printf("instantiation tests ok\n");
    return 0;
}
