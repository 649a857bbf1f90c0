// The drop-in's own four-way instantiation test, spelled like pgslam user code: what the reference's
// tests/instantiation.cpp:4-19 does (float / double x single / multi thread), through the forwarding headers, plus the
// three-string constructors' and SetIcpConfig's error path.  Runs WITHOUT a GPU: device contexts are made at first use,
// so constructing and configuring the facades touches no device.
#include "pgslam/PoseGraphSlam.h"
#include "pgslam/PoseGraphSlamMT.h"
#include <pointmatcher/PointMatcher.h>

#include <cstdio>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <type_traits>

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

// ---- worker-level spelling (reference PoseGraphSlam.h:18-68, Localizer.h:33-37, LoopCloser.h:36-37): code that derives from the
// base, names the workers through it and configures them one by one
template <typename T>
class MySlam : public pgslam::PoseGraphSlamBase<T, pgslam::MapManager, pgslam::Localizer, pgslam::LoopCloser, pgslam::Optimizer> {
public:
    using Base = pgslam::PoseGraphSlamBase<T, pgslam::MapManager, pgslam::Localizer, pgslam::LoopCloser, pgslam::Optimizer>;
    using typename Base::Localizer;
    using typename Base::LoopCloser;
    MySlam(const std::string &f, const std::string &a, const std::string &b)
    {
        // the members have the reference's names and types
        typename Base::LocalizerPtr loc = this->localizer_ptr_;
        typename Base::LoopCloserPtr lc = this->loop_closer_ptr_;
        typename Base::MapManagerPtr mm = this->map_manager_ptr_;
        typename Base::OptimizerPtr opt = this->optimizer_ptr_;
        (void)mm; (void)opt;
        loc->SetLocalMapMaxSize(4);
        loc->SetOverlapThreshold(T(0.75));
        loc->SetMinimalOverlapThreshold(T(0.4));
        loc->SetInputFiltersConfig(f);
        loc->SetIcpConfig(a);
        lc->SetCandidateLocalMapMaxSize(5);          // (ignored, as upstream: LoopCloser.hpp:53-56)
        lc->SetTopologicalDistanceThreshold(T(10));
        lc->SetGeometricalDistanceThreshold(T(3));
        lc->SetOverlapThreshold(T(0.8));
        lc->SetResidualErrorThreshold(T(5000));
        lc->SetIcpConfig(b);
    }
};
template <typename T>
class MySlamMT : public pgslam::PoseGraphSlamMT<T> {
public:
    MySlamMT() { typename pgslam::PoseGraphSlamMT<T>::Base::LocalizerPtr loc = this->localizer_ptr_; loc->SetLocalMapMaxSize(3); }
};
static_assert(std::is_same<pgslam::PoseGraphSlam<float>, pgslam::PoseGraphSlamBase<float, pgslam::MapManager, pgslam::Localizer, pgslam::LoopCloser, pgslam::Optimizer>>::value,
              "PoseGraphSlam<T> is the alias upstream declares (PoseGraphSlam.h:63-66)");
static_assert(std::is_base_of<pgslam::PoseGraphSlamBase<double, pgslam::MapManagerMT, pgslam::LocalizerMT, pgslam::LoopCloserMT, pgslam::OptimizerMT>, pgslam::PoseGraphSlamMT<double>>::value,
              "PoseGraphSlamMT<T> derives from the base with the MT workers (PoseGraphSlamMT.h:17-20)");

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
    { MySlam<float> s1(f, a, b); MySlam<double> s2(f, a, b); MySlamMT<float> s3; }
    {   // the workers by themselves, constructed as upstream constructs them (PoseGraphSlam.hpp:13-24)
        auto mm = std::make_shared<pgslam::MapManager<float>>();
        auto opt = std::make_shared<pgslam::Optimizer<float>>(mm);
        auto lc = std::make_shared<pgslam::LoopCloser<float>>(mm, opt);
        auto loc = std::make_shared<pgslam::Localizer<float>>(mm);
        mm->SetLocalizer(loc);
        mm->SetLoopCloser(lc);
    }
    {   // an ICP chain by itself: construction and YAML loading need no device either
        PointMatcher<float>::ICP icp;
        PointMatcher<double>::ICPSequence seq;
        icp.setDefault();
    }
    std::printf("instantiation tests ok\n");
    return 0;
}
