// Forwarding header of the MI355X drop-in: `#include "pgslam/PoseGraphSlam.h"` of existing pgslam user code
// (reference src/pgslam/PoseGraphSlam.h:17-68) resolves to include/pgslam_amd/slam.hpp: pgslam::PoseGraphSlamBase<T, MapManager,
// Localizer, LoopCloser, Optimizer> with the reference's member names, and pgslam::PoseGraphSlam<T> as its alias -- same
// constructors, SetIcpConfig / AddData / WriteGraphviz; the workers pgslam::MapManager / Localizer / LoopCloser / Optimizer
// with the reference's constructors and setters (SetLocalMapMaxSize, SetInputFiltersConfig(path), SetCandidateLocalMapMaxSize ...).
#ifndef PGSLAM_AMD_FORWARD_POSE_GRAPH_SLAM_H
#define PGSLAM_AMD_FORWARD_POSE_GRAPH_SLAM_H
#include "../pgslam_amd/slam.hpp"
#endif
