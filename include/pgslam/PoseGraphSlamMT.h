// Forwarding header of the MI355X drop-in: `#include "pgslam/PoseGraphSlamMT.h"` (reference src/pgslam/PoseGraphSlamMT.h:17-30)
// resolves to pgslam::PoseGraphSlamMT<T> of include/pgslam_amd/slam.hpp, derived -- as upstream -- from
// PoseGraphSlamBase<T, MapManagerMT, LocalizerMT, LoopCloserMT, OptimizerMT>: constructors, SetIcpConfig, Run, AddData.
#ifndef PGSLAM_AMD_FORWARD_POSE_GRAPH_SLAM_MT_H
#define PGSLAM_AMD_FORWARD_POSE_GRAPH_SLAM_MT_H
#include "PoseGraphSlam.h"
#endif
