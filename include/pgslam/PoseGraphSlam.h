// Forwarding header of the MI355X drop-in: `#include "pgslam/PoseGraphSlam.h"` of existing pgslam user code
// (reference src/pgslam/PoseGraphSlam.h:17-68) resolves to pgslam::PoseGraphSlam<T> of include/pgslam_amd/slam.hpp --
// same class name, constructors, SetIcpConfig / AddData / WriteGraphviz.
#ifndef PGSLAM_AMD_FORWARD_POSE_GRAPH_SLAM_H
#define PGSLAM_AMD_FORWARD_POSE_GRAPH_SLAM_H
#include "../pgslam_amd/slam.hpp"
#endif
