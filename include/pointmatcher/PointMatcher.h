// Forwarding header of the MI355X drop-in: `#include <pointmatcher/PointMatcher.h>` (reference src/pgslam/types.h:7) resolves
// to the PointMatcher<T> shim of include/pgslam_amd/pointmatcher.hpp -- the ICP-chain object model in front of libpgicp.so.
#ifndef PGSLAM_AMD_FORWARD_POINTMATCHER_H
#define PGSLAM_AMD_FORWARD_POINTMATCHER_H
#include "../pgslam_amd/pointmatcher.hpp"
#endif
