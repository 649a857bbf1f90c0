// Forwarding header of the MI355X drop-in: pgslam::Types<T> and IMPORT_PGSLAM_TYPES (reference src/pgslam/types.h:13-82).
#ifndef PGSLAM_AMD_FORWARD_TYPES_H
#define PGSLAM_AMD_FORWARD_TYPES_H
#include "../pgslam_amd/pgslam.hpp"
#endif
