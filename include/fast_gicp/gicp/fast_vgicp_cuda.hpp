// <fast_gicp/gicp/fast_vgicp_cuda.hpp> -- upstream's header name, forwarding to the MI355X adapter.
// Included by Mapping/src/global_manager/include/global_manager/global_manager.h:76-81; with <repo>/include ahead
// of (or instead of) the fast_gicp catkin package on the include path the Mapping node builds against
// libmrslam_hip.so unchanged.
#pragma once
#include "fast_gicp_mrslam.hpp"
