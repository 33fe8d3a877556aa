// oracle/ref_cuda_host/device_launch_parameters.h -- TEST INFRASTRUCTURE ONLY (see cuda_runtime.h here).
#pragma once
#include "cuda_runtime.h"
