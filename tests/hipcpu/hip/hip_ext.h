// TEST INFRASTRUCTURE -- host stand-in for <hip/hip_ext.h>: a launch that would carry timing events is a plain launch
#pragma once
#include "hip_runtime.h"
#define hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, ev_start, ev_stop, flags, ...) \
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__)
