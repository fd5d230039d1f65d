// Shared helpers for the gfx950 kernels of libta_hip.so (wave64, 256-thread workgroups).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ta_hip.h"

namespace ta {

constexpr int kBlock = 256;           // 4 wavefronts of 64
constexpr int kWave = 64;
constexpr int kVec = 4;               // fp32 per 16-byte lane access
constexpr int kUnroll = 3;            // 16-byte accesses in flight per lane and operand
constexpr int kTile = kBlock * kVec * kUnroll;   // 3072 elements per workgroup tile; 150528 = 49 tiles

void set_error(const char* fmt, ...);
int check_launch(const char* what);
int sum_order_lanes();                // 0 = the kernels' own order; 8 / 16 = ATen's cascade (ta_set_sum_order, runtime.hip)

// Launch timing (ta_timing_begin / ta_timing_end, runtime.hip): while armed, every fused update claims a pair of HIP
// events that ride on the dispatch packets of its own kernels (hipExtLaunchKernelGGL), so the elapsed time is the
// kernels' begin -> end as the command processor stamps it, without the marker packets of hipEventRecord in between.
struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};
LaunchEvents claim_launch_events();

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Deterministic 64-lane sum: fixed butterfly order, every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// Deterministic workgroup sum (4 waves): wave butterflies, then waves added in index order.
// `lds` must hold kBlock/kWave floats.  Every thread returns the total.
__device__ __forceinline__ float block_sum(float v, float* lds) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[wave] = v;
    __syncthreads();
    float t = lds[0];
#pragma unroll
    for (int w = 1; w < kBlock / kWave; ++w) t += lds[w];
    __syncthreads();
    return t;
}

__device__ __forceinline__ float sign_of(float m) {   // torch.sign: NaN -> 0, +-0 -> 0
    return static_cast<float>(m > 0.0f) - static_cast<float>(m < 0.0f);
}

}  // namespace ta

// a launch that carries timing events when it is the first (start) / last (stop) kernel of a timed call
#define TA_LAUNCH_TIMED(kernel, grid, block, st, ev_start, ev_stop, ...)                                 \
    do {                                                                                                 \
        if ((ev_start) != nullptr || (ev_stop) != nullptr)                                               \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, st, ev_start, ev_stop, 0, __VA_ARGS__);        \
        else                                                                                             \
            hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                                 \
    } while (0)

#define TA_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ta::set_error(__VA_ARGS__);       \
            return TA_EINVAL;                 \
        }                                     \
    } while (0)
