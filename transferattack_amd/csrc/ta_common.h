// Shared helpers for the gfx950 kernels of libta_hip.so (wave64, 256-thread workgroups).
#pragma once
#include <hip/hip_runtime.h>
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
int xcd_major_tiles();       // TA_XCD_MAJOR_TILES (tuning knob, read once): 1 = tile kernels walk the tiles XCD-major

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

// Tile kernels: which tile this workgroup works on.  The hardware deals consecutive workgroup ids round-robin over the
// 8 XCDs (each with its own L2), so neighbouring tiles of a plane -- which share halo rows -- land on 8 different L2s.
// With `xcd_major` the ids are re-read XCD-major: the workgroups of XCD k take the k-th contiguous eighth of the tiles,
// so neighbours share an L2.  A bijection on the first 8 * (n / 8) ids; the tail keeps its ids.
constexpr int kXcds = 8;
__device__ __forceinline__ unsigned tile_id(int xcd_major) {
    const unsigned id = blockIdx.x, n = gridDim.x;
    const unsigned per = n / kXcds;
    if (!xcd_major || id >= per * kXcds) return id;
    return (id % kXcds) * per + id / kXcds;
}

__device__ __forceinline__ float sign_of(float m) {   // torch.sign: NaN -> 0, +-0 -> 0
    return static_cast<float>(m > 0.0f) - static_cast<float>(m < 0.0f);
}

}  // namespace ta

#define TA_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ta::set_error(__VA_ARGS__);       \
            return TA_EINVAL;                 \
        }                                     \
    } while (0)
