// Device helpers of the update stack (update.hip).
#pragma once
#include "ta_common.h"

namespace ta {

typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int VEC> struct Pack;
template <> struct Pack<4> {
    float4 v;
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
    // streaming (non-temporal) forms: the operand is touched once per launch, do not keep it in L2 / MALL
    __device__ __forceinline__ void load_nt(const float* p) {
        const floatx4 t = __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(p));
        v = make_float4(t.x, t.y, t.z, t.w);
    }
    __device__ __forceinline__ void store_nt(float* p) const {
        floatx4 t;
        t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        __builtin_nontemporal_store(t, reinterpret_cast<floatx4*>(p));
    }
    __device__ __forceinline__ float& operator[](int i) { return (&v.x)[i]; }
    __device__ __forceinline__ float operator[](int i) const { return (&v.x)[i]; }
};
template <> struct Pack<1> {
    float v;
    __device__ __forceinline__ void load(const float* p) { v = *p; }
    __device__ __forceinline__ void store(float* p) const { *p = v; }
    __device__ __forceinline__ void load_nt(const float* p) { v = *p; }
    __device__ __forceinline__ void store_nt(float* p) const { *p = v; }
    __device__ __forceinline__ float& operator[](int) { return v; }
    __device__ __forceinline__ float operator[](int) const { return v; }
};

// Per-thread slot layout of a tile: slot u (0..SLOTS-1) of thread t starts at element
// (u*kBlock + t)*VEC of the tile; SLOTS*VEC*kBlock == kTile.
template <int VEC> struct Slots { static constexpr int n = kTile / (kBlock * VEC); };

// sum of an image's tile partials, identical in every lane of every wave (fixed order)
__device__ __forceinline__ float image_total(const float* __restrict__ ws, int64_t img, int tiles) {
    const int lane = threadIdx.x & 63;
    float t = 0.0f;
    for (int i = lane; i < tiles; i += kWave) t += ws[img * tiles + i];
    return wave_sum(t);
}

// float(k) / 255.0f for a byte k with the bits of the IEEE division (what utils.py:136's `astype(np.float32) / 255` stores):
// q = k * fl(1/255) is off by one ulp for 126 of the 256 bytes; one Newton step on the residual r = k - 255 q (exact in an
// fma) lands on the correctly rounded quotient for every byte (tests/test_kernel_logic_host.py checks all 256).
__device__ __forceinline__ float u8_to_unit(uint32_t k) {
    const float kf = static_cast<float>(k);
    const float r255 = 1.0f / 255.0f;                       // folded at compile time
    const float q = kf * r255;
    const float r = __builtin_fmaf(-q, 255.0f, kf);
    return __builtin_fmaf(r, r255, q);
}

// a value every lane holds alike, moved to a scalar register so that a branch on it is a scalar branch
__device__ __forceinline__ int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Channel of an element of an image of C planes of hw elements, for the elements of ONE workgroup tile [tile0, tile0 + TILE).
// A tile no longer than a plane holds at most two channels: the first is a wave-uniform value (ONE 32-bit scalar division per
// workgroup; the per-channel constants come in through scalar loads), an element is in the second iff it lies past the plane
// boundary -- one compare per element instead of a 64-bit division per element (which cost the Normalize kernels a third of
// their bandwidth).  Shorter planes (tiny test images) keep a 32-bit division per element.  Images of < 2^31 elements
// (host-checked).
template <int TILE>
struct TileChannels {
    unsigned hw, next;
    int c_lo, c_hi;
    bool two;
    __device__ __forceinline__ TileChannels(int64_t tile0, int64_t hw_, int64_t e) {
        hw = static_cast<unsigned>(hw_);
        c_lo = static_cast<int>(static_cast<unsigned>(tile0) / hw);        // uniform: blockIdx-derived
        const unsigned long long nx = static_cast<unsigned long long>(c_lo + 1) * hw;
        next = nx > 0xffffffffull ? 0xffffffffu : static_cast<unsigned>(nx);
        c_hi = nx < static_cast<unsigned long long>(e) ? c_lo + 1 : c_lo;   // the plane after this one, if the image has one
        two = hw_ >= TILE;
    }
    __device__ __forceinline__ int of(int64_t off) const {
        return two ? (static_cast<unsigned>(off) >= next ? c_hi : c_lo) : static_cast<int>(static_cast<unsigned>(off) / hw);
    }
    // table[channel of off] (t0 / t1 = table[c_lo] / table[c_hi], preloaded by the caller: scalar loads)
    __device__ __forceinline__ float pick(const float* __restrict__ table, float t0, float t1, int64_t off) const {
        return two ? (static_cast<unsigned>(off) >= next ? t1 : t0) : table[static_cast<unsigned>(off) / hw];
    }
};

struct StepParams {
    float decay, alpha, neg_eps, eps;
};

__device__ __forceinline__ float project(float d, float x, float neg_eps, float eps) {
    d = fminf(fmaxf(d, neg_eps), eps);          // torch.clamp(., -eps, eps)   attack.py:147
    d = fmaxf(d, 0.0f - x);                      // clamp(., img_min - x, .)    utils.py:68-69
    return fminf(d, 1.0f - x);                   // clamp(., ., img_max - x)
}

}  // namespace ta
