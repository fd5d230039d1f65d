// Counter-based random numbers for the device-side draws (random start, VMI / SIA noise): Philox4x32-10.
#pragma once
#include "ta_common.h"

namespace ta {

// ---- Philox4x32-10 (Salmon et al. 2011), counter = (lo32(i), hi32(i), lo32(offset), hi32(offset)) ---
// 32 x 32 -> 64-bit product in ONE instruction (v_mad_u64_u32 with a zero addend) instead of the v_mul_lo_u32 + v_mul_hi_u32
// pair the compiler emits: integer multiplies are quarter-rate on CDNA and Philox is forty of them per four outputs --
// 325 vs 408 SIMD-cycles per Philox block and wave on MI355X (tools/philox_rate.hip, profiles/r03/philox_rate_r3a.txt)
__device__ __forceinline__ void mul_wide(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t p;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(a), "v"(b) : "vcc");
    hi = static_cast<uint32_t>(p >> 32);
    lo = static_cast<uint32_t>(p);
#else
    hi = __umulhi(a, b);
    lo = a * b;
#endif
}

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mul_wide(M0, c.x, hi0, lo0);
        mul_wide(M1, c.z, hi1, lo1);
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0;
        k.y += W1;
    }
    return c;
}

// four U[-r, r) draws for the float4 group `quad` (element index / 4)
__device__ __forceinline__ float4 uniform4(uint64_t quad, uint64_t seed, uint64_t offset, float r) {
    const uint4 bits = philox4x32_10(
        make_uint4(static_cast<uint32_t>(quad), static_cast<uint32_t>(quad >> 32),
                   static_cast<uint32_t>(offset), static_cast<uint32_t>(offset >> 32)),
        make_uint2(static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32)));
    constexpr float k24 = 1.0f / 16777216.0f;
    const float two_r = 2.0f * r;
    float4 o;
    o.x = static_cast<float>(bits.x >> 8) * k24 * two_r - r;
    o.y = static_cast<float>(bits.y >> 8) * k24 * two_r - r;
    o.z = static_cast<float>(bits.z >> 8) * k24 * two_r - r;
    o.w = static_cast<float>(bits.w >> 8) * k24 * two_r - r;
    return o;
}

// one U[-r, r) draw for element `index` of a tensor: lane `index & 3` of the group `index >> 2`, so a kernel that walks the
// tensor in any order sees the same value at the same element
__device__ __forceinline__ float uniform1(uint64_t index, uint64_t seed, uint64_t offset, float r) {
    const float4 q = uniform4(index >> 2, seed, offset, r);
    const unsigned k = static_cast<unsigned>(index) & 3u;
    return k == 0u ? q.x : k == 1u ? q.y : k == 2u ? q.z : q.w;
}

}  // namespace ta
