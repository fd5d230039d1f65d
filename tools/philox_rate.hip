// Rate of the two ways to get a 32 x 32 -> 64-bit product on gfx950 (v_mul_lo_u32 + v_mul_hi_u32 vs one v_mad_u64_u32),
// measured on the Philox4x32-10 round function itself: every thread chains R Philox blocks, nothing but ALU.
//   hipcc --offload-arch=gfx950 -O3 tools/philox_rate.hip -o tools/bin/philox_rate && tools/bin/philox_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <bool MAD64>
__device__ __forceinline__ void mul_wide(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    if (MAD64) {
        uint64_t p;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(a), "v"(b) : "vcc");
        hi = static_cast<uint32_t>(p >> 32);
        lo = static_cast<uint32_t>(p);
    } else {
        hi = __umulhi(a, b);
        lo = a * b;
    }
}

template <bool MAD64>
__device__ __forceinline__ uint4 philox(uint4 c, uint2 k) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mul_wide<MAD64>(M0, c.x, hi0, lo0);
        mul_wide<MAD64>(M1, c.z, hi1, lo1);
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0;
        k.y += W1;
    }
    return c;
}

template <bool MAD64>
__global__ __launch_bounds__(256) void chain(uint4* out, int reps) {
    uint4 c = make_uint4(blockIdx.x * 256 + threadIdx.x, 1, 2, 3);
    for (int i = 0; i < reps; ++i) c = philox<MAD64>(c, make_uint2(i, 7));
    out[blockIdx.x * 256 + threadIdx.x] = c;
}

template <bool MAD64>
static void run(const char* name, int blocks, int reps) {
    uint4* out;
    hipMalloc(&out, sizeof(uint4) * blocks * 256);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(chain<MAD64>, dim3(blocks), dim3(256), 0, 0, out, reps);
    hipEventRecord(a, 0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(chain<MAD64>, dim3(blocks), dim3(256), 0, 0, out, reps);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    uint4 first;
    hipMemcpy(&first, out, sizeof(first), hipMemcpyDeviceToHost);
    const double waves = blocks * 4.0, blocks_per_wave = reps;
    // cycles a SIMD spends per Philox block of one wave, at 2.4 GHz, 1024 SIMDs
    const double cyc = ms * 1e-3 * 2.4e9 * 1024 / (waves * blocks_per_wave);
    printf("%-28s grid %6d x256  reps %4d  %8.3f ms  %7.1f SIMD-cycles (at 2.4 GHz) per Philox block and wave   [%08x]\n", name,
           blocks, reps, ms, cyc, first.x);
    hipFree(out);
}

int main() {
    for (int blocks : {1568, 8192, 65536}) {
        run<false>("v_mul_lo + v_mul_hi", blocks, 64);
        run<true>("v_mad_u64_u32", blocks, 64);
    }
    return 0;
}
