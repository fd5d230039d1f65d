// Spectrum transform of SSM / FGSRA for gfx950 (reference: SSM.transform, input_transformation/ssm.py:41-54, and
// FGSRA's neighbour sampling, gradient/fgsra.py:125-140):
//     y = IDCT2( DCT2(x + noise) * mask )            per (n, c) plane of 224 x 224
// The reference evaluates the 2-D DCT pair as four 1-D FFT passes (Makhoul's factorisation, ~60 ATen launches per
// transform).  On MI355X the natural unit for a 224-point transform of 224 rows is the matrix core: with the DCT-II
// matrix C (C[k][m] = 2 cos(pi (2m+1) k / 2n)) and D = C^-1,
//     U = C A C^T,   y = D (U * mask) D^T,           A = x + noise
// are four dense 224^3 products per plane -- 1792 FLOP per element against 16 B of HBM traffic: MFMA-bound.
//
// One kernel, launched twice per transform (the mask product separates the two two-sided products):
//     OUT[:, cb] = ( L . ( (IN + ADD) . R^T[:, cb] ) ) * MUL[:, cb]          cb = a block of 32 columns
//   launch 1: IN = x, ADD = noise, L = R = C, MUL = mask     launch 2: IN = that, L = R = D
//   backward (the pair is linear): the same two launches with L = R = D^T, then C^T.
// A workgroup (4 waves) owns one plane and one 32-column block.  Step 1 builds P = A . R^T[:, cb] (n x 32) with
// v_mfma_f32_16x16x4_f32 -- exact fp32, a k-ordered fmaf chain -- K-chunked by 32: the n x 32 chunk of A and the 32 x 32
// chunk of R are staged in LDS (row stride 34 dwords: the 16 rows x 2 k-values a 32-lane read group touches fall on 32
// distinct banks), each wave keeps n/32 accumulator tiles (7 at n = 224: 16 x 16 each, one B operand read serves them
// all).  P goes to LDS transposed (stride n + 2: conflict-free as the B operand of step 2), step 2 streams L through the
// same chunk buffer and leaves OUT[:, cb] in the accumulators; the mask multiply is the epilogue.  Every chunk's global
// loads are issued back to back one chunk ahead (registers), so they fly while the current chunk is multiplied.  70 KB
// of LDS -> two workgroups per CU.  Results are deterministic (fixed k order, no atomics) and agree with the FFT form to fp32
// rounding (tests: <= 2e-6 of max|y|).
#include "ta_common.h"

namespace ta {

constexpr int kSpecCB = 32;                    // output columns per workgroup
constexpr int kSpecKC = 32;                    // K chunk staged in LDS
constexpr int kSpecLD = kSpecKC + 2;           // LDS row stride of a staged chunk (dwords)
constexpr int kSpecMaxN = 256;                 // plane side: a multiple of 32 up to this
constexpr int kSpecMaxTiles = kSpecMaxN / 32;  // accumulator tiles per wave

typedef float f32x4 __attribute__((ext_vector_type(4)));

// D[16x16] += A[16x4] . B[4x16]; lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15], D[4 * (l >> 4) + reg][l & 15]
__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#else
    return c;                                         // hipcc's host pass only parses this function
#endif
}

// One K chunk of an n x n row-major operand on its way to LDS: a thread owns n / 32 float4 (row q >> 3, columns
// 4 * (q & 7) .., q = thread + 256 * u).  `fetch` issues all loads of the thread back to back (one memory round trip
// per chunk, in flight while the previous chunk is multiplied); `commit` adds the second operand and writes LDS.
struct ChunkRegs {
    float4 v[kSpecMaxTiles];
    float4 a[kSpecMaxTiles];
};

template <bool HAS_ADD>
__device__ __forceinline__ void fetch_chunk(ChunkRegs& regs, const float* __restrict__ src, const float* __restrict__ add,
                                            int n, int c0) {
#pragma unroll
    for (int u = 0; u < kSpecMaxTiles; ++u) {
        const int q = threadIdx.x + kBlock * u;
        if (u < n / 32) {
            const int64_t off = static_cast<int64_t>(q >> 3) * n + c0 + (q & 7) * 4;
            regs.v[u] = *reinterpret_cast<const float4*>(src + off);
            if (HAS_ADD) regs.a[u] = *reinterpret_cast<const float4*>(add + off);
        }
    }
}

template <bool HAS_ADD>
__device__ __forceinline__ void commit_chunk(float* __restrict__ chunk, const ChunkRegs& regs, int n) {
#pragma unroll
    for (int u = 0; u < kSpecMaxTiles; ++u) {
        const int q = threadIdx.x + kBlock * u;
        if (u < n / 32) {
            float4 v = regs.v[u];
            if (HAS_ADD) { v.x += regs.a[u].x; v.y += regs.a[u].y; v.z += regs.a[u].z; v.w += regs.a[u].w; }
            float2* dst = reinterpret_cast<float2*>(chunk + (q >> 3) * kSpecLD + (q & 7) * 4);   // rows 8-byte aligned
            dst[0] = float2{v.x, v.y};
            dst[1] = float2{v.z, v.w};
        }
    }
}

template <bool HAS_ADD, bool HAS_MUL>
__global__ __launch_bounds__(kBlock) void dct_pair_kernel(const float* __restrict__ in, const float* __restrict__ add,
                                                          const float* __restrict__ mul, float* __restrict__ out,
                                                          const float* __restrict__ lmat, const float* __restrict__ rmat,
                                                          int n) {
    __shared__ __attribute__((aligned(16))) float chunk[kSpecMaxN * kSpecLD];        // A chunk (step 1) / L chunk (step 2)
    __shared__ __attribute__((aligned(16))) float rch[kSpecCB * kSpecLD];            // R chunk
    __shared__ __attribute__((aligned(16))) float pt[kSpecCB * (kSpecMaxN + 2)];     // P transposed: pt[k][r]
    const int cbs = n / kSpecCB;
    const int64_t plane = blockIdx.x / cbs;
    const int cb = blockIdx.x % cbs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int ct = wave & 1, rt0 = wave >> 1;            // this wave: column tile ct, row tiles rt0, rt0 + 2, ...
    const int cnt = n / 32;                              // row tiles per wave (n / 16 tiles, two waves per column tile)
    const int ldp = n + 2;
    const float* inp = in + plane * n * n;
    const float* addp = HAS_ADD ? add + plane * n * n : nullptr;
    const float* rrow = rmat + static_cast<int64_t>(cb * kSpecCB + (threadIdx.x >> 3)) * n + (threadIdx.x & 7) * 4;

    f32x4 acc[kSpecMaxTiles];
#pragma unroll
    for (int i = 0; i < kSpecMaxTiles; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    auto multiply = [&](const float* __restrict__ bsrc, int bstride, int b0) {
#pragma unroll
        for (int s = 0; s < kSpecKC / 4; ++s) {
            const float b = bsrc[(ct * 16 + li) * bstride + b0 + 4 * s + lk];
#pragma unroll
            for (int i = 0; i < kSpecMaxTiles; ++i)
                if (i < cnt) acc[i] = mfma_16x16x4(chunk[((rt0 + 2 * i) * 16 + li) * kSpecLD + 4 * s + lk], b, acc[i]);
        }
    };

    // ---- step 1: P[r][k] = sum_m A[r][m] * R[cb * 32 + k][m]
    ChunkRegs regs;
    float4 rreg;
    fetch_chunk<HAS_ADD>(regs, inp, addp, n, 0);
    rreg = *reinterpret_cast<const float4*>(rrow);
    for (int m0 = 0; m0 < n; m0 += kSpecKC) {
        commit_chunk<HAS_ADD>(chunk, regs, n);
        {
            float2* dst = reinterpret_cast<float2*>(rch + (threadIdx.x >> 3) * kSpecLD + (threadIdx.x & 7) * 4);
            dst[0] = float2{rreg.x, rreg.y};
            dst[1] = float2{rreg.z, rreg.w};
        }
        __syncthreads();
        if (m0 + kSpecKC < n) {                           // the next chunk's loads fly while this one is multiplied
            fetch_chunk<HAS_ADD>(regs, inp, addp, n, m0 + kSpecKC);
            rreg = *reinterpret_cast<const float4*>(rrow + m0 + kSpecKC);
        } else {
            fetch_chunk<false>(regs, lmat, nullptr, n, 0);                   // ... or the first chunk of L for step 2
        }
        multiply(rch, kSpecLD, 0);
        __syncthreads();                                  // the chunk has been consumed
    }
    // ---- P to LDS, transposed: pt[k][r]
#pragma unroll
    for (int i = 0; i < kSpecMaxTiles; ++i)
        if (i < cnt) {
            float2* dst = reinterpret_cast<float2*>(pt + (ct * 16 + li) * ldp + (rt0 + 2 * i) * 16 + lk * 4);
            dst[0] = float2{acc[i].x, acc[i].y};
            dst[1] = float2{acc[i].z, acc[i].w};
            acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    // ---- step 2: OUT[j][k] = sum_r L[j][r] * P[r][k]
    for (int r0 = 0; r0 < n; r0 += kSpecKC) {
        commit_chunk<false>(chunk, regs, n);
        __syncthreads();                                  // chunk (and, on the first trip, pt) complete
        if (r0 + kSpecKC < n) fetch_chunk<false>(regs, lmat, nullptr, n, r0 + kSpecKC);
        multiply(pt, ldp, r0);
        __syncthreads();
    }
    // ---- epilogue: (* mask), 64-byte row segments per 16 lanes
    const int k = cb * kSpecCB + ct * 16 + li;
#pragma unroll
    for (int i = 0; i < kSpecMaxTiles; ++i)
        if (i < cnt) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = (rt0 + 2 * i) * 16 + lk * 4 + reg;
                const int64_t idx = plane * n * n + static_cast<int64_t>(j) * n + k;
                float v = acc[i][reg];
                if (HAS_MUL) v *= mul[idx];
                out[idx] = v;
            }
        }
}

}  // namespace ta

using namespace ta;

extern "C" int ta_dct_pair(const float* in, const float* add, const float* mul, float* out, const float* lmat,
                           const float* rmat, int64_t planes, int n, void* stream) {
    TA_REQUIRE(in && out && lmat && rmat && in != out, "null or aliased pointers");
    TA_REQUIRE(planes > 0 && n >= 32 && n <= kSpecMaxN && n % 32 == 0, "plane side %d: a multiple of 32 up to %d", n, kSpecMaxN);
    TA_REQUIRE(aligned16(in) && aligned16(lmat) && aligned16(rmat) && (add == nullptr || aligned16(add)), "16-byte alignment");
    const int64_t blocks = planes * (n / kSpecCB);
    TA_REQUIRE(blocks < (1ll << 31), "too many tiles");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(static_cast<unsigned>(blocks));
#define TA_SPEC(HA, HM) hipLaunchKernelGGL((dct_pair_kernel<HA, HM>), grid, dim3(kBlock), 0, st, in, add, mul, out, lmat, rmat, n)
    if (add && mul) TA_SPEC(true, true);
    else if (add) TA_SPEC(true, false);
    else if (mul) TA_SPEC(false, true);
    else TA_SPEC(false, false);
#undef TA_SPEC
    return check_launch("dct_pair");
}
