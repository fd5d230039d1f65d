// d(loss)/d(image) through the stem convolution of the ImageNet CNN surrogates (ResNet, Inception-style stems of the same
// shape): 7 x 7, stride 2, padding 3, 3 -> 64 channels.  It is the last convolution of every surrogate backward, i.e. the
// producer of the gradient the attack's update consumes (transferattack/attack.py:118-122), and on PyTorch-ROCm it is the
// single most expensive kernel of a ResNet-50 iteration: MIOpen's implicit-GEMM backward-data reaches 13.5 TFLOP/s on it
// (2.18 ms of 29.3 at batch 125, profiles/r03/steady_state_b125_r3a.json) because a GEMM dimension of 3 input channels
// starves its tiles.  Here the four output phases of the stride-2 transposed convolution share one GEMM:
//
//   dx[n, c, 2i+py, 2j+px] = sum_{a,b < 4} sum_{k < 64} dy[n, i-1+a, j-1+b, k] * w[k, c, 5-2a+py, 5-2b+px]     (taps outside 0..6: 0)
//
//   M = output positions (n, i, j) on the dy grid,   K = 16 window taps x 64 channels = 1024,   N = 4 phases x 3 channels = 12 (-> 16)
//
// on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fixed-order fp32 accumulation, deterministic -- no
// atomics).  A workgroup owns 4 rows x 32 columns of positions; the 7 x 35 x 64 window of dy behind them is staged ONCE in
// LDS (row stride 68 dwords: the eight lanes of a 16-byte read group fall on 32 distinct banks), each wave owns one row =
// two 16-position tiles, and per tap reads its A operands with four ds_read_b128 per tile (a lane's sixteen k-values are
// contiguous channels: K is enumerated as channel = 16 * kk + s for MFMA step s, lane quarter kk).  The B operand -- the
// prepared weights W2[tap][s][kk][col], 64 KB, built once per model by ta_stem7s2_prepare -- is read straight from
// L2: 64 consecutive floats per MFMA, one coalesced load per wave.  dy: channels_last [N, OH, OW, 64] (or NCHW, ta_stem7s2_input_grad_nchw);
// dx: NCHW [N, 3, 2*OH, 2*OW].
// Useful work is 49/64 of the taps and 12/16 of the columns: 57 % of the MFMA slots.
#include "ta_common.h"

namespace ta {

constexpr int kStemRows = 4;                    // output-position rows per workgroup (one per wave)
constexpr int kStemCols = 32;                   // output-position columns per workgroup (two 16-row MFMA tiles per wave)
constexpr int kStemWinRows = kStemRows + 3;     // dy rows behind them
constexpr int kStemWinCols = kStemCols + 3;
constexpr int kStemLd = 68;                     // dwords per staged dy pixel (64 channels + 4: bank spread)
constexpr int kStemK = 64;                      // output channels of the convolution = K per tap

typedef float stem_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ stem_f32x4 stem_mfma(float a, float b, stem_f32x4 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#else
    return c;                                         // hipcc's host pass only parses this function
#endif
}

// W2[tap = a*4+b][s][kk][col]  <-  w[k = 16*kk + s][c][ky = 5-2a+py][kx = 5-2b+px],  col = (py*2 + px)*3 + c  (12..15: zero)
__global__ __launch_bounds__(kBlock) void stem7s2_prepare_kernel(const float* __restrict__ w, float* __restrict__ w2) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;          // one thread per W2 element: 16 * 16 * 4 * 16 = 16384
    if (idx >= 16 * 16 * 4 * 16) return;
    const int col = idx & 15, kk = (idx >> 4) & 3, s = (idx >> 6) & 15, tap = idx >> 10;
    const int a = tap >> 2, b = tap & 3;
    float v = 0.0f;
    if (col < 12) {
        const int c = col % 3, px = (col / 3) & 1, py = col / 6;
        const int ky = 5 - 2 * a + py, kx = 5 - 2 * b + px;
        if (ky >= 0 && ky < 7 && kx >= 0 && kx < 7) v = w[((16 * kk + s) * 3 + c) * 49 + ky * 7 + kx];
    }
    w2[idx] = v;
}

// SUMS: this kernel is the last writer of the gradient the update consumes, and with the surrogate's Normalize folded into
// the update (ta_mi_update_std) what it writes IS the update's operand: it then also leaves, per workgroup, the sum of
// |dx / std[c]| over the elements it stored (fixed order: lane, wave butterfly, waves in index order) -- the per-image
// sums of |g| that get_momentum's mean needs (attack.py:128), without another pass over dx.
// DY_NCHW: dy is [N, 64, OH, OW] (the layout autograd hands over on the reference-literal module path); only the staging of the
// window differs -- 4-byte loads along a row (a 35-pixel run per channel and window row) into the same pixel-major LDS window.
template <bool SUMS, bool DY_NCHW>
__global__ __launch_bounds__(kBlock) void stem7s2_input_grad_kernel(const float* __restrict__ dy, const float* __restrict__ w2,
                                                                    float* __restrict__ dx, int oh, int ow,
                                                                    const float* __restrict__ stdv, float* __restrict__ ws) {
    // 66 640 B of LDS: more than the 64 KB of every pre-gfx950 part -- this library targets MI355X (gfx950, 160 KB per CU)
    // only, as does philox.h's v_mad_u64_u32 path; the Makefile builds nothing else
    static_assert(sizeof(float) * kStemWinRows * kStemWinCols * kStemLd <= 160 * 1024, "dy window exceeds gfx950's LDS");
    __shared__ __attribute__((aligned(16))) float win[kStemWinRows * kStemWinCols * kStemLd];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j0 = static_cast<int>(blockIdx.x) * kStemCols, i0 = static_cast<int>(blockIdx.y) * kStemRows;
    const int n = static_cast<int>(blockIdx.z);
    const float* dyn = dy + static_cast<int64_t>(n) * oh * ow * kStemK;

    // -- stage the dy window (rows i0-1 .. i0+5, columns j0-1 .. j0+33, zeros outside the map)
    if (DY_NCHW) {
        const int64_t plane = static_cast<int64_t>(oh) * ow;
        for (int q = threadIdx.x; q < kStemK * kStemWinRows * kStemWinCols; q += kBlock) {
            const int wc = q % kStemWinCols, t = q / kStemWinCols;
            const int wr = t % kStemWinRows, ch = t / kStemWinRows;
            const int oy = i0 - 1 + wr, ox = j0 - 1 + wc;
            float v = 0.0f;
            if (oy >= 0 && oy < oh && ox >= 0 && ox < ow) v = dyn[ch * plane + static_cast<int64_t>(oy) * ow + ox];
            win[(wr * kStemWinCols + wc) * kStemLd + ch] = v;
        }
    } else                                                                       // channels_last: one float4 per step
    for (int q = threadIdx.x; q < kStemWinRows * kStemWinCols * (kStemK / 4); q += kBlock) {
        const int pix = q >> 4, c4 = (q & 15) * 4;
        const int wr = pix / kStemWinCols, wc = pix - wr * kStemWinCols;
        const int oy = i0 - 1 + wr, ox = j0 - 1 + wc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (oy >= 0 && oy < oh && ox >= 0 && ox < ow)
            v = *reinterpret_cast<const float4*>(dyn + (static_cast<int64_t>(oy) * ow + ox) * kStemK + c4);
        *reinterpret_cast<float4*>(win + pix * kStemLd + c4) = v;
    }
    __syncthreads();

    const int m = lane & 15, kk = lane >> 4;
    stem_f32x4 acc[2];
    acc[0] = stem_f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = stem_f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wl = w2 + lane;                                // B: 64 consecutive floats per (tap, s)
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
        const int a = tap >> 2, b = tap & 3;
        float bv[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) bv[s] = wl[(tap * 16 + s) * 64];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float* ap = win + ((wave + a) * kStemWinCols + mt * 16 + m + b) * kStemLd + kk * 16;
            float av[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(ap + 4 * q);
                av[4 * q] = t.x; av[4 * q + 1] = t.y; av[4 * q + 2] = t.z; av[4 * q + 3] = t.w;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) acc[mt] = stem_mfma(av[s], bv[s], acc[mt]);
        }
    }

    // -- D[4 * kk + r][col]: position j0 + mt*16 + 4*kk + r of row i0 + wave, column col = (py, px, c)
    const int col = lane & 15, i = i0 + wave;
    float part = 0.0f;
    if (col < 12 && i < oh) {
        const int c = col % 3, px = (col / 3) & 1, py = col / 6;
        const int h = 2 * oh, wd = 2 * ow;
        float* row = dx + ((static_cast<int64_t>(n) * 3 + c) * h + (2 * i + py)) * wd + px;
        const float sd = SUMS ? stdv[c] : 1.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + mt * 16 + 4 * kk + r;
                if (j < ow) {
                    row[2 * j] = acc[mt][r];
                    if (SUMS) part += fabsf(acc[mt][r] / sd);
                }
            }
    }
    if (SUMS) {
        __shared__ float red[kBlock / kWave];
        const float total = block_sum(part, red);
        if (threadIdx.x == 0)
            ws[(static_cast<int64_t>(n) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = total;
    }
}

}  // namespace ta

using namespace ta;

extern "C" int ta_stem7s2_prepare(const float* w, float* w2, void* stream) {
    TA_REQUIRE(w && w2 && aligned16(w2), "null or unaligned pointer");
    hipLaunchKernelGGL(stem7s2_prepare_kernel, dim3(16 * 16 * 4 * 16 / kBlock), dim3(kBlock), 0, static_cast<hipStream_t>(stream), w, w2);
    return check_launch("stem7s2_prepare");
}

extern "C" int64_t ta_stem_tiles(int oh, int ow) {
    return oh > 0 && ow > 0 ? ceil_div(ow, kStemCols) * ceil_div(oh, kStemRows) : 0;
}

static int stem_input_grad(const float* dy, const float* w2, float* dx, const float* stdv, float* ws, int64_t n, int oh, int ow,
                           bool dy_nchw, void* stream) {
    TA_REQUIRE(dy && w2 && dx && aligned16(dy) && aligned16(w2), "null or unaligned pointer");
    TA_REQUIRE((stdv == nullptr) == (ws == nullptr), "std and the sums' buffer come together");
    TA_REQUIRE(n > 0 && n <= 65535 && oh > 0 && ow > 0 && oh <= 4096 && ow <= 4096, "shape (n=%lld, oh=%d, ow=%d)", (long long)n, oh, ow);
    const dim3 grid(static_cast<unsigned>(ceil_div(ow, kStemCols)), static_cast<unsigned>(ceil_div(oh, kStemRows)),
                    static_cast<unsigned>(n));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define TA_STEM(SUMS, NCHW) hipLaunchKernelGGL((stem7s2_input_grad_kernel<SUMS, NCHW>), grid, dim3(kBlock), 0, st, dy, w2, dx, oh, ow, stdv, ws)
    if (ws != nullptr) { if (dy_nchw) TA_STEM(true, true); else TA_STEM(true, false); }
    else { if (dy_nchw) TA_STEM(false, true); else TA_STEM(false, false); }
#undef TA_STEM
    return check_launch("stem7s2_input_grad");
}

extern "C" int ta_stem7s2_input_grad(const float* dy, const float* w2, float* dx, const float* stdv, float* ws, int64_t n, int oh,
                                     int ow, void* stream) {
    return stem_input_grad(dy, w2, dx, stdv, ws, n, oh, ow, false, stream);
}

extern "C" int ta_stem7s2_input_grad_nchw(const float* dy, const float* w2, float* dx, const float* stdv, float* ws, int64_t n, int oh,
                                          int ow, void* stream) {
    return stem_input_grad(dy, w2, dx, stdv, ws, n, oh, ow, true, stream);
}
