// TIM gradient smoothing for gfx950: depthwise k x k 'same' correlation of every (n,c) plane with one
// kernel (reference: F.conv2d(grad, kernel, padding='same', groups=3), input_transformation/tim.py:72-74).
//
// Roofline: 8 B/element of HBM traffic but 2*k*k = 450 FLOP/element at k = 15 -> bound by the fp32
// VALU, not by HBM.  The tap order is the row-major FMA chain of the reference's CPU
// path (acc = fma(w[ky][kx], in[y+ky-7][x+kx-7], acc), zero padding included in the chain), so the
// result is bit-identical to it.
//
// Tiling: one workgroup = 16 rows x 224 columns of one plane (224 = the image side the path is defined
// on, utils.py:12; wider images take several column tiles).  The (16+k-1) x (224+k-1) input window is
// staged in LDS once, the interior starting at a 16-byte aligned column (ds_write_b128, conflict-free);
// each lane produces 14 consecutive outputs of TWO consecutive rows as 2 x 7 packed pairs and keeps one
// register window per LDS row: 15 aligned 8-byte LDS reads give the pairs (W[2j], W[2j+1]); the pairs
// shifted by one element, which every second tap needs (v_pk_fma_f32 reads even-aligned register pairs),
// are built once per window row with one v_pk_mov_b32 each.  Window row w serves kernel row w of the
// lane's first output row and kernel row w-1 of its second, so the 15 LDS reads and 14 moves of a window
// row are shared by 2 x 105 packed FMAs (224 moves per 3150 FMAs and lane; with one output row per lane it
// was 210 per 1575, and the kernel is bound by VALU issue: PMC in profiles/r02/).  The LDS row stride is == 32 (mod 64) dwords, the only residue
// for which the 32 lanes of a ds_read_b64 group (16 column groups x 2 rows, 14-dword pitch) fall on
// distinct bank pairs.  Weights are wave-uniform -> scalar loads.
//
// Software pipeline over the kernel rows: the window and the 15 weights of row ky+1 are requested while
// row ky is multiplied.  LDS and scalar loads share one counter (lgkmcnt) and scalar loads return out of
// order, so a wait with both in flight is always "wait for everything"; each phase therefore FIRST waits
// for everything requested a full row ago (it has long landed), THEN requests the next row, THEN
// multiplies -- no wait ever covers a request that was just issued.
//
// The kernel also emits sum|out| of its tile (ws, nullable): TIM.get_grad is the last kernel that writes
// the gradient, so the fused update takes the per-image mean|g| from these sums instead of re-reading g.
#include <stdlib.h>
#include "ta_common.h"

namespace ta {

constexpr int kConvRG = 16;        // row groups (lanes down a tile)
constexpr int kConvRPL = 2;        // output rows per lane
constexpr int kConvTH = kConvRG * kConvRPL;   // 32 output rows per workgroup
constexpr int kConvPT = 14;        // outputs per lane
constexpr int kConvXG = 16;        // lanes across a row
constexpr int kConvTW = kConvPT * kConvXG;   // 224
constexpr int kConvOff = 8;        // LDS column of the tile's first interior element (16-byte aligned, >= (k-1)/2)
constexpr int kConvLS = 288;       // LDS row stride in dwords: >= 8 + 224 + 8, == 32 (mod 64)

typedef float v2f __attribute__((ext_vector_type(2)));     // one v_pk_fma_f32 operand: an even-aligned register pair

// "everything this wave requested from LDS / scalar memory has arrived" (vmcnt and expcnt left alone)
__device__ __forceinline__ void wait_lgkm_all() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
}
// (a.y, b.x): the register pair one element to the right of a -- one v_pk_mov_b32 (left to itself the compiler
// sometimes spends two v_mov_b32 on it)
__device__ __forceinline__ v2f shifted_pair(v2f a, v2f b) {
#if defined(__HIP_DEVICE_COMPILE__)
    v2f o;
    asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(o) : "v"(a), "v"(b));
    return o;
#else
    return v2f{a.y, b.x};
#endif
}
__device__ __forceinline__ void pin_schedule() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

template <int K, bool FAST_LOAD>
__global__ __launch_bounds__(kConvRG * kConvXG) void dwconv_same_kernel(const float* __restrict__ in,
                                                                       float* __restrict__ out,
                                                                       const float* __restrict__ w,
                                                                       float* __restrict__ ws, int h, int wd,
                                                                       int tiles_x, int tiles_y) {
    static_assert(K % 2 == 1 && K <= 2 * kConvOff + 1 && kConvPT % 2 == 0, "pair layout: odd kernel, even strip");
    constexpr int TH = kConvTH;
    constexpr int LO = (K - 1) / 2;
    constexpr int LH = TH + K - 1;
    constexpr int NT = kConvRG * kConvXG;              // lanes of the workgroup
    constexpr int LS = kConvLS;
    constexpr int D = kConvOff - LO;                   // window element i = r + kx + D feeds output r at tap kx
    constexpr int J0 = D / 2;                          // first aligned pair a lane reads
    constexpr int J1 = (kConvPT + K - 2 + D) / 2;      // last one
    constexpr int NE = J1 - J0 + 1;
    constexpr int NC = kConvPT / 2;                    // accumulator pairs
    __shared__ __attribute__((aligned(16))) float tile[LH * LS];
    __shared__ float red[NT / kWave];

    const int tiles = tiles_x * tiles_y;
    const unsigned tid = blockIdx.x;
    const int64_t plane = tid / tiles;
    const int t = tid % tiles;
    const int y0 = (t / tiles_x) * TH;
    const int x0 = (t % tiles_x) * kConvTW;
    const float* ip = in + plane * static_cast<int64_t>(h) * wd;

    if (FAST_LOAD) {
        // wd <= 224, wd % 4 == 0, 16-byte aligned rows (the 224 x 224 case): the left/right halo is pure zero
        // padding, the interior is fetched with 16-byte loads -- ALL of a lane's loads are issued before the first
        // LDS write, so the window arrives in one HBM/L2 round trip instead of one per element.
        constexpr int Q = kConvTW / 4;                       // 16-byte groups per row
        constexpr int PER_LANE = (LH * Q + NT - 1) / NT;
        const int quads = wd / 4;
        float4 v[PER_LANE];
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j) {
            const int idx = j * NT + threadIdx.x;
            const int r = idx / Q, q = idx - r * Q;
            const int gy = y0 + r - LO;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < LH && q < quads && gy >= 0 && gy < h)
                v[j] = *reinterpret_cast<const float4*>(ip + static_cast<int64_t>(gy) * wd + q * 4);
        }
        for (int idx = threadIdx.x; idx < LH * 4; idx += NT) {           // zero the two 8-column halo strips
            const int r = idx >> 2, c = idx & 3;
            *reinterpret_cast<float4*>(&tile[r * LS + (c < 2 ? 4 * c : kConvOff + kConvTW + 4 * (c - 2))]) =
                make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j) {
            const int idx = j * NT + threadIdx.x;
            const int r = idx / Q, q = idx - r * Q;
            if (r < LH) *reinterpret_cast<float4*>(&tile[r * LS + kConvOff + q * 4]) = v[j];
        }
    } else {
        constexpr int LW = kConvTW + K - 1;
        for (int idx = threadIdx.x; idx < LH * LW; idx += NT) {
            const int r = idx / LW, c = idx - r * LW;
            const int gy = y0 + r - LO, gx = x0 + c - LO;
            float v = 0.0f;
            if (gy >= 0 && gy < h && gx >= 0 && gx < wd) v = ip[static_cast<int64_t>(gy) * wd + gx];
            tile[r * LS + c + D] = v;
        }
    }
    __syncthreads();

    const int xg = threadIdx.x % kConvXG;
    const int row = (threadIdx.x / kConvXG) * kConvRPL;      // the lane's first output row, relative to the tile
    v2f acc_a[NC], acc_b[NC];                               // output rows row, row + 1
#pragma unroll
    for (int c = 0; c < NC; ++c) acc_a[c] = acc_b[c] = v2f{0.0f, 0.0f};
    const v2f* lane_tile = reinterpret_cast<const v2f*>(&tile[row * LS + xg * kConvPT + 2 * J0]);
    auto load_pairs = [&](v2f (&win)[NE], int w_row) {
        const v2f* lp = lane_tile + w_row * (LS / 2);
#pragma unroll
        for (int j = 0; j < NE; ++j) win[j] = lp[j];
    };
    // weight rows ky and ky - 1 (the second only where it exists): what window row ky multiplies
    auto load_weights = [&](float (&cur)[K], float (&prev)[K], int ky) {
        const int kc = ky < K ? ky : K - 1, kp = ky > 0 ? ky - 1 : 0;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            cur[kx] = w[kc * K + kx];
            prev[kx] = w[kp * K + kx];
        }
    };
    auto shifted = [&](v2f (&winO)[NE - 1], const v2f (&winE)[NE]) {
#pragma unroll
        for (int j = 0; j < NE - 1; ++j) winO[j] = shifted_pair(winE[j], winE[j + 1]);
    };
    // every accumulator sees its taps in (ky, kx) order: bit-identical to the reference chain
    auto fma_pairs = [&](v2f (&acc)[NC], const v2f (&winE)[NE], const v2f (&winO)[NE - 1], const float (&wk)[K]) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const v2f wv = v2f{wk[kx], wk[kx]};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int start = 2 * c + kx + D;                 // window element of the pair's first half
                acc[c] = __builtin_elementwise_fma(wv, (start & 1) ? winO[(start - 1) / 2 - J0] : winE[start / 2 - J0], acc[c]);
            }
        }
    };
    v2f win_a[NE], win_b[NE], odd[NE - 1];
    float cur_a[K], prev_a[K], cur_b[K], prev_b[K];
    // window row 0: kernel row 0 of the first output row only
    load_pairs(win_a, 0);
    load_weights(cur_a, prev_a, 0);
    wait_lgkm_all();
    load_pairs(win_b, 1);
    load_weights(cur_b, prev_b, 1);
    pin_schedule();
    shifted(odd, win_a);
    fma_pairs(acc_a, win_a, odd, cur_a);
    pin_schedule();
    // window rows 1 .. K - 1, two per trip: kernel row wr of the first output row, wr - 1 of the second
#pragma unroll 1
    for (int wr = 1; wr + 1 < K; wr += 2) {
        wait_lgkm_all();                                // row wr: requested a full row ago
        load_pairs(win_a, wr + 1);
        load_weights(cur_a, prev_a, wr + 1);
        pin_schedule();
        shifted(odd, win_b);
        fma_pairs(acc_a, win_b, odd, cur_b);
        fma_pairs(acc_b, win_b, odd, prev_b);
        pin_schedule();
        wait_lgkm_all();                                // row wr + 1
        load_pairs(win_b, wr + 2);                      // K odd: wr + 2 <= K, the last window row
        load_weights(cur_b, prev_b, wr + 2);
        pin_schedule();
        shifted(odd, win_a);
        fma_pairs(acc_a, win_a, odd, cur_a);
        fma_pairs(acc_b, win_a, odd, prev_a);
        pin_schedule();
    }
    // window row K: kernel row K - 1 of the second output row only
    wait_lgkm_all();
    shifted(odd, win_b);
    fma_pairs(acc_b, win_b, odd, prev_b);

    float asum = 0.0f;
#pragma unroll
    for (int r = 0; r < kConvRPL; ++r) {
        const v2f (&acc)[NC] = r == 0 ? acc_a : acc_b;
        const int oy = y0 + row + r;
        if (oy < h) {
            float* op = out + plane * static_cast<int64_t>(h) * wd + static_cast<int64_t>(oy) * wd;
            const int ox = x0 + xg * kConvPT;
            if (FAST_LOAD) {                            // wd % 4 == 0 and aligned planes: 8-byte stores
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (ox + 2 * c < wd) {
                        *reinterpret_cast<v2f*>(op + ox + 2 * c) = acc[c];
                        asum += fabsf(acc[c].x);
                        asum += fabsf(acc[c].y);
                    }
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (ox + 2 * c < wd) { op[ox + 2 * c] = acc[c].x; asum += fabsf(acc[c].x); }
                    if (ox + 2 * c + 1 < wd) { op[ox + 2 * c + 1] = acc[c].y; asum += fabsf(acc[c].y); }
                }
            }
        }
    }
    const float total = block_sum(asum, red);
    if (ws != nullptr && threadIdx.x == 0) ws[tid] = total;
}

// any k <= 31: weights and window in dynamic LDS, runtime loops, one output row per lane (fallback for unusual kernel sizes)
constexpr int kConvGenTH = 16;
__global__ __launch_bounds__(kBlock) void dwconv_same_generic_kernel(const float* __restrict__ in,
                                                                     float* __restrict__ out,
                                                                     const float* __restrict__ w,
                                                                     float* __restrict__ ws, int k, int h,
                                                                     int wd, int tiles_x, int tiles_y, int ls) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[kBlock / kWave];
    const int lo = (k - 1) / 2;
    const int lw = kConvTW + k - 1, lh = kConvGenTH + k - 1;
    float* tile = smem;
    float* wl = smem + lh * ls;
    const int tiles = tiles_x * tiles_y;
    const unsigned tid = blockIdx.x;
    const int64_t plane = tid / tiles;
    const int t = tid % tiles;
    const int y0 = (t / tiles_x) * kConvGenTH;
    const int x0 = (t % tiles_x) * kConvTW;
    const float* ip = in + plane * static_cast<int64_t>(h) * wd;
    for (int idx = threadIdx.x; idx < k * k; idx += kBlock) wl[idx] = w[idx];
    for (int idx = threadIdx.x; idx < lh * lw; idx += kBlock) {
        const int r = idx / lw, c = idx - r * lw;
        const int gy = y0 + r - lo, gx = x0 + c - lo;
        float v = 0.0f;
        if (gy >= 0 && gy < h && gx >= 0 && gx < wd) v = ip[static_cast<int64_t>(gy) * wd + gx];
        tile[r * ls + c] = v;
    }
    __syncthreads();
    const int xg = threadIdx.x % kConvXG;
    const int row = threadIdx.x / kConvXG;
    float acc[kConvPT];
#pragma unroll
    for (int r = 0; r < kConvPT; ++r) acc[r] = 0.0f;
    for (int ky = 0; ky < k; ++ky) {
        const float* lp = &tile[(row + ky) * ls + xg * kConvPT];
        for (int kx = 0; kx < k; ++kx) {
            const float wv = wl[ky * k + kx];
#pragma unroll
            for (int r = 0; r < kConvPT; ++r) acc[r] = fmaf(wv, lp[r + kx], acc[r]);
        }
    }
    const int oy = y0 + row;
    float asum = 0.0f;
    if (oy < h) {
        float* op = out + plane * static_cast<int64_t>(h) * wd + static_cast<int64_t>(oy) * wd;
        const int ox = x0 + xg * kConvPT;
#pragma unroll
        for (int r = 0; r < kConvPT; ++r)
            if (ox + r < wd) { op[ox + r] = acc[r]; asum += fabsf(acc[r]); }
    }
    const float total = block_sum(asum, red);
    if (ws != nullptr && threadIdx.x == 0) ws[tid] = total;
}

constexpr int conv_generic_stride(int k) {
    int s = kConvTW + k - 1;
    while (s % 64 != 32) ++s;
    return s;
}

}  // namespace ta

using namespace ta;

static bool conv_fixed_size(int k) { return k == 3 || k == 5 || k == 7 || k == 15; }

extern "C" int64_t ta_conv_tiles(int k, int h, int w_) {
    if (h <= 0 || w_ <= 0) return 0;
    return ceil_div(w_, kConvTW) * ceil_div(h, conv_fixed_size(k) ? kConvTH : kConvGenTH);
}

extern "C" int ta_depthwise_conv2d_same(const float* in, float* out, const float* w, float* ws, int k, int64_t planes,
                                        int h, int w_, void* stream) {
    TA_REQUIRE(in && out && w && in != out, "null or aliased pointers");
    TA_REQUIRE(k >= 1 && k <= 31 && planes > 0 && h > 0 && w_ > 0, "bad shape (k=%d)", k);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tiles_x = static_cast<int>(ceil_div(w_, kConvTW));
    const int tiles_y = static_cast<int>(ceil_div(h, conv_fixed_size(k) ? kConvTH : kConvGenTH));
    const int64_t blocks = planes * tiles_x * tiles_y;
    TA_REQUIRE(blocks < (1ll << 31), "too many tiles");
    const dim3 grid(static_cast<unsigned>(blocks));
    const bool fast = w_ <= kConvTW && w_ % 4 == 0 && aligned16(in) && aligned16(out) &&
                      (static_cast<int64_t>(h) * w_) % 4 == 0;
    switch (k) {
#define TA_CONV(KK)                                                                                               \
    case KK:                                                                                                      \
        if (fast)                                                                                                 \
            hipLaunchKernelGGL((dwconv_same_kernel<KK, true>), grid, dim3(kConvRG * kConvXG), 0, st, in, out, w, ws, h, \
                               w_, tiles_x, tiles_y);                                                             \
        else                                                                                                      \
            hipLaunchKernelGGL((dwconv_same_kernel<KK, false>), grid, dim3(kConvRG * kConvXG), 0, st, in, out, w, ws, h, \
                               w_, tiles_x, tiles_y);                                                             \
        break;
        TA_CONV(3) TA_CONV(5) TA_CONV(7) TA_CONV(15)
#undef TA_CONV
        default: {
            const int ls = conv_generic_stride(k);
            const size_t smem = sizeof(float) * (static_cast<size_t>(kConvGenTH + k - 1) * ls + k * k);
            hipLaunchKernelGGL(dwconv_same_generic_kernel, grid, dim3(kBlock), smem, st, in, out, w, ws, k, h, w_, tiles_x,
                               tiles_y, ls);
        }
    }
    return check_launch("depthwise_conv2d_same");
}
