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
// staged in LDS once; each lane produces 14 consecutive outputs of one row, re-using a 14+k-1 register
// window per kernel row (14 ds_read_b64 per 14*k FMAs).  The LDS row stride is == 32 (mod 64) dwords, the
// only residue for which the 32 lanes of a ds_read_b64 group (16 column groups x 2 rows, 14-dword pitch)
// fall on distinct bank pairs.  Weights are wave-uniform -> scalar loads.
#include <stdlib.h>
#include "ta_common.h"

namespace ta {

constexpr int kConvTH = 16;        // output rows per workgroup
constexpr int kConvPT = 14;        // outputs per lane
constexpr int kConvXG = 16;        // lanes across a row
constexpr int kConvTW = kConvPT * kConvXG;   // 224

constexpr int kTimVariantDefault = 1;           // pipelined: +9 % at N=32, equal at N=160 (profiles/r01/tim_variants.txt)

constexpr int conv_lds_stride(int k) {
    int s = kConvTW + k - 1;
    while (s % 64 != 32) ++s;
    return s;
}

typedef float v2f __attribute__((ext_vector_type(2)));     // one v_pk_fma_f32 operand: an even-aligned register pair

template <int K, int TH, bool FAST_LOAD, bool PIPELINED, bool PAIRS = false>
__global__ __launch_bounds__(TH * kConvXG) void dwconv_same_kernel(const float* __restrict__ in,
                                                             float* __restrict__ out,
                                                             const float* __restrict__ w, int h, int wd,
                                                             int tiles_x, int tiles_y, int xcd_major) {
    constexpr int LO = (K - 1) / 2;
    constexpr int LW = kConvTW + K - 1;
    constexpr int LH = TH + K - 1;
    constexpr int NT = TH * kConvXG;                   // lanes of the workgroup
    constexpr int LS = conv_lds_stride(K);
    __shared__ __attribute__((aligned(16))) float tile[LH * LS];

    const int tiles = tiles_x * tiles_y;
    const unsigned tid = tile_id(xcd_major);
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
        for (int idx = threadIdx.x; idx < LH * (K - 1); idx += NT) {     // zero the two halo strips
            const int r = idx / (K - 1), c = idx - r * (K - 1);
            tile[r * LS + (c < LO ? c : kConvTW + c)] = 0.0f;
        }
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j) {
            const int idx = j * NT + threadIdx.x;
            const int r = idx / Q, q = idx - r * Q;
            if (r < LH) {
                float* dst = &tile[r * LS + LO + q * 4];
                dst[0] = v[j].x; dst[1] = v[j].y; dst[2] = v[j].z; dst[3] = v[j].w;
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < LH * LW; idx += NT) {
            const int r = idx / LW, c = idx - r * LW;
            const int gy = y0 + r - LO, gx = x0 + c - LO;
            float v = 0.0f;
            if (gy >= 0 && gy < h && gx >= 0 && gx < wd) v = ip[static_cast<int64_t>(gy) * wd + gx];
            tile[r * LS + c] = v;
        }
    }
    __syncthreads();

    const int xg = threadIdx.x % kConvXG;
    const int row = threadIdx.x / kConvXG;
    float acc[kConvPT];
    if constexpr (PAIRS) {
        // Explicit register pairs (TA_TIM_VARIANT=3).  The packed FMA reads EVEN-aligned register pairs, so with the
        // outputs paired (2c, 2c+1) the window pair (win[2c+kx], win[2c+kx+1]) is aligned for even kx only; left to
        // itself the compiler re-pairs with ~35 v_mov per kernel row and falls back to scalar v_fmac for the last row.
        // Here the window is held twice -- winE[j] = (win[2j], win[2j+1]) straight from the 8-byte LDS reads, and
        // winO[j] = (win[2j+1], win[2j+2]) built once per kernel row with one v_pk_mov_b32 per pair and shared by all
        // odd kx -- and every FMA of every row is a packed one: 105 + 13 VALU per kernel row instead of ~141 (+ ~98
        // for the unpacked last row).  Each accumulator still sees its taps in (ky, kx) order: bit-identical.
        static_assert(K % 2 == 1 && kConvPT % 2 == 0, "pair layout assumes an odd kernel and an even strip");
        constexpr int NE = (kConvPT + K) / 2;               // pairs covering win[0 .. PT + K - 2]
        constexpr int NC = kConvPT / 2;                     // accumulator pairs
        v2f acc2[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc2[c] = v2f{0.0f, 0.0f};
        auto load_pairs = [&](v2f (&win)[NE], int ky) {
            const v2f* lp = reinterpret_cast<const v2f*>(&tile[(row + ky) * LS + xg * kConvPT]);
#pragma unroll
            for (int j = 0; j < NE; ++j) win[j] = lp[j];
        };
        auto load_row_weights = [&](float (&wk)[K], int ky) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) wk[kx] = w[ky * K + kx];
        };
        auto fma_pairs = [&](const v2f (&winE)[NE], const float (&wk)[K]) {
            v2f winO[NE - 1];
#pragma unroll
            for (int j = 0; j < NE - 1; ++j) winO[j] = v2f{winE[j].y, winE[j + 1].x};
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const v2f wv = v2f{wk[kx], wk[kx]};
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    acc2[c] = __builtin_elementwise_fma(wv, (kx & 1) ? winO[c + kx / 2] : winE[c + kx / 2], acc2[c]);
            }
        };
        v2f win_a[NE], win_b[NE];
        float wk_a[K], wk_b[K];
        load_pairs(win_a, 0);
        load_row_weights(wk_a, 0);
#pragma unroll 1
        for (int ky = 0; ky + 1 < K; ky += 2) {
            load_pairs(win_b, ky + 1);
            load_row_weights(wk_b, ky + 1);
            fma_pairs(win_a, wk_a);
            load_pairs(win_a, ky + 2);                      // K odd: row ky + 2 <= K - 1 always exists
            load_row_weights(wk_a, ky + 2);
            fma_pairs(win_b, wk_b);
        }
        fma_pairs(win_a, wk_a);                             // the last row, packed like the others
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            acc[2 * c] = acc2[c].x;
            acc[2 * c + 1] = acc2[c].y;
        }
    } else {
#pragma unroll
    for (int r = 0; r < kConvPT; ++r) acc[r] = 0.0f;

    // Software-pipelined over kernel rows with two register sets: while row ky is being multiplied, the LDS window
    // and the 15 scalar weights of row ky+1 are already in flight, so a wave never stalls on lgkmcnt at the top of a
    // row (full unrolling would let the compiler hoist all 15 windows and spill).  Tap order is unchanged.
    constexpr int WN = kConvPT + K - 1 + 1;
    auto load_window = [&](float (&win)[WN], int ky) {
        const float2* lp = reinterpret_cast<const float2*>(&tile[(row + ky) * LS + xg * kConvPT]);
#pragma unroll
        for (int j = 0; j < (kConvPT + K) / 2; ++j) {
            const float2 v = lp[j];
            win[2 * j] = v.x;
            win[2 * j + 1] = v.y;
        }
    };
    auto load_weights = [&](float (&wk)[K], int ky) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) wk[kx] = w[ky * K + kx];
    };
    auto fma_row = [&](const float (&win)[WN], const float (&wk)[K]) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int r = 0; r < kConvPT; ++r) acc[r] = fmaf(wk[kx], win[r + kx], acc[r]);
    };
    if constexpr (PIPELINED) {
        float win_a[WN], win_b[WN], wk_a[K], wk_b[K];
        load_window(win_a, 0);
        load_weights(wk_a, 0);
#pragma unroll 1
        for (int ky = 0; ky + 1 < K; ky += 2) {
            load_window(win_b, ky + 1);
            load_weights(wk_b, ky + 1);
            fma_row(win_a, wk_a);
            if (ky + 2 < K) {
                load_window(win_a, ky + 2);
                load_weights(wk_a, ky + 2);
            }
            fma_row(win_b, wk_b);
        }
        if (K % 2 == 1) fma_row(win_a, wk_a);        // odd K: the last row was loaded by the final loop trip (or is row 0)
    } else {
#pragma unroll 1      // one kernel row at a time (full unrolling spills: the compiler hoists all 15 windows)
        for (int ky = 0; ky < K; ++ky) {
            float win[WN], wk[K];
            load_window(win, ky);
            load_weights(wk, ky);
            fma_row(win, wk);
        }
    }
    }   // !PAIRS

    const int oy = y0 + row;
    if (oy < h) {
        float* op = out + plane * static_cast<int64_t>(h) * wd + static_cast<int64_t>(oy) * wd;
        const int ox = x0 + xg * kConvPT;
#pragma unroll
        for (int r = 0; r < kConvPT; ++r)
            if (ox + r < wd) op[ox + r] = acc[r];
    }
}

// Separable form for kernels that are outer products wy (x) wx -- all three kernel types of the reference are
// (tim.py:42-66).  NOT the reference's arithmetic: oneDNN evaluates the 2-D kernel directly (the kernels above reproduce
// that chain bit for bit); this one evaluates  t = sum_kx wx[kx] * in[y][x + kx - lo]  and  out = sum_ky wy[ky] * t[y + ky - lo][x],
// each as an ascending FMA chain -- 2k instead of k*k taps, which turns the compute-bound 15 x 15 convolution into an
// HBM-bound pass.  The result differs from the direct convolution by rounding only (~1e-7 relative, inside the 1e-5
// gradient budget of BASELINE.json) and is pinned bit for bit to its own restatement in oracle/ta_oracle.c.  Opt-in.
template <int K, bool FAST_LOAD>
__global__ __launch_bounds__(kConvTH * kConvXG) void dwconv_separable_kernel(const float* __restrict__ in,
                                                                            float* __restrict__ out,
                                                                            const float* __restrict__ wy,
                                                                            const float* __restrict__ wx, int h, int wd,
                                                                            int tiles_x, int tiles_y, int xcd_major) {
    static_assert(K % 2 == 1 && kConvPT % 2 == 0, "pair layout assumes an odd kernel and an even strip");
    constexpr int TH = kConvTH;
    constexpr int LO = (K - 1) / 2;
    constexpr int LW = kConvTW + K - 1;
    constexpr int LH = TH + K - 1;
    constexpr int NT = TH * kConvXG;
    constexpr int LS = conv_lds_stride(K);
    constexpr int NE = (kConvPT + K) / 2;               // window pairs of one strip
    constexpr int NC = kConvPT / 2;                     // output pairs of one strip
    static_assert(LH - TH <= TH, "the halo rows are handled by one extra row pass");
    __shared__ __attribute__((aligned(16))) float tile[LH * LS];

    const int tiles = tiles_x * tiles_y;
    const unsigned tid = tile_id(xcd_major);
    const int64_t plane = tid / tiles;
    const int t = tid % tiles;
    const int y0 = (t / tiles_x) * TH;
    const int x0 = (t % tiles_x) * kConvTW;
    const float* ip = in + plane * static_cast<int64_t>(h) * wd;

    if (FAST_LOAD) {                                     // same staging as dwconv_same_kernel
        constexpr int Q = kConvTW / 4;
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
        for (int idx = threadIdx.x; idx < LH * (K - 1); idx += NT) {
            const int r = idx / (K - 1), c = idx - r * (K - 1);
            tile[r * LS + (c < LO ? c : kConvTW + c)] = 0.0f;
        }
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j) {
            const int idx = j * NT + threadIdx.x;
            const int r = idx / Q, q = idx - r * Q;
            if (r < LH) {
                float* dst = &tile[r * LS + LO + q * 4];
                dst[0] = v[j].x; dst[1] = v[j].y; dst[2] = v[j].z; dst[3] = v[j].w;
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < LH * LW; idx += NT) {
            const int r = idx / LW, c = idx - r * LW;
            const int gy = y0 + r - LO, gx = x0 + c - LO;
            float v = 0.0f;
            if (gy >= 0 && gy < h && gx >= 0 && gx < wd) v = ip[static_cast<int64_t>(gy) * wd + gx];
            tile[r * LS + c] = v;
        }
    }
    __syncthreads();

    const int xg = threadIdx.x % kConvXG;
    const int row = threadIdx.x / kConvXG;
    const bool second = row < LH - TH;                   // this lane also owns halo row TH + row
    // ---- horizontal pass, in place: T[r][x] replaces the input at tile[r][x] (no left-halo offset any more)
    auto load_pairs = [&](v2f (&win)[NE], int r) {
        const v2f* lp = reinterpret_cast<const v2f*>(&tile[r * LS + xg * kConvPT]);
#pragma unroll
        for (int j = 0; j < NE; ++j) win[j] = lp[j];
    };
    auto row_pass = [&](const v2f (&winE)[NE], v2f (&acc2)[NC]) {
        v2f winO[NE - 1];
#pragma unroll
        for (int j = 0; j < NE - 1; ++j) winO[j] = v2f{winE[j].y, winE[j + 1].x};
#pragma unroll
        for (int c = 0; c < NC; ++c) acc2[c] = v2f{0.0f, 0.0f};
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const v2f wv = v2f{wx[kx], wx[kx]};
#pragma unroll
            for (int c = 0; c < NC; ++c)
                acc2[c] = __builtin_elementwise_fma(wv, (kx & 1) ? winO[c + kx / 2] : winE[c + kx / 2], acc2[c]);
        }
    };
    v2f win_a[NE], win_b[NE];
    load_pairs(win_a, row);
    load_pairs(win_b, second ? TH + row : row);
    __syncthreads();                                     // every window is in registers before a row is overwritten
    {
        v2f t2[NC];
        row_pass(win_a, t2);
        v2f* dst = reinterpret_cast<v2f*>(&tile[row * LS + xg * kConvPT]);
#pragma unroll
        for (int c = 0; c < NC; ++c) dst[c] = t2[c];
        if (second) {
            row_pass(win_b, t2);
            dst = reinterpret_cast<v2f*>(&tile[(TH + row) * LS + xg * kConvPT]);
#pragma unroll
            for (int c = 0; c < NC; ++c) dst[c] = t2[c];
        }
    }
    __syncthreads();
    // ---- vertical pass: pairs of adjacent columns are aligned for every ky
    v2f acc2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc2[c] = v2f{0.0f, 0.0f};
#pragma unroll 3                         // a full unroll hoists all K x 7 LDS reads: 203 VGPRs, 2 waves per SIMD
    for (int ky = 0; ky < K; ++ky) {
        const v2f wv = v2f{wy[ky], wy[ky]};
        const v2f* lp = reinterpret_cast<const v2f*>(&tile[(row + ky) * LS + xg * kConvPT]);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc2[c] = __builtin_elementwise_fma(wv, lp[c], acc2[c]);
    }
    const int oy = y0 + row;
    if (oy < h) {
        float* op = out + plane * static_cast<int64_t>(h) * wd + static_cast<int64_t>(oy) * wd;
        const int ox = x0 + xg * kConvPT;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (ox + 2 * c < wd) op[ox + 2 * c] = acc2[c].x;
            if (ox + 2 * c + 1 < wd) op[ox + 2 * c + 1] = acc2[c].y;
        }
    }
}

// any k <= 31: weights and window in dynamic LDS, runtime loops (fallback for unusual kernel sizes)
__global__ __launch_bounds__(kBlock) void dwconv_same_generic_kernel(const float* __restrict__ in,
                                                                     float* __restrict__ out,
                                                                     const float* __restrict__ w, int k, int h,
                                                                     int wd, int tiles_x, int tiles_y, int ls,
                                                                     int xcd_major) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lo = (k - 1) / 2;
    const int lw = kConvTW + k - 1, lh = kConvTH + k - 1;
    float* tile = smem;
    float* wl = smem + lh * ls;
    const int tiles = tiles_x * tiles_y;
    const unsigned tid = tile_id(xcd_major);
    const int64_t plane = tid / tiles;
    const int t = tid % tiles;
    const int y0 = (t / tiles_x) * kConvTH;
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
    if (oy < h) {
        float* op = out + plane * static_cast<int64_t>(h) * wd + static_cast<int64_t>(oy) * wd;
        const int ox = x0 + xg * kConvPT;
#pragma unroll
        for (int r = 0; r < kConvPT; ++r)
            if (ox + r < wd) op[ox + r] = acc[r];
    }
}

}  // namespace ta

using namespace ta;

extern "C" int ta_depthwise_conv2d_same_separable(const float* in, float* out, const float* wy, const float* wx, int k,
                                                  int64_t planes, int h, int w_, void* stream) {
    TA_REQUIRE(in && out && wy && wx && in != out, "null or aliased pointers");
    TA_REQUIRE(planes > 0 && h > 0 && w_ > 0, "bad shape");
    TA_REQUIRE(k == 3 || k == 5 || k == 7 || k == 15, "separable form is built for k in {3, 5, 7, 15}, got %d", k);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tiles_x = static_cast<int>(ceil_div(w_, kConvTW));
    const int tiles_y = static_cast<int>(ceil_div(h, kConvTH));
    const int64_t blocks = planes * tiles_x * tiles_y;
    TA_REQUIRE(blocks < (1ll << 31), "too many tiles");
    const dim3 grid(static_cast<unsigned>(blocks));
    const bool fast = w_ <= kConvTW && w_ % 4 == 0 && aligned16(in) && (static_cast<int64_t>(h) * w_) % 4 == 0;
    switch (k) {
#define TA_SEP(KK)                                                                                                   \
    case KK:                                                                                                         \
        if (fast)                                                                                                    \
            hipLaunchKernelGGL((dwconv_separable_kernel<KK, true>), grid, dim3(kConvTH * kConvXG), 0, st, in, out, wy, \
                               wx, h, w_, tiles_x, tiles_y, xcd_major_tiles());                                      \
        else                                                                                                         \
            hipLaunchKernelGGL((dwconv_separable_kernel<KK, false>), grid, dim3(kConvTH * kConvXG), 0, st, in, out, wy, \
                               wx, h, w_, tiles_x, tiles_y, xcd_major_tiles());                                      \
        break;
        TA_SEP(3) TA_SEP(5) TA_SEP(7) TA_SEP(15)
#undef TA_SEP
    }
    return check_launch("depthwise_conv2d_same_separable");
}

extern "C" int ta_depthwise_conv2d_same(const float* in, float* out, const float* w, int k, int64_t planes, int h,
                                        int w_, void* stream) {
    TA_REQUIRE(in && out && w && in != out, "null or aliased pointers");
    TA_REQUIRE(k >= 1 && k <= 31 && planes > 0 && h > 0 && w_ > 0, "bad shape (k=%d)", k);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tiles_x = static_cast<int>(ceil_div(w_, kConvTW));
    const int tiles_y = static_cast<int>(ceil_div(h, kConvTH));
    const int64_t blocks = planes * tiles_x * tiles_y;
    TA_REQUIRE(blocks < (1ll << 31), "too many tiles");
    const dim3 grid(static_cast<unsigned>(blocks));
    const bool fast = w_ <= kConvTW && w_ % 4 == 0 && aligned16(in) && (static_cast<int64_t>(h) * w_) % 4 == 0;
    // TA_TIM_VARIANT (tuning knob): 0 = 16-row tiles, rolled rows; 1 = 16-row tiles, software-pipelined rows;
    // 2 = 14-row tiles (224 lanes, 32.3 KB LDS -> 5 workgroups per CU), rolled rows; 3 = variant 1 with explicit
    // register pairs (every FMA packed, one v_pk_mov per shifted window pair)
    static const int variant = []() {
        const char* e = getenv("TA_TIM_VARIANT");
        return e == nullptr ? kTimVariantDefault : atoi(e);
    }();
    switch (k) {
#define TA_CONV_LAUNCH(KK, TH, FAST, PIPE, PAIRS)                                                              \
    hipLaunchKernelGGL((dwconv_same_kernel<KK, TH, FAST, PIPE, PAIRS>), dim3(static_cast<unsigned>(planes *      \
                       tiles_x * ceil_div(h, TH))), dim3(TH * kConvXG), 0, st, in, out, w, h, w_, tiles_x,      \
                       static_cast<int>(ceil_div(h, TH)), xcd_major_tiles())
#define TA_CONV(KK)                                                                                  \
    case KK:                                                                                         \
        if (fast && variant == 3) { TA_CONV_LAUNCH(KK, 16, true, true, true); }                      \
        else if (fast && variant == 2) { TA_CONV_LAUNCH(KK, 14, true, false, false); }               \
        else if (fast && variant == 1) { TA_CONV_LAUNCH(KK, 16, true, true, false); }                \
        else if (fast) { TA_CONV_LAUNCH(KK, 16, true, false, false); }                               \
        else { TA_CONV_LAUNCH(KK, 16, false, false, false); }                                        \
        break;
        TA_CONV(3) TA_CONV(5) TA_CONV(7) TA_CONV(15)
#undef TA_CONV
#undef TA_CONV_LAUNCH
        default: {
            const int ls = conv_lds_stride(k);
            const size_t smem = sizeof(float) * (static_cast<size_t>(kConvTH + k - 1) * ls + k * k);
            hipLaunchKernelGGL(dwconv_same_generic_kernel, grid, dim3(kBlock), smem, st, in, out, w, k, h, w_, tiles_x,
                               tiles_y, ls, xcd_major_tiles());
        }
    }
    return check_launch("depthwise_conv2d_same");
}
