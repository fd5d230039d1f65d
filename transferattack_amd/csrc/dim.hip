// DIM input diversity for gfx950 (reference: DIM.transform, input_transformation/dim.py:42-68):
//     y = bilinear_{resize->size}( zero_pad_{resize, top, left}( bilinear_{size->rnd}(x) ) )
// as ONE gather kernel forward (no rnd x rnd / resize x resize intermediates in HBM: 8 B/element) and ONE
// gather kernel backward (exact adjoint, no float atomics, 8 B/element).
//
// Numerics follow ATen's CPU kernels, which is what the reference runs (restated in oracle/ta_oracle.c):
//   taps      src = fma(in/out, dst + 0.5, -0.5) clamped at 0;  i0 = floor, i1 = min(i0+1, in-1);
//             l1 = src - i0, l0 = 1 - l1                         (UpSample.h compute_source_index_and_lambda)
//   forward   width first  a = fma(lx0, v[i0], lx1 * v[i1]), then height  fma(ly0, a, ly1 * b)
//   backward  four updates per output pixel in output order, acc = fma(ly*lx, g, acc)
//             (cpu_upsample_linear_backward).  The gather visits, for one target pixel, exactly the
//             updates that hit it, in that same order -> bit-identical accumulation.
//
// Structure (both directions): a workgroup owns a 32 x 32 tile of the result.  It first builds, in LDS, only the
// 1-D tables its tile needs (<= ~150 entries, one per lane): forward = tap pairs; backward = per target index the
// short, ordered list of source indices that hit it with their one or two weights ("hits").  Then two stages through
// LDS: the window of the intermediate (padded / rescaled) image the tile touches is evaluated once per pixel into
// LDS, and the tile is produced from that window.
#include <limits.h>
#include "ta_common.h"

namespace ta {

struct Tap {
    int i0, i1;
    float l0, l1;
};

__device__ __forceinline__ Tap make_tap(int o, int in_size, int out_size) {
    const float scale = static_cast<float>(in_size) / static_cast<float>(out_size);
    float src = fmaf(scale, static_cast<float>(o) + 0.5f, -0.5f);
    src = src < 0.0f ? 0.0f : src;
    int i0 = static_cast<int>(src);
    i0 = i0 > in_size - 1 ? in_size - 1 : i0;
    const int i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    const float l1 = fminf(fmaxf(src - static_cast<float>(i0), 0.0f), 1.0f);
    return Tap{i0, i1, 1.0f - l1, l1};
}

constexpr int kDimMaxSide = 1024;
constexpr int kDimTile = 32;            // 32 x 32 results per workgroup, 4 per lane
constexpr int kDimMaxMid = 96;          // bound on the side of the LDS-resident intermediate window (64 + 2*96 <= 256 lanes)

// ---------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(kBlock) void dim_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         int size, int resize, int rnd, int top, int left,
                                                         int tiles_per_side, int mid_cap) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Tap* t2y = reinterpret_cast<Tap*>(smem_raw);         // [32]       output row -> padded rows
    Tap* t2x = t2y + kDimTile;                           // [32]       output col -> padded cols
    Tap* t1y = t2x + kDimTile;                           // [mid_cap]  padded row -> x rows   (i0 < 0: zero padding)
    Tap* t1x = t1y + mid_cap;                            // [mid_cap]
    float* mid = reinterpret_cast<float*>(t1x + mid_cap);        // [mh][mw] window of the padded image

    const int tiles = tiles_per_side * tiles_per_side;
    const int64_t plane = blockIdx.x / tiles;
    const int t = blockIdx.x % tiles;
    const int oy0 = (t / tiles_per_side) * kDimTile, ox0 = (t % tiles_per_side) * kDimTile;
    const int oy1 = min(oy0 + kDimTile, size) - 1, ox1 = min(ox0 + kDimTile, size) - 1;
    const float* xp = x + plane * static_cast<int64_t>(size) * size;
    float* yp = y + plane * static_cast<int64_t>(size) * size;

    // window of the padded image this tile reads (taps are monotone in the output index)
    const int py_lo = make_tap(oy0, resize, size).i0, py_hi = make_tap(oy1, resize, size).i1;
    const int px_lo = make_tap(ox0, resize, size).i0, px_hi = make_tap(ox1, resize, size).i1;
    const int mh = py_hi - py_lo + 1, mw = px_hi - px_lo + 1;        // <= mid_cap (host-checked)

    {   // one table entry per lane
        const int k = threadIdx.x;
        if (k < kDimTile) {
            if (oy0 + k <= oy1) t2y[k] = make_tap(oy0 + k, resize, size);
        } else if (k < 2 * kDimTile) {
            if (ox0 + k - kDimTile <= ox1) t2x[k - kDimTile] = make_tap(ox0 + k - kDimTile, resize, size);
        } else if (k < 2 * kDimTile + mh) {
            const int ry = py_lo + (k - 2 * kDimTile) - top;
            t1y[k - 2 * kDimTile] = (ry >= 0 && ry < rnd) ? make_tap(ry, size, rnd) : Tap{-1, -1, 0.f, 0.f};
        } else if (k < 2 * kDimTile + mh + mw) {
            const int rx = px_lo + (k - 2 * kDimTile - mh) - left;
            t1x[k - 2 * kDimTile - mh] = (rx >= 0 && rx < rnd) ? make_tap(rx, size, rnd) : Tap{-1, -1, 0.f, 0.f};
        }
    }
    __syncthreads();

    // stage 1: the padded, rescaled image inside the window (zero outside the rnd x rnd patch, dim.py:65)
    for (int idx = threadIdx.x; idx < mh * mw; idx += kBlock) {
        const Tap ty = t1y[idx / mw], tx = t1x[idx % mw];
        float val = 0.0f;
        if (ty.i0 >= 0 && tx.i0 >= 0) {
            const float* r0 = xp + static_cast<int64_t>(ty.i0) * size;
            const float* r1 = xp + static_cast<int64_t>(ty.i1) * size;
            const float a = fmaf(tx.l0, r0[tx.i0], tx.l1 * r0[tx.i1]);
            const float b = fmaf(tx.l0, r1[tx.i0], tx.l1 * r1[tx.i1]);
            val = fmaf(ty.l0, a, ty.l1 * b);
        }
        mid[idx] = val;
    }
    __syncthreads();

    // stage 2: second resampling out of the LDS window
#pragma unroll
    for (int u = 0; u < kDimTile * kDimTile / kBlock; ++u) {
        const int local = u * kBlock + threadIdx.x;
        const int ly = local / kDimTile, lx = local % kDimTile;
        if (oy0 + ly > oy1 || ox0 + lx > ox1) continue;
        const Tap ty = t2y[ly], tx = t2x[lx];
        const float* m0 = mid + (ty.i0 - py_lo) * mw - px_lo;
        const float* m1 = mid + (ty.i1 - py_lo) * mw - px_lo;
        const float a = fmaf(tx.l0, m0[tx.i0], tx.l1 * m0[tx.i1]);
        const float b = fmaf(tx.l0, m1[tx.i0], tx.l1 * m1[tx.i1]);
        yp[static_cast<int64_t>(oy0 + ly) * size + ox0 + lx] = fmaf(ty.l0, a, ty.l1 * b);
    }
}

// --------------------------------------------------------------------------------------- backward
// A "hit": source index `src` of a resampling reaches the target index with weight w0 (as its i0 tap, flag bit 0)
// and / or w1 (as its i1 tap, flag bit 1).  Lists are in ascending source order = the order of ATen's scatter loop.
constexpr int kMaxHits = 8;

struct Hit {
    int src_flags;              // src | flags << 28
    float w0, w1;
};

struct HitList {
    int count;
    Hit h[kMaxHits];
};

// all sources o of the resampling (in_size -> out_size) whose taps touch target index p (p lives in "in" space)
__device__ __forceinline__ void build_hits(HitList& list, int p, int in_size, int out_size) {
    // i0(o) ~ (in/out) * (o + 0.5) - 0.5  =>  o ~ (p + 0.5) * out/in - 0.5; scan a safe neighbourhood of it
    const float inv = static_cast<float>(out_size) / static_cast<float>(in_size);
    int lo = static_cast<int>(floorf((static_cast<float>(p) - 0.5f) * inv - 0.5f)) - 1;
    int hi = static_cast<int>(ceilf((static_cast<float>(p) + 1.5f) * inv - 0.5f)) + 1;
    lo = max(lo, 0);
    hi = min(hi, out_size - 1);
    int n = 0;
    for (int o = lo; o <= hi; ++o) {
        const Tap tp = make_tap(o, in_size, out_size);
        const int flags = (tp.i0 == p ? 1 : 0) | (tp.i1 == p ? 2 : 0);
        if (flags != 0 && n < kMaxHits) {
            list.h[n].src_flags = o | (flags << 28);
            list.h[n].w0 = tp.l0;
            list.h[n].w1 = tp.l1;
            ++n;
        }
    }
    list.count = n;
}

// sum over (sy in ylist, sx in xlist, a, b) of the updates fma(wy_a * wx_b, value(sy, sx), acc), in ATen's order
template <typename Load>
__device__ __forceinline__ float gather_adjoint(const HitList& ly, const HitList& lx, Load&& value) {
    float acc = 0.0f;
    const int ny = ly.count, nx = lx.count;
    for (int i = 0; i < ny; ++i) {
        const Hit hy = ly.h[i];
        const int sy = hy.src_flags & 0x0FFFFFFF, fy = hy.src_flags >> 28;
        for (int j = 0; j < nx; ++j) {
            const Hit hx = lx.h[j];
            const int sx = hx.src_flags & 0x0FFFFFFF, fx = hx.src_flags >> 28;
            const float g = value(sy, sx);
            if (fy & 1) {
                if (fx & 1) acc = fmaf(hy.w0 * hx.w0, g, acc);
                if (fx & 2) acc = fmaf(hy.w0 * hx.w1, g, acc);
            }
            if (fy & 2) {
                if (fx & 1) acc = fmaf(hy.w1 * hx.w0, g, acc);
                if (fx & 2) acc = fmaf(hy.w1 * hx.w1, g, acc);
            }
        }
    }
    return acc;
}

__global__ __launch_bounds__(kBlock) void dim_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                         int size, int resize, int rnd, int top, int left,
                                                         int tiles_per_side, int mid_cap) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int* range = reinterpret_cast<int*>(smem_raw);                       // ry_lo, ry_hi, rx_lo, rx_hi
    HitList* by = reinterpret_cast<HitList*>(smem_raw + 16);             // [32]       x row <- rescaled rows
    HitList* bx = by + kDimTile;                                         // [32]       x col <- rescaled cols
    HitList* ay = bx + kDimTile;                                         // [mid_cap]  padded row (ry+top) <- output rows
    HitList* ax = ay + mid_cap;                                          // [mid_cap]
    float* mid = reinterpret_cast<float*>(ax + mid_cap);                 // [mh][mw]   d(rescaled) window

    const int tiles = tiles_per_side * tiles_per_side;
    const int64_t plane = blockIdx.x / tiles;
    const int t = blockIdx.x % tiles;
    const int iy0 = (t / tiles_per_side) * kDimTile, ix0 = (t % tiles_per_side) * kDimTile;
    const int iy1 = min(iy0 + kDimTile, size) - 1, ix1 = min(ix0 + kDimTile, size) - 1;
    const float* gyp = gy + plane * static_cast<int64_t>(size) * size;
    float* gxp = gx + plane * static_cast<int64_t>(size) * size;

    if (threadIdx.x == 0) { range[0] = INT_MAX; range[1] = -1; range[2] = INT_MAX; range[3] = -1; }
    __syncthreads();
    {   // hit lists of the second adjoint (x <- rescaled), one target index per lane; collect the rescaled window
        const int k = threadIdx.x;
        if (k < 2 * kDimTile) {
            const bool is_y = k < kDimTile;
            const int p = is_y ? iy0 + k : ix0 + k - kDimTile;
            HitList& list = is_y ? by[k] : bx[k - kDimTile];
            list.count = 0;
            if (p <= (is_y ? iy1 : ix1)) {
                build_hits(list, p, size, rnd);
                if (list.count > 0) {
                    atomicMin(&range[is_y ? 0 : 2], list.h[0].src_flags & 0x0FFFFFFF);
                    atomicMax(&range[is_y ? 1 : 3], list.h[list.count - 1].src_flags & 0x0FFFFFFF);
                }
            }
        }
    }
    __syncthreads();
    const int ry_lo = range[0], ry_hi = range[1], rx_lo = range[2], rx_hi = range[3];
    const int mh = ry_hi >= ry_lo ? ry_hi - ry_lo + 1 : 0, mw = rx_hi >= rx_lo ? rx_hi - rx_lo + 1 : 0;   // <= mid_cap
    {   // hit lists of the first adjoint (padded <- output) for the window's rows / cols
        const int k = threadIdx.x;
        if (k < mh) build_hits(ay[k], ry_lo + k + top, resize, size);
        else if (k < mh + mw) build_hits(ax[k - mh], rx_lo + (k - mh) + left, resize, size);
    }
    __syncthreads();

    // stage A: d(rescaled)[ry][rx] = d(padded)[ry + top][rx + left]
    for (int idx = threadIdx.x; idx < mh * mw; idx += kBlock)
        mid[idx] = gather_adjoint(ay[idx / mw], ax[idx % mw],
                                  [&](int oy, int ox) { return gyp[static_cast<int64_t>(oy) * size + ox]; });
    __syncthreads();

    // stage B: the gx tile from the LDS window
#pragma unroll
    for (int u = 0; u < kDimTile * kDimTile / kBlock; ++u) {
        const int local = u * kBlock + threadIdx.x;
        const int ly = local / kDimTile, lx = local % kDimTile;
        if (iy0 + ly > iy1 || ix0 + lx > ix1) continue;
        gxp[static_cast<int64_t>(iy0 + ly) * size + ix0 + lx] = gather_adjoint(
            by[ly], bx[lx], [&](int ry, int rx) { return mid[(ry - ry_lo) * mw + (rx - rx_lo)]; });
    }
}

}  // namespace ta

using namespace ta;

// side bound of the intermediate window of a 32-pixel tile; also validates the geometry
static int window_side(int64_t planes, int size, int resize, int rnd, int top, int left, int* side) {
    TA_REQUIRE(planes > 0 && size > 0 && size <= kDimMaxSide && resize <= kDimMaxSide, "bad shape");
    TA_REQUIRE(rnd > 0 && rnd <= resize && top >= 0 && left >= 0 && top + rnd <= resize && left + rnd <= resize,
               "geometry (rnd=%d, top=%d, left=%d) does not fit resize=%d", rnd, top, left, resize);
    const int64_t num = resize > rnd ? resize : rnd;
    *side = static_cast<int>(ceil_div(static_cast<int64_t>(kDimTile + 2) * num, size)) + 3;
    // kMaxHits = 8 hits per target covers resampling ratios up to ~2.9; the window bound is the tighter limit
    TA_REQUIRE(*side <= kDimMaxMid && 2 * rnd >= size && 2 * resize >= size,
               "resize ratio (%d, %d)/%d outside the range the fused DIM kernels support", rnd, resize, size);
    return 0;
}

extern "C" int ta_dim_fwd(const float* x, float* y, int64_t planes, int size, int resize, int rnd, int top, int left,
                          void* stream) {
    TA_REQUIRE(x && y && x != y, "null or aliased pointers");
    int side = 0;
    if (int rc = window_side(planes, size, resize, rnd, top, left, &side)) return rc;
    const int tps = static_cast<int>(ceil_div(size, kDimTile));
    const int64_t blocks = planes * tps * tps;
    TA_REQUIRE(blocks < (1ll << 31), "too many tiles");
    const size_t smem = sizeof(Tap) * (2 * kDimTile + 2 * side) + sizeof(float) * side * side;
    hipLaunchKernelGGL(dim_fwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), smem,
                       static_cast<hipStream_t>(stream), x, y, size, resize, rnd, top, left, tps, side);
    return check_launch("dim_fwd");
}

extern "C" int ta_dim_bwd(const float* gy, float* gx, int64_t planes, int size, int resize, int rnd, int top, int left,
                          void* stream) {
    TA_REQUIRE(gy && gx && gy != gx, "null or aliased pointers");
    int side = 0;
    if (int rc = window_side(planes, size, resize, rnd, top, left, &side)) return rc;
    const int tps = static_cast<int>(ceil_div(size, kDimTile));
    const int64_t blocks = planes * tps * tps;
    TA_REQUIRE(blocks < (1ll << 31), "too many tiles");
    const size_t smem = 16 + sizeof(HitList) * (2 * kDimTile + 2 * side) + sizeof(float) * side * side;
    hipLaunchKernelGGL(dim_bwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), smem,
                       static_cast<hipStream_t>(stream), gy, gx, size, resize, rnd, top, left, tps, side);
    return check_launch("dim_bwd");
}
