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
// Two kernel families: the lane-per-column kernels (dim_fwd_lanes_kernel / dim_bwd_lanes_kernel) cover every geometry
// the reference can draw (resize <= 1.5 * size); the table-driven gathers (dim_fwd_kernel / dim_bwd_kernel) take the
// rest (resize ratios up to ~2.9).  Same arithmetic, same bits.  The backward kernels also emit the per-tile sums of
// |gx| (ws, nullable): when DIM's backward is the last kernel that writes the input gradient, the fused update reads
// them instead of running its own pass over g (update.hip, K1).
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include "ta_common.h"

namespace ta {

// Phase clock (a tuning aid, compiled in only by tools/dim_phase_clock.py with -DTA_DIM_PHASE_CLOCK; libta_hip.so is built
// without it and these macros vanish): thread 0 of every workgroup of the two lane-per-column kernels adds the shader-clock
// cycles between consecutive phase boundaries (the barriers) to a device-side table -- where a tile's ~12 000 cycles go.
#ifdef TA_DIM_PHASE_CLOCK
__device__ unsigned long long ta_dim_phase_cycles[2][8];
__device__ unsigned long long ta_dim_phase_groups[2];
#define TA_PHASE_BEGIN() unsigned long long ta_phase_t = clock64()
#define TA_PHASE(kernel, phase)                                                       \
    do {                                                                              \
        if (threadIdx.x == 0) {                                                       \
            const unsigned long long ta_phase_now = clock64();                        \
            atomicAdd(&ta_dim_phase_cycles[kernel][phase], ta_phase_now - ta_phase_t); \
            if ((phase) == 0) atomicAdd(&ta_dim_phase_groups[kernel], 1ull);           \
            ta_phase_t = ta_phase_now;                                                \
        }                                                                             \
    } while (0)
#else
#define TA_PHASE_BEGIN() do { } while (0)
#define TA_PHASE(kernel, phase) do { } while (0)
#endif

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

constexpr int kDimMaxSide = 1024;       // LDS tables are sized for sides up to this

// ---------------------------------------------------------------------------------------- forward
constexpr int kDimFwdTile = 32;         // 32 x 32 outputs per workgroup, 4 per lane
constexpr int kDimFwdMaxMid = 100;      // side of the LDS-resident window of the padded image; resize/size <= ~2.9 (OPS)

// Two stages through LDS: (1) the window of the zero-padded, rescaled image that this output tile touches is
// computed once per pixel (4 taps of x each) into LDS; (2) every output pixel blends 4 LDS values.  Each padded pixel
// is evaluated once per tile instead of up to 4 times, and no intermediate ever reaches HBM.
__global__ __launch_bounds__(kBlock) void dim_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         int size, int resize, int rnd, int top, int left,
                                                         int tiles_per_side) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Tap* t2 = reinterpret_cast<Tap*>(smem_raw);          // [size]   resize -> size   (second resample)
    Tap* t1 = t2 + size;                                 // [rnd]    size -> rnd      (first resample)
    float* mid = reinterpret_cast<float*>(t1 + rnd);     // [mh][mw] padded-image window
    for (int o = threadIdx.x; o < size; o += kBlock) t2[o] = make_tap(o, resize, size);
    for (int o = threadIdx.x; o < rnd; o += kBlock) t1[o] = make_tap(o, size, rnd);
    __syncthreads();

    const int tiles = tiles_per_side * tiles_per_side;
    const unsigned tid = blockIdx.x;
    const int64_t plane = tid / tiles;
    const int t = tid % tiles;
    const int oy0 = (t / tiles_per_side) * kDimFwdTile, ox0 = (t % tiles_per_side) * kDimFwdTile;
    const int oy1 = min(oy0 + kDimFwdTile, size) - 1, ox1 = min(ox0 + kDimFwdTile, size) - 1;
    const float* xp = x + plane * static_cast<int64_t>(size) * size;
    float* yp = y + plane * static_cast<int64_t>(size) * size;

    const int py_lo = t2[oy0].i0, py_hi = t2[oy1].i1, px_lo = t2[ox0].i0, px_hi = t2[ox1].i1;   // taps are monotone
    const int mh = py_hi - py_lo + 1, mw = px_hi - px_lo + 1;                                     // <= kDimFwdMaxMid

    for (int idx = threadIdx.x; idx < mh * mw; idx += kBlock) {
        const int py = py_lo + idx / mw, px = px_lo + idx % mw;
        const int ry = py - top, rx = px - left;
        float val = 0.0f;                                                  // the zero padding of dim.py:65
        if (ry >= 0 && ry < rnd && rx >= 0 && rx < rnd) {
            const Tap ty = t1[ry], tx = t1[rx];
            const float* r0 = xp + static_cast<int64_t>(ty.i0) * size;
            const float* r1 = xp + static_cast<int64_t>(ty.i1) * size;
            const float a = fmaf(tx.l0, r0[tx.i0], tx.l1 * r0[tx.i1]);
            const float b = fmaf(tx.l0, r1[tx.i0], tx.l1 * r1[tx.i1]);
            val = fmaf(ty.l0, a, ty.l1 * b);
        }
        mid[idx] = val;
    }
    __syncthreads();

#pragma unroll
    for (int u = 0; u < kDimFwdTile * kDimFwdTile / kBlock; ++u) {
        const int local = u * kBlock + threadIdx.x;
        const int oy = oy0 + local / kDimFwdTile, ox = ox0 + local % kDimFwdTile;
        if (oy >= size || ox >= size) continue;
        const Tap ty = t2[oy], tx = t2[ox];
        const float* m0 = mid + (ty.i0 - py_lo) * mw - px_lo;
        const float* m1 = mid + (ty.i1 - py_lo) * mw - px_lo;
        const float a = fmaf(tx.l0, m0[tx.i0], tx.l1 * m0[tx.i1]);
        const float b = fmaf(tx.l0, m1[tx.i0], tx.l1 * m1[tx.i1]);
        yp[static_cast<int64_t>(oy) * size + ox] = fmaf(ty.l0, a, ty.l1 * b);
    }
}

// Lane-per-column separable forward (bit-identical: ATen's bilinear IS "width first, then height"), written for
// instruction count: the table-driven gather above spends ~190 instructions per output pixel (per-workgroup rebuild of all 460 taps, div/mod by the runtime window
// width, 64-bit addressing), and its measured time is what instruction issue alone predicts.  Here
//   * a tile is THO=32 output rows x `tw` (<= 64) output columns, `tw` chosen on the host so that the window of the
//     padded image behind it is at most 64 columns wide: one LANE per window column, all four passes run over rows;
//   * every lane builds only its own two column taps (registers) and at most one row tap (LDS); the two divisions
//     in/out are done once on the host (same IEEE single division);
//   * the x rows behind the window are fetched with all loads of a lane issued back to back (RPW rows per wave).
// Arithmetic and rounding order are ATen's (width first, then height) -> bit-identical to dim_fwd_kernel.
//   H1  T[r][c]   = fma(lx0, x[r][i0], lx1 * x[r][i1])          rows of x behind the window, window columns
//   V1  mid[p][c] = fma(ly0, T[i0][c], ly1 * T[i1][c])  (or 0)   the zero-padded, rescaled window
//   H2  u[p][ox]  = fma(lx0, mid[p][i0], lx1 * mid[p][i1])       window rows, the tile's output columns
//   V2  y[oy][ox] = fma(ly0, u[i0][ox], ly1 * u[i1][ox])
constexpr int kDimLaneRows = 32;        // THO

__device__ __forceinline__ Tap make_tap_scaled(int o, int in_size, float scale) {
    float src = fmaf(scale, static_cast<float>(o) + 0.5f, -0.5f);
    src = src < 0.0f ? 0.0f : src;
    int i0 = static_cast<int>(src);
    i0 = i0 > in_size - 1 ? in_size - 1 : i0;
    const int i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    const float l1 = fminf(fmaxf(src - static_cast<float>(i0), 0.0f), 1.0f);
    return Tap{i0, i1, 1.0f - l1, l1};
}

template <int RPW>                      // rows per wave of the LDS rectangles: ROWS = 4 * RPW
__global__ __launch_bounds__(kBlock) void dim_fwd_lanes_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int size, int resize, int rnd, int top, int left,
                                                               float scale1, float scale2, int tw, int tiles_x,
                                                               int tiles_y) {
    constexpr int ROWS = 4 * RPW;
    TA_PHASE_BEGIN();
    __shared__ __attribute__((aligned(16))) Tap ty2[kDimLaneRows];     // output row  -> padded rows
    __shared__ __attribute__((aligned(16))) Tap ty1[ROWS];             // window row  -> x rows (valid rows only)
    __shared__ int corner[2];                                          // px_lo, px_hi
    __shared__ __attribute__((aligned(16))) float T[ROWS * 64];        // H1 result; re-used as `u` by H2 / V2
    __shared__ __attribute__((aligned(16))) float mid[ROWS * 64];      // the zero-padded, rescaled window

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: row logic runs on the SALU
    const int tiles = tiles_x * tiles_y;
    const int tid = static_cast<int>(blockIdx.x);
    const int plane = tid / tiles;                                     // grid < 2^31 (host-checked)
    const int t = tid - plane * tiles;
    const int tyi = t / tiles_x;
    const int oy0 = tyi * kDimLaneRows, ox0 = (t - tyi * tiles_x) * tw;
    const int th = min(kDimLaneRows, size - oy0), twc = min(tw, size - ox0);    // rows / columns of this tile
    const float* xp = x + static_cast<int64_t>(plane) * size * size;
    float* yp = y + static_cast<int64_t>(plane) * size * size;

    // -- taps that do not depend on the window origin
    Tap tx2{0, 0, 0.f, 0.f};
    if (lane < twc) tx2 = make_tap_scaled(ox0 + lane, resize, scale2);
    if (threadIdx.x < th) ty2[threadIdx.x] = make_tap_scaled(oy0 + threadIdx.x, resize, scale2);
    if (threadIdx.x == 64) corner[0] = tx2.i0;                         // lane 0 of wave 1
    if (threadIdx.x == 64 + twc - 1) corner[1] = tx2.i1;
    __syncthreads();
    TA_PHASE(0, 0);
    const int py_lo = ty2[0].i0, py_hi = ty2[th - 1].i1, px_lo = corner[0], px_hi = corner[1];
    const int mh = py_hi - py_lo + 1, mw = px_hi - px_lo + 1;          // <= ROWS - 1, <= 64 (host-checked)
    const int p_a = max(top - py_lo, 0), p_b = min(top + rnd - 1 - py_lo, mh - 1);   // window rows inside the image
    // -- taps of the first resample for this lane's window column and for the valid window rows
    const int rx = px_lo + lane - left;
    const bool col_ok = lane < mw && rx >= 0 && rx < rnd;
    Tap tx1{0, 0, 0.f, 0.f};
    if (col_ok) tx1 = make_tap_scaled(rx, size, scale1);
    {
        const int p = static_cast<int>(threadIdx.x) - 64;              // waves 1.. build the row table
        if (p >= p_a && p <= p_b) ty1[p] = make_tap_scaled(py_lo + p - top, size, scale1);
    }
    __syncthreads();
    TA_PHASE(0, 1);
    const bool any_rows = p_a <= p_b;
    const int sr_lo = any_rows ? ty1[p_a].i0 : 0;
    const int sh = any_rows ? ty1[p_b].i1 - sr_lo + 1 : 0;             // <= ROWS

    // -- H1: T[r][c] = fma(lx0, x[r][i0], lx1 * x[r][i1]); all loads of the lane first, 32-bit element offsets
    {
        float a[RPW], b[RPW];
        // 32-bit BYTE offsets from the wave-uniform plane pointer (4 * size * size < 2^32, host-checked): the loads
        // take the scalar-base + 32-bit-offset form, no 64-bit address arithmetic per access
        const char* base = reinterpret_cast<const char*>(xp);
        const int w0 = min(wave, max(sh - 1, 0));                      // this wave's first row, inside the rectangle
        const unsigned row0 = static_cast<unsigned>((sr_lo + w0) * size), bstep = 16u * static_cast<unsigned>(size);
        const unsigned b0 = (row0 + static_cast<unsigned>(tx1.i0)) * 4u, b1 = (row0 + static_cast<unsigned>(tx1.i1)) * 4u;
        // rows past the rectangle are clamped to its last row (scalar min) and lanes outside the image read column 0:
        // every load is in bounds and unpredicated; what they produce is never read by V1
        const int last = max(sh - 1 - w0, 0);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const unsigned off = static_cast<unsigned>(min(4 * i, last & ~3)) * (bstep / 4u);
            a[i] = *reinterpret_cast<const float*>(base + (b0 + off));
            b[i] = *reinterpret_cast<const float*>(base + (b1 + off));
        }
        float* out = T + wave * 64 + lane;
#pragma unroll
        for (int i = 0; i < RPW; ++i) out[i * 256] = fmaf(tx1.l0, a[i], tx1.l1 * b[i]);
    }
    __syncthreads();
    TA_PHASE(0, 2);
    // -- V1: mid[p][c] = fma(ly0, T[i0][c], ly1 * T[i1][c]) inside the rescaled image, 0 in the padding (dim.py:65)
    {
        const float* Tc = T + lane - sr_lo * 64;
        float* out = mid + wave * 64 + lane;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int p = wave + 4 * i;
            float val = 0.0f;
            if (col_ok && p >= p_a && p <= p_b) {
                const Tap ty = ty1[p];
                val = fmaf(ty.l0, Tc[ty.i0 * 64], ty.l1 * Tc[ty.i1 * 64]);
            }
            out[i * 256] = val;
        }
    }
    __syncthreads();
    TA_PHASE(0, 3);
    // -- H2: u[p][ox] = fma(lx0, mid[p][i0], lx1 * mid[p][i1])      (u overwrites T: every lane is past V1)
    float* u = T;
    if (lane < twc) {
        const float* m0 = mid + wave * 64 + (tx2.i0 - px_lo);
        const float* m1 = mid + wave * 64 + (tx2.i1 - px_lo);
        float* out = u + wave * 64 + lane;
#pragma unroll
        for (int i = 0; i < RPW; ++i)
            if (wave + 4 * i < mh) out[i * 256] = fmaf(tx2.l0, m0[i * 256], tx2.l1 * m1[i * 256]);
    }
    __syncthreads();
    TA_PHASE(0, 4);
    // -- V2: y[oy][ox] = fma(ly0, u[i0][ox], ly1 * u[i1][ox])
    if (lane < twc) {
        const float* uc = u + lane - py_lo * 64;
        char* base = reinterpret_cast<char*>(yp);
        const unsigned first = static_cast<unsigned>((oy0 + wave) * size + ox0 + lane) * 4u, bstep = 16u * static_cast<unsigned>(size);
#pragma unroll
        for (int i = 0; i < kDimLaneRows / 4; ++i) {
            const int r = wave + 4 * i;
            if (r < th) {
                const Tap ty = ty2[r];
                *reinterpret_cast<float*>(base + (first + i * bstep)) = fmaf(ty.l0, uc[ty.i0 * 64], ty.l1 * uc[ty.i1 * 64]);
            }
        }
    }
    TA_PHASE(0, 5);
}

// --------------------------------------------------------------------------------------- backward
constexpr int kDimBwdTile = 32;         // 32 x 32 pixels of gx per workgroup
constexpr int kDimBwdMaxMid = 104;      // side of the LDS-resident window of d(rescaled); rate <= ~2.9 (OPS; 62.5 KB of LDS at 224 -> 649)

struct Range {
    int lo, hi;
};

__global__ __launch_bounds__(kBlock) void dim_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                         float* __restrict__ ws, int size, int resize, int rnd, int top,
                                                         int left, int tiles_per_side) {
    __shared__ float red[kBlock / kWave];
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Tap* t2 = reinterpret_cast<Tap*>(smem_raw);                  // [size]    out pixel  -> padded index
    Tap* t1 = t2 + size;                                         // [rnd]     rescaled   -> x index
    Range* inv2 = reinterpret_cast<Range*>(t1 + rnd);            // [resize]  padded idx -> out pixels touching it
    Range* inv1 = inv2 + resize;                                 // [size]    x index    -> rescaled pixels touching it
    float* mid = reinterpret_cast<float*>(inv1 + size);          // [mh][mw] d(rescaled) window (host-sized)

    for (int i = threadIdx.x; i < resize; i += kBlock) inv2[i] = Range{INT_MAX, -1};
    for (int i = threadIdx.x; i < size; i += kBlock) inv1[i] = Range{INT_MAX, -1};
    __syncthreads();
    for (int o = threadIdx.x; o < size; o += kBlock) {
        const Tap tp = make_tap(o, resize, size);
        t2[o] = tp;
        atomicMin(&inv2[tp.i0].lo, o);
        atomicMax(&inv2[tp.i1].hi, o);        // i1 >= i0 and taps are monotone: [lo, hi] covers both taps
        atomicMin(&inv2[tp.i1].lo, o);
        atomicMax(&inv2[tp.i0].hi, o);
    }
    for (int o = threadIdx.x; o < rnd; o += kBlock) {
        const Tap tp = make_tap(o, size, rnd);
        t1[o] = tp;
        atomicMin(&inv1[tp.i0].lo, o);
        atomicMax(&inv1[tp.i1].hi, o);
        atomicMin(&inv1[tp.i1].lo, o);
        atomicMax(&inv1[tp.i0].hi, o);
    }
    __syncthreads();

    const int tiles = tiles_per_side * tiles_per_side;
    const unsigned tid = blockIdx.x;
    const int64_t plane = tid / tiles;
    const int t = tid % tiles;
    const int iy0 = (t / tiles_per_side) * kDimBwdTile, ix0 = (t % tiles_per_side) * kDimBwdTile;
    const int iy1 = min(iy0 + kDimBwdTile, size) - 1, ix1 = min(ix0 + kDimBwdTile, size) - 1;
    const float* gyp = gy + plane * static_cast<int64_t>(size) * size;
    float* gxp = gx + plane * static_cast<int64_t>(size) * size;

    // rescaled-pixel window [ry_lo, ry_hi] x [rx_lo, rx_hi] that feeds this tile of x (ranges are monotone)
    int ry_lo = INT_MAX, ry_hi = -1, rx_lo = INT_MAX, rx_hi = -1;
    for (int i = iy0; i <= iy1; ++i) { ry_lo = min(ry_lo, inv1[i].lo); ry_hi = max(ry_hi, inv1[i].hi); }
    for (int i = ix0; i <= ix1; ++i) { rx_lo = min(rx_lo, inv1[i].lo); rx_hi = max(rx_hi, inv1[i].hi); }
    const int mh = ry_hi - ry_lo + 1, mw = rx_hi - rx_lo + 1;     // <= kDimBwdMaxMid (checked on the host)

    // stage A: d(rescaled)[ry][rx] = d(padded)[ry+top][rx+left] = gather over the outputs touching it
    for (int idx = threadIdx.x; idx < mh * mw; idx += kBlock) {
        const int ry = ry_lo + idx / mw, rx = rx_lo + idx % mw;
        const int py = ry + top, px = rx + left;
        const Range oy_r = inv2[py], ox_r = inv2[px];
        float acc = 0.0f;
        for (int oy = oy_r.lo; oy <= oy_r.hi; ++oy) {
            const Tap ty = t2[oy];
            const int ys[2] = {ty.i0, ty.i1};
            const float ly[2] = {ty.l0, ty.l1};
            for (int ox = ox_r.lo; ox <= ox_r.hi; ++ox) {
                const Tap tx = t2[ox];
                const int xs[2] = {tx.i0, tx.i1};
                const float lx[2] = {tx.l0, tx.l1};
                const float g = gyp[static_cast<int64_t>(oy) * size + ox];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        if (ys[a] == py && xs[b] == px) acc = fmaf(ly[a] * lx[b], g, acc);
            }
        }
        mid[idx] = acc;
    }
    __syncthreads();

    // stage B: gx[iy][ix] = gather over the rescaled pixels touching it
    float asum = 0.0f;
#pragma unroll
    for (int u = 0; u < kDimBwdTile * kDimBwdTile / kBlock; ++u) {
        const int local = u * kBlock + threadIdx.x;
        const int iy = iy0 + local / kDimBwdTile, ix = ix0 + local % kDimBwdTile;
        if (iy >= size || ix >= size) continue;
        const Range ry_r = inv1[iy], rx_r = inv1[ix];
        float acc = 0.0f;
        for (int ry = ry_r.lo; ry <= ry_r.hi; ++ry) {
            const Tap ty = t1[ry];
            const int ys[2] = {ty.i0, ty.i1};
            const float ly[2] = {ty.l0, ty.l1};
            for (int rx = rx_r.lo; rx <= rx_r.hi; ++rx) {
                const Tap tx = t1[rx];
                const int xs[2] = {tx.i0, tx.i1};
                const float lx[2] = {tx.l0, tx.l1};
                const float g = mid[(ry - ry_lo) * mw + (rx - rx_lo)];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        if (ys[a] == iy && xs[b] == ix) acc = fmaf(ly[a] * lx[b], g, acc);
            }
        }
        gxp[static_cast<int64_t>(iy) * size + ix] = acc;
        asum += fabsf(acc);
    }
    const float total = block_sum(asum, red);
    if (ws != nullptr && threadIdx.x == 0) ws[tid] = total;
}


// Lane-per-column form of the backward, the counterpart of dim_fwd_lanes_kernel.  The adjoint
// cannot be made separable without changing the rounding (ATen accumulates fma(ly*lx, g, acc) over the output pixels in
// row-major order), so both stages stay 2-D gathers in that order -- but the bookkeeping changes:
//   * the outputs that touch one source index form a contiguous run o = first .. first+n-1 (taps are monotone), n <= 3
//     when downsampling (scale >= 1) and <= 4 when upsampling by at most 1.5: a `Hit` holds first, n and the weight each
//     of them contributes (the tap that equals the index; both taps only at the clamped border);
//   * a lane owns one COLUMN of the stage's target (its Hit lives in registers), waves stride over rows whose Hit is
//     wave-uniform (scalar loop bounds), so the inner loops are n_y x K predicated fma's with no searching, no div/mod;
//   * tile = 32 rows x tw <= 64 columns of gx, tw chosen on the host so the window of d(rescaled) is <= 64 columns.
constexpr int kHitSlots = 4;
struct Hit {
    int first, n;
    unsigned both;                 // bit k: output first+k hits the index with BOTH taps (clamped border): w then w2
    float w[kHitSlots], w2[kHitSlots];
    int pad;
};

// outputs o of a 1-D resample (in_size -> out_size, scale = in/out) whose taps touch source index t
__device__ __forceinline__ Hit find_hits(int t, int in_size, int out_size, float scale) {
    Hit h;
    h.first = 0; h.n = 0; h.both = 0u; h.pad = 0;
#pragma unroll
    for (int k = 0; k < kHitSlots; ++k) { h.w[k] = 0.0f; h.w2[k] = 0.0f; }
    // src(o) >= t-1  <=>  o >= (t-0.5)/scale - 0.5 ; start two below the estimate, the taps themselves decide
    const float est = (static_cast<float>(t) - 0.5f) / scale - 0.5f;
    int c = static_cast<int>(floorf(est)) - 1;
    c = c < 0 ? 0 : c;
    int first = -1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int o = c + j;
        if (o < out_size && first < 0) {
            const Tap tp = make_tap_scaled(o, in_size, scale);
            if (tp.i0 == t || tp.i1 == t) first = o;
        }
    }
    if (first < 0) return h;
    h.first = first;
    bool open = true;
#pragma unroll
    for (int k = 0; k < kHitSlots; ++k) {
        const int o = first + k;
        if (open && o < out_size) {
            const Tap tp = make_tap_scaled(o, in_size, scale);
            const bool h0 = tp.i0 == t, h1 = tp.i1 == t;
            if (h0 || h1) {
                h.w[k] = h0 ? tp.l0 : tp.l1;
                h.w2[k] = tp.l1;
                if (h0 && h1) h.both |= 1u << k;
                h.n = k + 1;
            } else {
                open = false;
            }
        } else {
            open = false;
        }
    }
    return h;
}

// One source value g reached through a row slot (weight wy; wy2 if the row hits with both taps) and column slot k of
// hx, accumulated in ATen's order (a = 0, 1 outer; b = 0, 1 inner).  Branch-free: a slot that does not apply gets
// weight 0 and g = 0, and fma(0, 0, acc) == acc bit for bit (acc starts at +0 and a sum never produces -0; the masked
// g keeps a non-finite neighbour out).  FAST = no lane of the wave and not this row has a double hit (everywhere
// except the clamped last row / column): one multiply and one fma per slot.
template <bool FAST>
__device__ __forceinline__ float hit_accumulate(float acc, float g, float wy, float wy2, bool both_y, const Hit& hx, int k) {
    const bool on = k < hx.n;
    const float gm = on ? g : 0.0f;
    const float wx = hx.w[k];                                  // 0 beyond n (find_hits)
    acc = fmaf(wy * wx, gm, acc);
    if (!FAST) {
        const bool bx = on && ((hx.both >> k) & 1u);
        const float g2 = bx ? g : 0.0f;
        const float wx2 = bx ? hx.w2[k] : 0.0f;
        acc = fmaf(wy * wx2, g2, acc);
        if (both_y) {
            acc = fmaf(wy2 * wx, gm, acc);
            acc = fmaf(wy2 * wx2, g2, acc);
        }
    }
    return acc;
}

template <int RPW, int SB, int PP, int SA>      // SB: hit slots of stage B (3 when no index of x is touched by 4 rescaled pixels);
                                        // SA: hit slots of stage A (2 when no padded index is touched by 3 outputs: always so when the
                                        // second resample shrinks, resize > size -- 4 instead of 9 gathers per window pixel);
                                        // PP: planes per workgroup (they share the tile's hit tables)
__global__ __launch_bounds__(kBlock) void dim_bwd_lanes_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                               float* __restrict__ ws, int size, int resize, int rnd,
                                                               int top, int left,
                                                               float scale1, float scale2, int tw, int tiles_x,
                                                               int tiles_y) {
    constexpr int ROWS = 4 * RPW;
    TA_PHASE_BEGIN();
    __shared__ __attribute__((aligned(16))) Hit colB[64];               // tile column ix   -> rescaled columns
    __shared__ __attribute__((aligned(16))) Hit rowB[kDimLaneRows];     // tile row iy      -> rescaled rows
    __shared__ __attribute__((aligned(16))) Hit colA[64];               // window column px -> output columns
    __shared__ __attribute__((aligned(16))) Hit rowA[ROWS];             // window row py    -> output rows
    __shared__ __attribute__((aligned(16))) float mid[ROWS * 64];       // d(rescaled) window
    __shared__ float red[kBlock / kWave];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles = tiles_x * tiles_y;
    const int tid = static_cast<int>(blockIdx.x);
    const int group = tid / tiles;                      // PP consecutive planes share this tile's tables (the geometry
    const int t = tid - group * tiles;                  // is the same for every plane: built once, used PP times)
    const int tyi = t / tiles_x;
    const int iy0 = tyi * kDimLaneRows, ix0 = (t - tyi * tiles_x) * tw;
    const int th = min(kDimLaneRows, size - iy0), twc = min(tw, size - ix0);

    // -- stage-B tables: which rescaled pixels feed this tile of x
    if (wave == 0 && lane < twc) colB[lane] = find_hits(ix0 + lane, size, rnd, scale1);
    if (wave == 1 && lane < th) rowB[lane] = find_hits(iy0 + lane, size, rnd, scale1);
    __syncthreads();
    TA_PHASE(1, 0);
    const int rx_lo = colB[0].first, rx_hi = colB[twc - 1].first + colB[twc - 1].n - 1;
    const int ry_lo = rowB[0].first, ry_hi = rowB[th - 1].first + rowB[th - 1].n - 1;
    const int mw = rx_hi - rx_lo + 1, mh = ry_hi - ry_lo + 1;           // <= 64, <= ROWS (host-checked)
    // -- stage-A tables: which output pixels feed that window (through the zero padding: px = rx + left, py = ry + top)
    if (wave == 0 && lane < mw) colA[lane] = find_hits(rx_lo + lane + left, resize, size, scale2);
    {
        const int p = static_cast<int>(threadIdx.x) - 64;
        if (p >= 0 && p < mh) rowA[p] = find_hits(ry_lo + p + top, resize, size, scale2);
    }
    __syncthreads();
    TA_PHASE(1, 1);

#pragma unroll 1
    for (int q = 0; q < PP; ++q) {
    const int plane = group * PP + q;
    const char* gyp = reinterpret_cast<const char*>(gy + static_cast<int64_t>(plane) * size * size);
    char* gxp = reinterpret_cast<char*>(gx + static_cast<int64_t>(plane) * size * size);
    // -- stage A: mid[p][c] = d(rescaled)[ry_lo + p][rx_lo + c]
    {
        Hit hx = colA[lane < mw ? lane : 0];
        if (lane >= mw) hx.n = 0;
        const bool any_both_x = __builtin_amdgcn_readfirstlane(__any(hx.both != 0u)) != 0;
        unsigned col[SA];                                    // byte offsets of the lane's output columns
#pragma unroll
        for (int k = 0; k < SA; ++k) col[k] = static_cast<unsigned>(min(hx.first + k, size - 1)) * 4u;
        const unsigned row_bytes = 4u * static_cast<unsigned>(size);
        // software pipeline over the wave's rows: the 9 loads of row p + 4 are in flight while row p is accumulated
        auto fetch = [&](int p, float (&g)[SA][SA]) {
            const int first_y = __builtin_amdgcn_readfirstlane(rowA[p].first);
#pragma unroll
            for (int ky = 0; ky < SA; ++ky) {
                const unsigned row = static_cast<unsigned>(min(first_y + ky, size - 1)) * row_bytes;
#pragma unroll
                for (int kx = 0; kx < SA; ++kx) g[ky][kx] = *reinterpret_cast<const float*>(gyp + (row + col[kx]));
            }
        };
        auto reduce = [&](int p, const float (&g)[SA][SA]) {
            const Hit* hy = &rowA[p];
            const int n_y = __builtin_amdgcn_readfirstlane(hy->n);                    // <= 3: scale2 >= 1
            const unsigned both_y = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(hy->both)));
            float acc = 0.0f;
            if (!any_both_x && both_y == 0u) {
#pragma unroll
                for (int ky = 0; ky < SA; ++ky)
                    if (ky < n_y)
#pragma unroll
                        for (int kx = 0; kx < SA; ++kx)
                            acc = hit_accumulate<true>(acc, g[ky][kx], hy->w[ky], 0.0f, false, hx, kx);
            } else {
#pragma unroll
                for (int ky = 0; ky < SA; ++ky)
                    if (ky < n_y)
#pragma unroll
                        for (int kx = 0; kx < SA; ++kx)
                            acc = hit_accumulate<false>(acc, g[ky][kx], hy->w[ky], hy->w2[ky], (both_y >> ky) & 1u, hx, kx);
            }
            mid[p * 64 + lane] = acc;
        };
        float ga[SA][SA], gb[SA][SA];
        // the prefetch is unconditional (row index clamped to the window) so that the compiler's wait counters know,
        // on every path, that the nine newest loads are not the ones being consumed
        int p = wave;
        if (p < mh) fetch(p, ga);
#pragma unroll 1
        while (p < mh) {
            fetch(min(p + 4, mh - 1), gb);
            reduce(p, ga);
            p += 4;
            if (p >= mh) break;
            fetch(min(p + 4, mh - 1), ga);
            reduce(p, gb);
            p += 4;
        }
    }
    __syncthreads();
    TA_PHASE(1, 2);
    // -- stage B: gx[iy][ix]
    float asum = 0.0f;
    Hit hx = colB[lane < twc ? lane : 0];
    if (lane >= twc) hx.both = 0u;
    const bool any_both_x = __builtin_amdgcn_readfirstlane(__any(hx.both != 0u)) != 0;     // all lanes of the wave vote
    if (lane < twc) {
        int col[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) col[k] = min(hx.first - rx_lo + k, 63);
        unsigned out = static_cast<unsigned>((iy0 + wave) * size + ix0 + lane) * 4u;
        const unsigned bstep = 16u * static_cast<unsigned>(size);
#pragma unroll 1
        for (int r = wave; r < th; r += 4, out += bstep) {
            const Hit* hy = &rowB[r];
            const int first_y = __builtin_amdgcn_readfirstlane(hy->first) - ry_lo;
            const int n_y = __builtin_amdgcn_readfirstlane(hy->n);
            const unsigned both_y = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(hy->both)));
            float acc = 0.0f;
            if (!any_both_x && both_y == 0u) {
#pragma unroll
                for (int ky = 0; ky < SB; ++ky)
                    if (ky < n_y) {
                        const float* mrow = mid + (first_y + ky) * 64;
#pragma unroll
                        for (int kx = 0; kx < SB; ++kx)
                            acc = hit_accumulate<true>(acc, mrow[col[kx]], hy->w[ky], 0.0f, false, hx, kx);
                    }
            } else {
#pragma unroll
                for (int ky = 0; ky < SB; ++ky)
                    if (ky < n_y) {
                        const float* mrow = mid + (first_y + ky) * 64;
#pragma unroll
                        for (int kx = 0; kx < SB; ++kx)
                            acc = hit_accumulate<false>(acc, mrow[col[kx]], hy->w[ky], hy->w2[ky], (both_y >> ky) & 1u, hx, kx);
                    }
            }
            *reinterpret_cast<float*>(gxp + out) = acc;
            asum += fabsf(acc);
        }
    }
    TA_PHASE(1, 3);
    const float total = block_sum(asum, red);             // (its barriers also fence `mid` for the next plane)
    if (ws != nullptr && threadIdx.x == 0) ws[plane * tiles + t] = total;
    TA_PHASE(1, 4);
    }   // planes of the group
}


// ------------------------------------------------------------------------------------------------
// PreprocessingModel with a Resize (reference: transferattack/utils.py:50-53, 72-79 -- Inception-v3: 224 -> 299, mean = std =
// 0.5):  y = (bilinear_{in->out}(x) - mean[c]) / std[c]  as ONE kernel each way, with the DIM kernels' taps and rounding
// order (ATen's upsample_bilinear2d: width first, then height; backward fma(ly*lx, g, acc) over the outputs in row-major
// order).  The backward is the LAST kernel of that member's input gradient, so it also emits the per-tile sums of |gx|.
// A tile is 32 rows x 64 columns, one lane per column, waves stride over rows.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void resize_norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 const float* __restrict__ mean, const float* __restrict__ stdv,
                                                                 int channels, int in_size, int out_size, float scale,
                                                                 int tiles_x, int tiles_y) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles = tiles_x * tiles_y;
    const int tid = static_cast<int>(blockIdx.x);
    const int plane = tid / tiles;
    const int t = tid - plane * tiles;
    const int tyi = t / tiles_x;
    const int oy0 = tyi * kDimLaneRows, ox = (t - tyi * tiles_x) * 64 + lane;
    const int ch = plane % channels;
    const float mu = mean[ch], sd = stdv[ch];
    if (ox >= out_size) return;
    const Tap tx = make_tap_scaled(ox, in_size, scale);
    const float* xp = x + static_cast<int64_t>(plane) * in_size * in_size;
    float* yp = y + static_cast<int64_t>(plane) * out_size * out_size;
    const int rows = min(kDimLaneRows, out_size - oy0);
    for (int r = wave; r < rows; r += 4) {
        const Tap ty = make_tap_scaled(oy0 + r, in_size, scale);
        const float* r0 = xp + static_cast<int64_t>(ty.i0) * in_size;
        const float* r1 = xp + static_cast<int64_t>(ty.i1) * in_size;
        const float a = fmaf(tx.l0, r0[tx.i0], tx.l1 * r0[tx.i1]);
        const float b = fmaf(tx.l0, r1[tx.i0], tx.l1 * r1[tx.i1]);
        const float v = fmaf(ty.l0, a, ty.l1 * b);
        yp[static_cast<int64_t>(oy0 + r) * out_size + ox] = (v - mu) / sd;
    }
}

__global__ __launch_bounds__(kBlock) void resize_norm_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                 const float* __restrict__ stdv, float* __restrict__ ws,
                                                                 int channels, int in_size, int out_size, float scale,
                                                                 int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) Hit rowH[kDimLaneRows];
    __shared__ float red[kBlock / kWave];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles = tiles_x * tiles_y;
    const int tid = static_cast<int>(blockIdx.x);
    const int plane = tid / tiles;
    const int t = tid - plane * tiles;
    const int tyi = t / tiles_x;
    const int iy0 = tyi * kDimLaneRows, ix = (t - tyi * tiles_x) * 64 + lane;
    const int rows = min(kDimLaneRows, in_size - iy0);
    const float sd = stdv[plane % channels];
    if (static_cast<int>(threadIdx.x) < rows) rowH[threadIdx.x] = find_hits(iy0 + static_cast<int>(threadIdx.x), in_size, out_size, scale);
    __syncthreads();
    const float* gyp = gy + static_cast<int64_t>(plane) * out_size * out_size;
    float* gxp = gx + static_cast<int64_t>(plane) * in_size * in_size;
    float asum = 0.0f;
    if (ix < in_size) {
        const Hit hx = find_hits(ix, in_size, out_size, scale);
        int col[kHitSlots];
#pragma unroll
        for (int k = 0; k < kHitSlots; ++k) col[k] = min(hx.first + k, out_size - 1);
        for (int r = wave; r < rows; r += 4) {
            const Hit* hy = &rowH[r];
            float acc = 0.0f;
#pragma unroll
            for (int ky = 0; ky < kHitSlots; ++ky)
                if (ky < hy->n) {
                    const float* grow = gyp + static_cast<int64_t>(hy->first + ky) * out_size;
#pragma unroll
                    for (int kx = 0; kx < kHitSlots; ++kx)          // Normalize's backward first: gy / std, then the adjoint
                        acc = hit_accumulate<false>(acc, grow[col[kx]] / sd, hy->w[ky], hy->w2[ky], (hy->both >> ky) & 1u, hx, kx);
                }
            gxp[static_cast<int64_t>(iy0 + r) * in_size + ix] = acc;
            asum += fabsf(acc);
        }
    }
    const float total = block_sum(asum, red);
    if (ws != nullptr && threadIdx.x == 0) ws[tid] = total;
}

}  // namespace ta

using namespace ta;

static int check_geom(int64_t planes, int size, int resize, int rnd, int top, int left) {
    TA_REQUIRE(planes > 0 && size > 0 && size <= kDimMaxSide && resize <= kDimMaxSide, "bad shape");
    TA_REQUIRE(rnd > 0 && rnd <= resize && top >= 0 && left >= 0 && top + rnd <= resize && left + rnd <= resize,
               "geometry (rnd=%d, top=%d, left=%d) does not fit resize=%d", rnd, top, left, resize);
    return 0;
}

// largest number of outputs of a 1-D resample (in_size -> out_size) that touch one source index; same fp32 arithmetic as
// the kernels' make_tap (fmaf is the exact fused operation on the host as well)
static int max_hits(int in_size, int out_size) {
    const float scale = static_cast<float>(in_size) / static_cast<float>(out_size);
    int count[kDimMaxSide] = {0};                       // in_size <= kDimMaxSide (check_geom)
    for (int o = 0; o < out_size; ++o) {
        float src = fmaf(scale, static_cast<float>(o) + 0.5f, -0.5f);
        src = src < 0.0f ? 0.0f : src;
        int i0 = static_cast<int>(src);
        i0 = i0 > in_size - 1 ? in_size - 1 : i0;
        const int i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
        ++count[i0];
        if (i1 != i0) ++count[i1];
    }
    int best = 0;
    for (int i = 0; i < in_size; ++i) best = count[i] > best ? count[i] : best;
    return best;
}

static int64_t ta_dim_bwd_tiles_impl(int size, int resize) {
    if (size <= 0 || resize <= 0) return 0;
    if (resize > size && 2 * resize <= 3 * size && static_cast<int64_t>(size) * size < (1ll << 28)) {
        const double up = static_cast<double>(resize) / size;
        const int tw = static_cast<int>(fmin(64.0, floor(63.0 / up) - 1.0));
        const int rows = static_cast<int>(ceil((kDimLaneRows + 1) * up)) + 1;
        if (tw >= 8 && rows <= 68) return ceil_div(size, tw) * ceil_div(size, kDimLaneRows);
    }
    const int64_t tps = ceil_div(size, kDimBwdTile);
    return tps * tps;
}

extern "C" int64_t ta_dim_bwd_tiles(int size, int resize) { return ta_dim_bwd_tiles_impl(size, resize); }

extern "C" int ta_dim_fwd(const float* x, float* y, int64_t planes, int size, int resize, int rnd, int top, int left,
                          void* stream) {
    TA_REQUIRE(x && y && x != y, "null or aliased pointers");
    if (int rc = check_geom(planes, size, resize, rnd, top, left)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // lane-per-column kernel: every geometry with resize <= ~2 * size
    if (static_cast<int64_t>(size) * size < (1ll << 30)) {
        const float scale1 = static_cast<float>(size) / static_cast<float>(rnd);        // make_tap's divisions, once
        const float scale2 = static_cast<float>(resize) / static_cast<float>(size);
        const double ratio = static_cast<double>(resize) / size;
        const int tw = static_cast<int>(fmin(64.0, floor(61.0 / ratio) + 1.0));        // window <= (tw-1)*ratio + 3 <= 64
        const int rows = static_cast<int>(ceil((kDimLaneRows - 1) * ratio)) + 4;       // window rows + 1 row of x
        if (tw >= 8 && rows <= 68) {
            const int tiles_x = static_cast<int>(ceil_div(size, tw)), tiles_y = static_cast<int>(ceil_div(size, kDimLaneRows));
            // (one plane per workgroup: sharing the forward's taps over the three planes of an image was measured slower at
            // every size, r2e / r2f: 13.8 -> 16.0 us at 96 planes, 54 -> 65 us at 480)
            const int64_t lane_blocks = planes * tiles_x * tiles_y;
            TA_REQUIRE(lane_blocks < (1ll << 31) - 8, "too many tiles");
            const dim3 grid(static_cast<unsigned>(lane_blocks));
            if (rows <= 40)
                hipLaunchKernelGGL(dim_fwd_lanes_kernel<10>, grid, dim3(kBlock), 0, st, x, y, size, resize, rnd, top, left,
                                   scale1, scale2, tw, tiles_x, tiles_y);
            else
                hipLaunchKernelGGL(dim_fwd_lanes_kernel<17>, grid, dim3(kBlock), 0, st, x, y, size, resize, rnd, top, left,
                                   scale1, scale2, tw, tiles_x, tiles_y);
            return check_launch("dim_fwd_lanes");
        }
    }
    // table-driven gather: resize ratios up to ~2.9
    const int tps = static_cast<int>(ceil_div(size, kDimFwdTile));
    const int64_t blocks = planes * tps * tps;
    TA_REQUIRE(blocks < (1ll << 31), "too many tiles");
    // a 32-pixel output tile reads at most this many padded pixels per axis
    const int mid_side = static_cast<int>(ceil_div(static_cast<int64_t>(kDimFwdTile) * resize, size)) + 3;
    TA_REQUIRE(mid_side <= kDimFwdMaxMid, "resize ratio %d/%d too large for the fused forward", resize, size);
    const size_t smem = sizeof(Tap) * (static_cast<size_t>(size) + rnd) + sizeof(float) * mid_side * mid_side;
    hipLaunchKernelGGL(dim_fwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), smem, st, x, y, size, resize, rnd,
                       top, left, tps);
    return check_launch("dim_fwd");
}

extern "C" int ta_dim_bwd(const float* gy, float* gx, float* ws, int64_t planes, int size, int resize, int rnd, int top,
                          int left, void* stream) {
    TA_REQUIRE(gy && gx && gy != gx, "null or aliased pointers");
    if (int rc = check_geom(planes, size, resize, rnd, top, left)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // lane-per-column gather: resize > size, resize <= 1.5 * size and < 2^28 elements per plane (the choice depends on
    // (size, resize) only, so ta_dim_bwd_tiles tells the caller how many |gx| sums per plane `ws` receives)
    if (resize > size && 2 * resize <= 3 * size && static_cast<int64_t>(size) * size < (1ll << 28)) {
        const float scale1 = static_cast<float>(size) / static_cast<float>(rnd);
        const float scale2 = static_cast<float>(resize) / static_cast<float>(size);
        const double up = static_cast<double>(resize) / size;                          // bound for rnd / size
        const int tw = static_cast<int>(fmin(64.0, floor(63.0 / up) - 1.0));           // window <= (tw + 1) * up + 1 <= 64
        const int rows = static_cast<int>(ceil((kDimLaneRows + 1) * up)) + 1;
        if (tw >= 8 && rows <= 68) {
            const int tiles_x = static_cast<int>(ceil_div(size, tw)), tiles_y = static_cast<int>(ceil_div(size, kDimLaneRows));
            // the three planes of an RGB image share one workgroup's hit tables (a quarter of the backward's instructions)
            // when that still leaves >= 10 workgroups per CU: 116 -> 109 us at 480 planes, but 27 -> 32 us at 96 (r2e)
            const int pp = (planes % 3 == 0 && planes / 3 * tiles_x * tiles_y >= 2560) ? 3 : 1;
            const int64_t lane_blocks = planes / pp * tiles_x * tiles_y;
            TA_REQUIRE(planes * tiles_x * tiles_y < (1ll << 31) - 8, "too many tiles");
            const dim3 grid(static_cast<unsigned>(lane_blocks));
            const bool three = max_hits(size, rnd) <= 3;       // true for every rnd < 1.5 * size away from degenerate sizes
            const bool two_a = max_hits(resize, size) <= 2;    // stage A: outputs per padded index (<= 2 whenever resize > size)
#define TA_DIM_BWD_PP(RPW, SB, SA)                                                                                       \
    do {                                                                                                                 \
        if (pp == 3)                                                                                                     \
            hipLaunchKernelGGL((dim_bwd_lanes_kernel<RPW, SB, 3, SA>), grid, dim3(kBlock), 0, st, gy, gx, ws, size,      \
                               resize, rnd, top, left, scale1, scale2, tw, tiles_x, tiles_y);                            \
        else                                                                                                             \
            hipLaunchKernelGGL((dim_bwd_lanes_kernel<RPW, SB, 1, SA>), grid, dim3(kBlock), 0, st, gy, gx, ws, size,      \
                               resize, rnd, top, left, scale1, scale2, tw, tiles_x, tiles_y);                            \
    } while (0)
#define TA_DIM_BWD(RPW, SB) do { if (two_a) TA_DIM_BWD_PP(RPW, SB, 2); else TA_DIM_BWD_PP(RPW, SB, 3); } while (0)
            if (rows <= 40) { if (three) TA_DIM_BWD(10, 3); else TA_DIM_BWD(10, 4); }
            else { if (three) TA_DIM_BWD(17, 3); else TA_DIM_BWD(17, 4); }
#undef TA_DIM_BWD
#undef TA_DIM_BWD_PP
            return check_launch("dim_bwd_lanes");
        }
    }
    // a 32-pixel tile of x (plus one neighbour each side) is fed by at most this many rescaled pixels per axis
    const int mid_side = static_cast<int>(ceil_div(static_cast<int64_t>(kDimBwdTile + 2) * rnd, size)) + 3;
    TA_REQUIRE(mid_side <= kDimBwdMaxMid, "resize ratio %d/%d too large for the fused backward", rnd, size);
    const int tps = static_cast<int>(ceil_div(size, kDimBwdTile));
    const int64_t blocks = planes * tps * tps;
    TA_REQUIRE(blocks < (1ll << 31), "too many tiles");
    // LDS is sized for the window this geometry needs (not the 80 x 80 worst case): ~18 KB at 224/246 -> 8 workgroups/CU
    const size_t smem = sizeof(Tap) * (static_cast<size_t>(size) + rnd) + sizeof(Range) * (static_cast<size_t>(resize) + size) +
                        sizeof(float) * mid_side * mid_side;
    hipLaunchKernelGGL(dim_bwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), smem, st, gy, gx, ws, size, resize,
                       rnd, top, left, tps);
    return check_launch("dim_bwd");
}

extern "C" int64_t ta_resize_tiles(int side) { return side <= 0 ? 0 : ceil_div(side, 64) * ceil_div(side, kDimLaneRows); }

static int check_resize(int64_t n, int c, int in_size, int out_size) {
    TA_REQUIRE(n > 0 && c > 0 && in_size > 0 && out_size > 0 && in_size <= kDimMaxSide && out_size <= kDimMaxSide, "bad shape");
    TA_REQUIRE(n * c * ta_resize_tiles(in_size > out_size ? in_size : out_size) < (1ll << 31), "too many tiles");
    return 0;
}

extern "C" int ta_resize_normalize_fwd(const float* x, float* y, const float* mean, const float* stdv, int64_t n, int c,
                                       int in_size, int out_size, void* stream) {
    TA_REQUIRE(x && y && mean && stdv && x != y, "null or aliased pointers");
    if (int rc = check_resize(n, c, in_size, out_size)) return rc;
    const float scale = static_cast<float>(in_size) / static_cast<float>(out_size);      // ATen: area_pixel_compute_scale
    const int tiles_x = static_cast<int>(ceil_div(out_size, 64)), tiles_y = static_cast<int>(ceil_div(out_size, kDimLaneRows));
    hipLaunchKernelGGL(resize_norm_fwd_kernel, dim3(static_cast<unsigned>(n * c * tiles_x * tiles_y)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), x, y, mean, stdv, c, in_size, out_size, scale, tiles_x, tiles_y);
    return check_launch("resize_normalize_fwd");
}

// ws (nullable): n * c * ta_resize_tiles(in_size) sums of |gx|, c * tiles consecutive per image
extern "C" int ta_resize_normalize_bwd(const float* gy, float* gx, const float* stdv, float* ws, int64_t n, int c, int in_size,
                                       int out_size, void* stream) {
    TA_REQUIRE(gy && gx && stdv && gy != gx, "null or aliased pointers");
    if (int rc = check_resize(n, c, in_size, out_size)) return rc;
    TA_REQUIRE(max_hits(in_size, out_size) <= kHitSlots, "resize %d -> %d: more than %d outputs touch one input index", in_size,
               out_size, kHitSlots);
    const float scale = static_cast<float>(in_size) / static_cast<float>(out_size);
    const int tiles_x = static_cast<int>(ceil_div(in_size, 64)), tiles_y = static_cast<int>(ceil_div(in_size, kDimLaneRows));
    hipLaunchKernelGGL(resize_norm_bwd_kernel, dim3(static_cast<unsigned>(n * c * tiles_x * tiles_y)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), gy, gx, stdv, ws, c, in_size, out_size, scale, tiles_x, tiles_y);
    return check_launch("resize_normalize_bwd");
}

#ifdef TA_DIM_PHASE_CLOCK
// cycles[kernel][phase] (kernel 0 = dim_fwd_lanes, 1 = dim_bwd_lanes) summed over the workgroups since the last call, and the
// number of workgroups; resets the table.  tools/dim_phase_clock.py only.
extern "C" int ta_dim_phase_clock_read(unsigned long long* cycles16, unsigned long long* groups2) {
    if (hipError_t err = hipDeviceSynchronize()) return static_cast<int>(err);
    if (hipError_t err = hipMemcpyFromSymbol(cycles16, HIP_SYMBOL(ta::ta_dim_phase_cycles), sizeof(unsigned long long) * 16)) return static_cast<int>(err);
    if (hipError_t err = hipMemcpyFromSymbol(groups2, HIP_SYMBOL(ta::ta_dim_phase_groups), sizeof(unsigned long long) * 2)) return static_cast<int>(err);
    unsigned long long zeros[16] = {0};
    if (hipError_t err = hipMemcpyToSymbol(HIP_SYMBOL(ta::ta_dim_phase_cycles), zeros, sizeof(unsigned long long) * 16)) return static_cast<int>(err);
    return static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(ta::ta_dim_phase_groups), zeros, sizeof(unsigned long long) * 2));
}
#endif
