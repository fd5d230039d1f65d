// SIA block transform for gfx950 (reference: SIA.blocktransform / SIA.transform, input_transformation/sia.py:41-100):
// every copy of the batch is cut into num_block x num_block rectangles at random positions and each rectangle gets one
// of seven operations -- roll along rows / columns, flip rows / columns, rotate by 180 degrees, multiply by a random
// scalar, add uniform noise and clip to [0, 1].  The reference runs ~10 small ATen kernels per rectangle (180+ launches
// per iteration for 20 copies); here the whole stack is ONE gather kernel forward (reads x once through the cache,
// writes 4 * copies B/element) and ONE gather kernel backward.
//
// The host draws the cuts / operations / steps / scale factors in the reference's order from the reference's
// generators and hands them over as a small int32 table (`plan`); the noise is either given (tests: the reference's
// CPU draws) or generated in the kernel from a Philox stream keyed by the element's position in the output stack.
//
// Backward: the five geometric operations are permutations inside their rectangle (the gradient is fetched from where
// the element went), scaling multiplies by the factor, noise + clip passes the gradient where 0 <= x + noise <= 1
// (clamp's backward, bounds included).  The copies' contributions are added in DESCENDING copy order -- the order
// autograd accumulates them into x.grad (pinned by tests/golden/sia.npz).
#include "philox.h"

namespace ta {

constexpr int kSiaMaxBlocks = 8;                     // num_block per axis
constexpr int kSiaRows = 8;                          // rows of one plane per workgroup (2 per wave)

__host__ __device__ constexpr int sia_plan_stride(int nb) { return 2 * (nb + 1) + 3 * nb * nb; }

enum SiaOp { kRollRows = 0, kRollCols = 1, kFlipRows = 2, kFlipCols = 3, kRotate180 = 4, kScale = 5, kNoise = 6 };

constexpr int kSiaChunks = 5;                        // 64-lane column chunks a workgroup covers: 320 columns per column tile

// Everything about a rectangle is wave-uniform: the row a wave works on lies in ONE row band of a copy, and a 64-column
// chunk meets one or two column segments of that band.  `plan` is read through a uniform global pointer (scalar loads), so
// band / segment / operation live in SGPRs, the operation is a uniform branch, and the lanes only do the index map of
// their own column under the segment's exec mask: ~20 VALU per element instead of a per-lane table search.
struct SiaBand {
    int bi, r_lo, bh;
};
__device__ __forceinline__ SiaBand sia_band(const int* __restrict__ plan, int nb, int r) {
    SiaBand band;
    band.bi = 0;
    for (int m = 1; m < nb; ++m) band.bi += r >= plan[m] ? 1 : 0;
    band.r_lo = plan[band.bi];
    band.bh = plan[band.bi + 1] - band.r_lo;
    return band;
}

// Where an output element of a copy comes from (band, segment, operation, source offset) is the same for every plane
// (n, c) of the batch: a lane computes it ONCE and applies it to P planes -- the index map was ~20 VALU instructions per
// element and plane, the kernels were bound by issuing them (0.8 / 0.6 TB/s in round 2).  Per plane what is left is one load
// at a plane-strided address, the operation's arithmetic and the store; P loads are in flight per lane.
template <int P>
__global__ __launch_bounds__(kBlock) void sia_fwd_kernel(const float* __restrict__ x, const int* __restrict__ plan,
                                                         const float* __restrict__ noise, float* __restrict__ y,
                                                         int planes, int h, int w, int nb, int row_tiles, int col_tiles,
                                                         int plane_groups, float noise_radius, uint64_t seed, uint64_t offset) {
    const int stride = sia_plan_stride(nb);
    const int ctile = blockIdx.x % col_tiles;
    const int tile = (blockIdx.x / col_tiles) % row_tiles;
    const int pg = (blockIdx.x / (col_tiles * row_tiles)) % plane_groups;
    const int copy = blockIdx.x / (col_tiles * row_tiles * plane_groups);
    const int plane0 = pg * P, np = min(P, planes - plane0);
    const int* cp = plan + copy * stride;
    const int* cols = cp + nb + 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t hw = static_cast<int64_t>(h) * w;
    const float* xp = x + static_cast<int64_t>(plane0) * hw;
    const int64_t out0 = (static_cast<int64_t>(copy) * planes + plane0) * hw;
    for (int r = tile * kSiaRows + wave; r < min((tile + 1) * kSiaRows, h); r += kBlock / 64) {
        const SiaBand band = sia_band(cp, nb, r);
        const int lr0 = r - band.r_lo;
        for (int bj = 0; bj < nb; ++bj) {                              // segment parameters: once per (row, segment)
            const int c_lo = cols[bj], c_hi = cols[bj + 1];
            const int* blk = cp + 2 * (nb + 1) + 3 * (band.bi * nb + bj);
            const int op = blk[0], step = blk[1], bw = c_hi - c_lo;
            const float scale = __int_as_float(blk[2]);
#pragma unroll
            for (int ch = 0; ch < kSiaChunks; ++ch) {
                const int c0 = (ctile * kSiaChunks + ch) * 64;          // first column of this chunk (uniform)
                if (c0 >= w || c_hi <= c0 || c_lo >= c0 + 64) continue; // the segment does not meet this chunk
                const int c = c0 + lane;
                if (c >= c_lo && c < c_hi) {
                    int lr = lr0, lc = c - c_lo;
                    if (op == kRollRows) lr = lr - step + (lr < step ? band.bh : 0);          // out[r] = in[(r - step) mod bh]
                    if (op == kRollCols) lc = lc - step + (lc < step ? bw : 0);
                    if (op == kFlipRows || op == kRotate180) lr = band.bh - 1 - lr;
                    if (op == kFlipCols || op == kRotate180) lc = bw - 1 - lc;
                    const int src = (band.r_lo + lr) * w + c_lo + lc, dst = r * w + c;
                    float v[P];
#pragma unroll
                    for (int q = 0; q < P; ++q) v[q] = xp[static_cast<int64_t>(q < np ? q : 0) * hw + src];
                    if (op == kScale) {
#pragma unroll
                        for (int q = 0; q < P; ++q) v[q] = scale * v[q];
                    }
                    if (op == kNoise) {                                 // one rectangle in seven: a rolled loop keeps the Philox
#pragma unroll 1                                                        // state out of the common path's register budget
                        for (int q = 0; q < np; ++q) {
                            const int64_t o = out0 + static_cast<int64_t>(q) * hw + dst;
                            const float nz = noise ? noise[o] : uniform1(static_cast<uint64_t>(o), seed, offset, noise_radius);
                            y[o] = fminf(fmaxf(xp[static_cast<int64_t>(q) * hw + src] + nz, 0.0f), 1.0f);
                        }
                        continue;
                    }
#pragma unroll
                    for (int q = 0; q < P; ++q)
                        if (q < np) y[out0 + static_cast<int64_t>(q) * hw + dst] = v[q];
                }
            }
        }
    }
}

// backward: gx[plane][r][c] = sum over the copies, last copy first, of what came back for the element's image
template <int P>
__global__ __launch_bounds__(kBlock) void sia_bwd_kernel(const float* __restrict__ gy, const int* __restrict__ plan,
                                                         const float* __restrict__ x, const float* __restrict__ noise,
                                                         float* __restrict__ gx, int planes, int h, int w, int copies,
                                                         int nb, int row_tiles, int col_tiles, float noise_radius,
                                                         uint64_t seed, uint64_t offset) {
    const int stride = sia_plan_stride(nb);
    const int ctile = blockIdx.x % col_tiles;
    const int tile = (blockIdx.x / col_tiles) % row_tiles;
    const int pg = blockIdx.x / (col_tiles * row_tiles);
    const int plane0 = pg * P, np = min(P, planes - plane0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t hw = static_cast<int64_t>(h) * w;
    for (int r = tile * kSiaRows + wave; r < min((tile + 1) * kSiaRows, h); r += kBlock / 64) {
        const int64_t here = static_cast<int64_t>(plane0) * hw + static_cast<int64_t>(r) * w;
        float acc[kSiaChunks][P];
#pragma unroll
        for (int ch = 0; ch < kSiaChunks; ++ch)
#pragma unroll
            for (int q = 0; q < P; ++q) acc[ch][q] = 0.0f;
        for (int copy = copies - 1; copy >= 0; --copy) {                // copies outer: one band lookup per (copy, row)
            const int* cp = plan + copy * stride;
            const int* cols = cp + nb + 1;
            const SiaBand band = sia_band(cp, nb, r);
            const int64_t copy0 = (static_cast<int64_t>(copy) * planes + plane0) * hw;
            const int lr0 = r - band.r_lo;
            for (int bj = 0; bj < nb; ++bj) {
                const int c_lo = cols[bj], c_hi = cols[bj + 1];
                const int* blk = cp + 2 * (nb + 1) + 3 * (band.bi * nb + bj);
                const int op = blk[0], step = blk[1], bw = c_hi - c_lo;
                const float scale = __int_as_float(blk[2]);
#pragma unroll
                for (int ch = 0; ch < kSiaChunks; ++ch) {
                    const int c0 = (ctile * kSiaChunks + ch) * 64;
                    if (c0 >= w || c_hi <= c0 || c_lo >= c0 + 64) continue;
                    const int c = c0 + lane;
                    if (c >= c_lo && c < c_hi) {
                        int lr = lr0, lc = c - c_lo;
                        if (op == kRollRows) { lr += step; lr -= lr >= band.bh ? band.bh : 0; }   // where in[r] went
                        if (op == kRollCols) { lc += step; lc -= lc >= bw ? bw : 0; }
                        if (op == kFlipRows || op == kRotate180) lr = band.bh - 1 - lr;
                        if (op == kFlipCols || op == kRotate180) lc = bw - 1 - lc;
                        const int src = (band.r_lo + lr) * w + c_lo + lc;
                        float g[P];
#pragma unroll
                        for (int q = 0; q < P; ++q) g[q] = gy[copy0 + static_cast<int64_t>(q < np ? q : 0) * hw + src];
                        if (op == kScale) {
#pragma unroll
                            for (int q = 0; q < P; ++q) g[q] = g[q] * scale;
                        }
                        if (op == kNoise) {
                            unsigned pass = 0u;                           // bit q: plane q's sample stayed inside [0, 1]
#pragma unroll 1
                            for (int q = 0; q < np; ++q) {
                                const int64_t o = copy0 + static_cast<int64_t>(q) * hw + static_cast<int64_t>(r) * w + c;
                                const float nz = noise ? noise[o] : uniform1(static_cast<uint64_t>(o), seed, offset, noise_radius);
                                const float v = x[here + static_cast<int64_t>(q) * hw + c] + nz;
                                pass |= (v >= 0.0f && v <= 1.0f) ? (1u << q) : 0u;
                            }
#pragma unroll
                            for (int q = 0; q < P; ++q) g[q] = ((pass >> q) & 1u) ? g[q] : 0.0f;
                        }
#pragma unroll
                        for (int q = 0; q < P; ++q) acc[ch][q] = copy == copies - 1 ? g[q] : acc[ch][q] + g[q];
                    }
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < kSiaChunks; ++ch) {
            const int c = (ctile * kSiaChunks + ch) * 64 + lane;
            if (c < w) {
#pragma unroll
                for (int q = 0; q < P; ++q)
                    if (q < np) gx[here + static_cast<int64_t>(q) * hw + c] = acc[ch][q];
            }
        }
    }
}

}  // namespace ta

using namespace ta;

static int check_sia(const void* a, const void* b, const void* plan, int64_t planes, int h, int w, int copies, int nb) {
    TA_REQUIRE(a && b && plan && a != b, "null or aliased pointers");
    TA_REQUIRE(planes > 0 && h > 0 && w > 0 && copies > 0, "bad shape");
    TA_REQUIRE(nb >= 1 && nb <= kSiaMaxBlocks, "num_block %d outside 1..%d", nb, kSiaMaxBlocks);
    TA_REQUIRE(planes * ceil_div(h, kSiaRows) * ceil_div(w, 64 * kSiaChunks) * copies < (1ll << 31), "too many tiles");
    return 0;
}

extern "C" int ta_sia_fwd(const float* x, const int32_t* plan, const float* noise, float* y, int64_t planes, int h, int w,
                          int copies, int nb, float noise_radius, uint64_t seed, uint64_t offset, void* stream) {
    if (int rc = check_sia(x, y, plan, planes, h, w, copies, nb)) return rc;
    const int row_tiles = static_cast<int>(ceil_div(h, kSiaRows)), col_tiles = static_cast<int>(ceil_div(w, 64 * kSiaChunks));
    // planes per lane: as many as still leave >= 8 workgroups per CU (the geometry is shared, the loads are not)
    const int64_t tiles = static_cast<int64_t>(copies) * row_tiles * col_tiles;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define TA_SIA_FWD(PP)                                                                                                        \
    hipLaunchKernelGGL((sia_fwd_kernel<PP>), dim3(static_cast<unsigned>(tiles * ceil_div(planes, PP))), dim3(kBlock), 0, st, x, \
                       plan, noise, y, static_cast<int>(planes), h, w, nb, row_tiles, col_tiles,                             \
                       static_cast<int>(ceil_div(planes, PP)), noise_radius, seed, offset)
    if (tiles * ceil_div(planes, 6) >= 2048) TA_SIA_FWD(6);
    else if (tiles * ceil_div(planes, 3) >= 2048) TA_SIA_FWD(3);
    else TA_SIA_FWD(1);
#undef TA_SIA_FWD
    return check_launch("sia_fwd");
}

extern "C" int ta_sia_bwd(const float* gy, const int32_t* plan, const float* x, const float* noise, float* gx,
                          int64_t planes, int h, int w, int copies, int nb, float noise_radius, uint64_t seed,
                          uint64_t offset, void* stream) {
    if (int rc = check_sia(gy, gx, plan, planes, h, w, copies, nb)) return rc;
    TA_REQUIRE(x != nullptr, "x is needed for the clip mask");
    const int row_tiles = static_cast<int>(ceil_div(h, kSiaRows)), col_tiles = static_cast<int>(ceil_div(w, 64 * kSiaChunks));
    const int64_t tiles = static_cast<int64_t>(row_tiles) * col_tiles;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define TA_SIA_BWD(PP)                                                                                                        \
    hipLaunchKernelGGL((sia_bwd_kernel<PP>), dim3(static_cast<unsigned>(tiles * ceil_div(planes, PP))), dim3(kBlock), 0, st, gy, \
                       plan, x, noise, gx, static_cast<int>(planes), h, w, copies, nb, row_tiles, col_tiles, noise_radius,   \
                       seed, offset)
    if (tiles * ceil_div(planes, 3) >= 1024) TA_SIA_BWD(3);
    else TA_SIA_BWD(1);
#undef TA_SIA_BWD
    return check_launch("sia_bwd");
}
