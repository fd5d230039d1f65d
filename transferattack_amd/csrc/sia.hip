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
constexpr int kSiaMaxPlanInts = 4096;                // LDS copy of the plan in the backward (16 KB)

__host__ __device__ constexpr int sia_plan_stride(int nb) { return 2 * (nb + 1) + 3 * nb * nb; }

enum SiaOp { kRollRows = 0, kRollCols = 1, kFlipRows = 2, kFlipCols = 3, kRotate180 = 4, kScale = 5, kNoise = 6 };

struct SiaCell {                                     // where one element sits in its copy's partition
    int r_lo, bh, c_lo, bw, op, step;
    float scale;
};

// plan layout per copy: rows[nb + 1], cols[nb + 1], then (op, step, scale bits) per block, rows outer
__device__ __forceinline__ SiaCell sia_locate(const int* plan, int nb, int bi, int c) {
    const int* cols = plan + nb + 1;
    int bj = 0;
    for (int m = 1; m < nb; ++m) bj += c >= cols[m] ? 1 : 0;
    const int* blk = plan + 2 * (nb + 1) + 3 * (bi * nb + bj);
    SiaCell cell;
    cell.r_lo = plan[bi];
    cell.bh = plan[bi + 1] - cell.r_lo;
    cell.c_lo = cols[bj];
    cell.bw = cols[bj + 1] - cell.c_lo;
    cell.op = blk[0];
    cell.step = blk[1];
    cell.scale = __int_as_float(blk[2]);
    return cell;
}

__device__ __forceinline__ int sia_row_block(const int* plan, int nb, int r) {
    int bi = 0;
    for (int m = 1; m < nb; ++m) bi += r >= plan[m] ? 1 : 0;
    return bi;
}

// forward: y[copy][plane][r][c] = op(x[plane][source of (r, c)])
__global__ __launch_bounds__(kBlock) void sia_fwd_kernel(const float* __restrict__ x, const int* __restrict__ plan,
                                                         const float* __restrict__ noise, float* __restrict__ y,
                                                         int planes, int h, int w, int nb, int row_tiles,
                                                         float noise_radius, uint64_t seed, uint64_t offset) {
    __shared__ int plan_s[2 * (kSiaMaxBlocks + 1) + 3 * kSiaMaxBlocks * kSiaMaxBlocks];
    const int stride = sia_plan_stride(nb);
    const int tile = blockIdx.x % row_tiles;
    const int plane = (blockIdx.x / row_tiles) % planes;
    const int copy = blockIdx.x / (row_tiles * planes);
    for (int i = threadIdx.x; i < stride; i += kBlock) plan_s[i] = plan[copy * stride + i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* xp = x + static_cast<int64_t>(plane) * h * w;
    const int64_t out_plane = (static_cast<int64_t>(copy) * planes + plane) * h * w;
    for (int r = tile * kSiaRows + wave; r < min((tile + 1) * kSiaRows, h); r += kBlock / 64) {
        const int bi = sia_row_block(plan_s, nb, r);
        for (int c = lane; c < w; c += 64) {
            const SiaCell cell = sia_locate(plan_s, nb, bi, c);
            int lr = r - cell.r_lo, lc = c - cell.c_lo;
            if (cell.op == kRollRows) lr = lr - cell.step + (lr < cell.step ? cell.bh : 0);      // out[r] = in[(r - step) mod bh]
            if (cell.op == kRollCols) lc = lc - cell.step + (lc < cell.step ? cell.bw : 0);
            if (cell.op == kFlipRows || cell.op == kRotate180) lr = cell.bh - 1 - lr;
            if (cell.op == kFlipCols || cell.op == kRotate180) lc = cell.bw - 1 - lc;
            float v = xp[(cell.r_lo + lr) * w + cell.c_lo + lc];
            const int64_t o = out_plane + static_cast<int64_t>(r) * w + c;
            if (cell.op == kScale) v = cell.scale * v;
            if (cell.op == kNoise) {
                const float nz = noise ? noise[o] : uniform1(static_cast<uint64_t>(o), seed, offset, noise_radius);
                v = fminf(fmaxf(v + nz, 0.0f), 1.0f);
            }
            y[o] = v;
        }
    }
}

// backward: gx[plane][r][c] = sum over the copies, last copy first, of what came back for the element's image
__global__ __launch_bounds__(kBlock) void sia_bwd_kernel(const float* __restrict__ gy, const int* __restrict__ plan,
                                                         const float* __restrict__ x, const float* __restrict__ noise,
                                                         float* __restrict__ gx, int planes, int h, int w, int copies,
                                                         int nb, int row_tiles, float noise_radius, uint64_t seed,
                                                         uint64_t offset) {
    __shared__ int plan_s[kSiaMaxPlanInts];
    const int stride = sia_plan_stride(nb);
    for (int i = threadIdx.x; i < copies * stride; i += kBlock) plan_s[i] = plan[i];
    __syncthreads();
    const int tile = blockIdx.x % row_tiles;
    const int plane = blockIdx.x / row_tiles;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t plane_elems = static_cast<int64_t>(h) * w;
    for (int r = tile * kSiaRows + wave; r < min((tile + 1) * kSiaRows, h); r += kBlock / 64)
        for (int c = lane; c < w; c += 64) {
            const int64_t here = static_cast<int64_t>(plane) * plane_elems + static_cast<int64_t>(r) * w + c;
            float acc = 0.0f;
            for (int copy = copies - 1; copy >= 0; --copy) {
                const int* cp = plan_s + copy * stride;
                const SiaCell cell = sia_locate(cp, nb, sia_row_block(cp, nb, r), c);
                int lr = r - cell.r_lo, lc = c - cell.c_lo;
                if (cell.op == kRollRows) { lr += cell.step; lr -= lr >= cell.bh ? cell.bh : 0; }   // where in[r] went
                if (cell.op == kRollCols) { lc += cell.step; lc -= lc >= cell.bw ? cell.bw : 0; }
                if (cell.op == kFlipRows || cell.op == kRotate180) lr = cell.bh - 1 - lr;
                if (cell.op == kFlipCols || cell.op == kRotate180) lc = cell.bw - 1 - lc;
                const int64_t copy_plane = (static_cast<int64_t>(copy) * planes + plane) * plane_elems;
                float g = gy[copy_plane + static_cast<int64_t>(cell.r_lo + lr) * w + cell.c_lo + lc];
                if (cell.op == kScale) g = g * cell.scale;
                if (cell.op == kNoise) {
                    const int64_t o = copy_plane + static_cast<int64_t>(r) * w + c;
                    const float nz = noise ? noise[o] : uniform1(static_cast<uint64_t>(o), seed, offset, noise_radius);
                    const float v = x[here] + nz;
                    g = (v >= 0.0f && v <= 1.0f) ? g : 0.0f;
                }
                acc = copy == copies - 1 ? g : acc + g;
            }
            gx[here] = acc;
        }
}

}  // namespace ta

using namespace ta;

static int check_sia(const void* a, const void* b, const void* plan, int64_t planes, int h, int w, int copies, int nb) {
    TA_REQUIRE(a && b && plan && a != b, "null or aliased pointers");
    TA_REQUIRE(planes > 0 && h > 0 && w > 0 && copies > 0, "bad shape");
    TA_REQUIRE(nb >= 1 && nb <= kSiaMaxBlocks, "num_block %d outside 1..%d", nb, kSiaMaxBlocks);
    TA_REQUIRE(planes * ceil_div(h, kSiaRows) * copies < (1ll << 31), "too many tiles");
    return 0;
}

extern "C" int ta_sia_fwd(const float* x, const int32_t* plan, const float* noise, float* y, int64_t planes, int h, int w,
                          int copies, int nb, float noise_radius, uint64_t seed, uint64_t offset, void* stream) {
    if (int rc = check_sia(x, y, plan, planes, h, w, copies, nb)) return rc;
    const int row_tiles = static_cast<int>(ceil_div(h, kSiaRows));
    hipLaunchKernelGGL(sia_fwd_kernel, dim3(static_cast<unsigned>(copies * planes * row_tiles)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), x, plan, noise, y, static_cast<int>(planes), h, w, nb, row_tiles,
                       noise_radius, seed, offset);
    return check_launch("sia_fwd");
}

extern "C" int ta_sia_bwd(const float* gy, const int32_t* plan, const float* x, const float* noise, float* gx,
                          int64_t planes, int h, int w, int copies, int nb, float noise_radius, uint64_t seed,
                          uint64_t offset, void* stream) {
    if (int rc = check_sia(gy, gx, plan, planes, h, w, copies, nb)) return rc;
    TA_REQUIRE(x != nullptr, "x is needed for the clip mask");
    TA_REQUIRE(copies * sia_plan_stride(nb) <= kSiaMaxPlanInts, "plan of %d copies x %d blocks does not fit the kernel's table",
               copies, nb);
    const int row_tiles = static_cast<int>(ceil_div(h, kSiaRows));
    hipLaunchKernelGGL(sia_bwd_kernel, dim3(static_cast<unsigned>(planes * row_tiles)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), gy, plan, x, noise, gx, static_cast<int>(planes), h, w, copies, nb,
                       row_tiles, noise_radius, seed, offset);
    return check_launch("sia_bwd");
}
