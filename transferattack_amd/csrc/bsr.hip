// BSR -- block shuffle and rotation -- for gfx950 (reference: BSR.shuffle / BSR.transform,
// input_transformation/bsr.py:41-67): every copy of the batch is cut into `nb` strips along one axis, the strips are
// shuffled, each strip is rotated about its own centre by a random angle (bilinear, zero fill), cut into `nb` blocks
// along the other axis and shuffled again.  The reference does this with ~12 ATen launches per strip (split, affine grid,
// bmm, grid_sample, split, cat ...), 700+ per iteration at 20 copies; here the whole stack is ONE gather kernel, and its
// backward one gather kernel (the exact adjoint, no float atomics: every source pixel collects the <= 9 rotated pixels
// whose bilinear footprint contains it, copies added in autograd's descending order).
//
// Arithmetic follows torchvision's tensor rotate + ATen's grid_sample (restated in oracle/fgsm_oracle.py):
//   base grid   xb = px - w/2 + 0.5, yb = py - h/2 + 0.5                      (exact in fp32)
//   grid        gx = xb*rt00 + yb*rt10,  gy = xb*rt01 + yb*rt11               rt = theta^T / (w/2, h/2), fp32, from the host
//   unnormalise ix = fma(gx + 1, w/2, -0.5)                                   (align_corners = False)
//   bilinear    out = fma(v_se, se, fma(v_sw, sw, fma(v_ne, ne, v_nw * nw))), taps outside the strip read 0
// with the fused operations placed where the reference's CPU build (BLAS sgemm for the grid, ATen's vectorised
// grid_sampler compiled with contraction) places them [probe, torch 2.10 CPU: these expressions reproduce
// F.grid_sample's forward bit for bit] -> the forward equals the reference's.  The backward adds, per source pixel, the
// products weight * gy of the rotated pixels that reach it: in raster order by default; in ATen's order when
// TA_ATEN_SUM_LANES = 8 | 16 is set (its vectorised backward walks the flattened strip in chunks of `lanes` pixels and,
// inside a chunk, corner by corner -- nw, ne, sw, se -- then lane by lane), which makes the backward the reference's
// bit for bit as well.  Forward and backward use the SAME device function for the sampling point.
//
// plan (device int32, per copy, stride 1 + 7*nb + 3*nb*nb), strips and blocks in OUTPUT order:
//   [0] first axis (0 = rows, 1 = columns)
//   per strip i:  src_start, length, out_start, rt00, rt10, rt01, rt11 (float bits)
//   per (i, j):   src_start, length, out_start of block j of strip i along the second axis
#include <stdlib.h>
#include "ta_common.h"

namespace ta {

constexpr int kBsrMaxBlocks = 8;
constexpr int kBsrRows = 8;                    // rows of a plane per workgroup (2 per wave)
constexpr int kBsrMaxPlanInts = 8192;          // all copies' plans staged in LDS by the backward

__host__ __device__ inline int bsr_plan_stride(int nb) { return 1 + 7 * nb + 3 * nb * nb; }

struct BsrSample {
    int x0, y0;                // north-west tap (strip coordinates)
    float nw, ne, sw, se;      // bilinear weights
};

// sampling point of rotated-strip pixel (py, px) of a strip of h x w pixels
__device__ __forceinline__ BsrSample bsr_sample(const int* __restrict__ strip, int py, int px, int h, int w) {
    const float rt00 = __int_as_float(strip[3]), rt10 = __int_as_float(strip[4]);
    const float rt01 = __int_as_float(strip[5]), rt11 = __int_as_float(strip[6]);
    const float xb = static_cast<float>(px) - 0.5f * static_cast<float>(w) + 0.5f;
    const float yb = static_cast<float>(py) - 0.5f * static_cast<float>(h) + 0.5f;
    const float gx = fmaf(yb, rt10, xb * rt00), gy = fmaf(yb, rt11, xb * rt01);
    const float ix = fmaf(gx + 1.0f, 0.5f * static_cast<float>(w), -0.5f);
    const float iy = fmaf(gy + 1.0f, 0.5f * static_cast<float>(h), -0.5f);
    const float fx = floorf(ix), fy = floorf(iy);
    const float we = ix - fx, ww = 1.0f - we, ws = iy - fy, wn = 1.0f - ws;     // east / west / south / north shares
    BsrSample s;
    s.x0 = static_cast<int>(fx);
    s.y0 = static_cast<int>(fy);
    s.nw = wn * ww; s.ne = wn * we; s.sw = ws * ww; s.se = ws * we;
    return s;
}

// ---------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(kBlock) void bsr_fwd_kernel(const float* __restrict__ x, const int* __restrict__ plan,
                                                         float* __restrict__ y, int planes, int H, int W, int nb,
                                                         int row_tiles) {
    __shared__ int cp[1 + 7 * kBsrMaxBlocks + 3 * kBsrMaxBlocks * kBsrMaxBlocks];
    const int stride = bsr_plan_stride(nb);
    const int tile = blockIdx.x % row_tiles;
    const int plane = (blockIdx.x / row_tiles) % planes;
    const int copy = blockIdx.x / (row_tiles * planes);
    for (int i = threadIdx.x; i < stride; i += kBlock) cp[i] = plan[copy * stride + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d0 = cp[0];
    const int64_t plane_elems = static_cast<int64_t>(H) * W;
    const float* xp = x + static_cast<int64_t>(plane) * plane_elems;
    float* yp = y + (static_cast<int64_t>(copy) * planes + plane) * plane_elems;
    for (int Y = tile * kBsrRows + wave; Y < min((tile + 1) * kBsrRows, H); Y += kBlock / 64) {
        for (int X = lane; X < W; X += 64) {
            const int u = d0 == 0 ? Y : X, v = d0 == 0 ? X : Y;
            int i = 0;
            for (int k = 1; k < nb; ++k) i = u >= cp[1 + 7 * k + 2] ? k : i;
            const int* strip = cp + 1 + 7 * i;
            const int* blocks = cp + 1 + 7 * nb + 3 * nb * i;
            int j = 0;
            for (int k = 1; k < nb; ++k) j = v >= blocks[3 * k + 2] ? k : j;
            const int pu = u - strip[2], pv = blocks[3 * j] + (v - blocks[3 * j + 2]);
            const int s0 = strip[0], len = strip[1];
            const int h = d0 == 0 ? len : H, w = d0 == 0 ? W : len;
            const int py = d0 == 0 ? pu : pv, px = d0 == 0 ? pv : pu;
            const BsrSample s = bsr_sample(strip, py, px, h, w);
            const int oy = d0 == 0 ? s0 : 0, ox = d0 == 0 ? 0 : s0;         // strip origin inside the plane
            auto tap = [&](int cy, int cx) {
                return (cy >= 0 && cy < h && cx >= 0 && cx < w) ? xp[static_cast<int64_t>(oy + cy) * W + ox + cx] : 0.0f;
            };
            const float v_nw = tap(s.y0, s.x0), v_ne = tap(s.y0, s.x0 + 1), v_sw = tap(s.y0 + 1, s.x0), v_se = tap(s.y0 + 1, s.x0 + 1);
            yp[static_cast<int64_t>(Y) * W + X] = fmaf(v_se, s.se, fmaf(v_sw, s.sw, fmaf(v_ne, s.ne, v_nw * s.nw)));
        }
    }
}

// --------------------------------------------------------------------------------------------- backward
__global__ __launch_bounds__(kBlock) void bsr_bwd_kernel(const float* __restrict__ gy, const int* __restrict__ plan,
                                                         float* __restrict__ gx, float* __restrict__ ws, int planes, int H,
                                                         int W, int copies, int nb, int row_tiles, int lanes) {
    __shared__ int plans[kBsrMaxPlanInts];
    __shared__ float red[kBlock / kWave];
    const int stride = bsr_plan_stride(nb);
    for (int i = threadIdx.x; i < copies * stride; i += kBlock) plans[i] = plan[i];
    __syncthreads();
    const int tile = blockIdx.x % row_tiles;
    const int plane = blockIdx.x / row_tiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t plane_elems = static_cast<int64_t>(H) * W;
    float asum = 0.0f;
    for (int sy = tile * kBsrRows + wave; sy < min((tile + 1) * kBsrRows, H); sy += kBlock / 64) {
        for (int sx = lane; sx < W; sx += 64) {
            float acc = 0.0f;
            for (int copy = copies - 1; copy >= 0; --copy) {
                const int* cp = plans + copy * stride;
                const int d0 = cp[0];
                const int u = d0 == 0 ? sy : sx;
                int i = 0;
                for (int k = 1; k < nb; ++k) i = (u >= cp[1 + 7 * k] && u < cp[1 + 7 * k] + cp[1 + 7 * k + 1]) ? k : i;
                const int* strip = cp + 1 + 7 * i;
                const int* blocks = cp + 1 + 7 * nb + 3 * nb * i;
                const int s0 = strip[0], len = strip[1];
                const int h = d0 == 0 ? len : H, w = d0 == 0 ? W : len;
                const int cy = d0 == 0 ? sy - s0 : sy, cx = d0 == 0 ? sx : sx - s0;        // strip coordinates
                // centre of the rotated pixels that can reach (cy, cx): inverse rotation (the matrix is orthonormal)
                const float m0 = __int_as_float(strip[3]) * (0.5f * w), m1 = __int_as_float(strip[4]) * (0.5f * w);
                const float m3 = __int_as_float(strip[5]) * (0.5f * h), m4 = __int_as_float(strip[6]) * (0.5f * h);
                const float cxb = static_cast<float>(cx) - 0.5f * w + 0.5f, cyb = static_cast<float>(cy) - 0.5f * h + 0.5f;
                const int pxc = static_cast<int>(rintf(m0 * cxb + m3 * cyb + 0.5f * w - 0.5f));
                const int pyc = static_cast<int>(rintf(m1 * cxb + m4 * cyb + 0.5f * h - 0.5f));
                const float* gyp = gy + (static_cast<int64_t>(copy) * planes + plane) * plane_elems;
                float ck = 0.0f;
                float term[9];                                   // reference-order mode: the products, then sorted adds
                int key[9], hits = 0;
                for (int py = pyc - 1; py <= pyc + 1; ++py) {
                    for (int px = pxc - 1; px <= pxc + 1; ++px) {
                        if (py < 0 || py >= h || px < 0 || px >= w) continue;
                        const BsrSample s = bsr_sample(strip, py, px, h, w);
                        const int dy = cy - s.y0, dx = cx - s.x0;
                        if (dy < 0 || dy > 1 || dx < 0 || dx > 1) continue;
                        const float wgt = dy == 0 ? (dx == 0 ? s.nw : s.ne) : (dx == 0 ? s.sw : s.se);
                        // where rotated pixel (py, px) of this strip went in the output
                        const int pu = d0 == 0 ? py : px, pv = d0 == 0 ? px : py;
                        int j = 0;
                        for (int k = 1; k < nb; ++k) j = (pv >= blocks[3 * k] && pv < blocks[3 * k] + blocks[3 * k + 1]) ? k : j;
                        const int ou = strip[2] + pu, ov = blocks[3 * j + 2] + (pv - blocks[3 * j]);
                        const int Y = d0 == 0 ? ou : ov, X = d0 == 0 ? ov : ou;
                        const float prod = wgt * gyp[static_cast<int64_t>(Y) * W + X];
                        if (lanes == 0) {
                            ck += prod;                          // raster order of the rotated pixels
                        } else {
                            const int flat = py * w + px, corner = 2 * dy + dx;
                            term[hits] = prod;
                            key[hits++] = ((flat / lanes) * 4 + corner) * lanes + flat % lanes;
                        }
                    }
                }
                if (lanes != 0) {
                    for (int a = 1; a < hits; ++a) {             // insertion sort of <= 9 entries by ATen's visiting order
                        const int ka = key[a];
                        const float ta_ = term[a];
                        int b = a - 1;
                        while (b >= 0 && key[b] > ka) { key[b + 1] = key[b]; term[b + 1] = term[b]; --b; }
                        key[b + 1] = ka;
                        term[b + 1] = ta_;
                    }
                    for (int a = 0; a < hits; ++a) ck += term[a];
                }
                acc = copy == copies - 1 ? ck : acc + ck;
            }
            gx[static_cast<int64_t>(plane) * plane_elems + static_cast<int64_t>(sy) * W + sx] = acc;
            asum += fabsf(acc);
        }
    }
    const float total = block_sum(asum, red);
    if (ws != nullptr && threadIdx.x == 0) ws[blockIdx.x] = total;
}

}  // namespace ta

using namespace ta;

static int check_bsr(const void* a, const void* b, const void* plan, int64_t planes, int h, int w, int copies, int nb) {
    TA_REQUIRE(a && b && plan && a != b, "null or aliased pointers");
    TA_REQUIRE(planes > 0 && h > 0 && w > 0 && copies > 0, "bad shape");
    TA_REQUIRE(nb >= 1 && nb <= kBsrMaxBlocks, "num_block %d outside 1..%d", nb, kBsrMaxBlocks);
    TA_REQUIRE(planes * ceil_div(h, kBsrRows) * copies < (1ll << 31), "too many tiles");
    return 0;
}

extern "C" int64_t ta_bsr_tiles(int h) { return h > 0 ? ceil_div(h, kBsrRows) : 0; }

extern "C" int ta_bsr_fwd(const float* x, const int32_t* plan, float* y, int64_t planes, int h, int w, int copies, int nb,
                          void* stream) {
    if (int rc = check_bsr(x, y, plan, planes, h, w, copies, nb)) return rc;
    const int row_tiles = static_cast<int>(ceil_div(h, kBsrRows));
    hipLaunchKernelGGL(bsr_fwd_kernel, dim3(static_cast<unsigned>(copies * planes * row_tiles)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), x, plan, y, static_cast<int>(planes), h, w, nb, row_tiles);
    return check_launch("bsr_fwd");
}

extern "C" int ta_bsr_bwd(const float* gy, const int32_t* plan, float* gx, float* ws, int64_t planes, int h, int w, int copies,
                          int nb, void* stream) {
    if (int rc = check_bsr(gy, gx, plan, planes, h, w, copies, nb)) return rc;
    TA_REQUIRE(copies * bsr_plan_stride(nb) <= kBsrMaxPlanInts, "copies * plan stride exceeds the %d ints staged in LDS",
               kBsrMaxPlanInts);
    const int row_tiles = static_cast<int>(ceil_div(h, kBsrRows));
    const char* env = getenv("TA_ATEN_SUM_LANES");               // verification mode, read at every call like update.hip's
    const int value = env == nullptr ? 0 : atoi(env);
    const int lanes = (value == 8 || value == 16) ? value : 0;
    hipLaunchKernelGGL(bsr_bwd_kernel, dim3(static_cast<unsigned>(planes * row_tiles)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), gy, plan, gx, ws, static_cast<int>(planes), h, w, copies, nb,
                       row_tiles, lanes);
    return check_launch("bsr_bwd");
}
