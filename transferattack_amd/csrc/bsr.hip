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
// ta_set_sum_order(8 | 16) is in force (its vectorised backward walks the flattened strip in chunks of `lanes` pixels and,
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
constexpr int kBsrRows = 8;                    // rows of a plane per forward workgroup (2 per wave)
constexpr int kBsrBwdRows = 2;                 // rows of a plane per backward workgroup (= rows per |gx| tile sum): the
                                               // backward is long per pixel, small tiles keep >= 3 workgroups per CU
                                               // at the planes-per-thread counts that pay (measured: r2k)
constexpr int kBsrMaxPlanInts = 8192;          // all copies' plans staged in LDS by the backward (32 KB at most)

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
// Where an output pixel of a copy samples from (strip, block, rotated sampling point, weights) is the same for every
// image and channel: a thread owns one output pixel of P planes, finds the four taps once and reads / blends / writes
// the P planes through them.
template <int P>
__global__ __launch_bounds__(kBlock) void bsr_fwd_kernel(const float* __restrict__ x, const int* __restrict__ plan,
                                                         float* __restrict__ y, int planes, int H, int W, int nb,
                                                         int row_tiles) {
    __shared__ int cp[1 + 7 * kBsrMaxBlocks + 3 * kBsrMaxBlocks * kBsrMaxBlocks];
    const int stride = bsr_plan_stride(nb);
    const int groups = planes / P;
    const int tile = blockIdx.x % row_tiles;
    const int plane0 = ((blockIdx.x / row_tiles) % groups) * P;
    const int copy = blockIdx.x / (row_tiles * groups);
    for (int i = threadIdx.x; i < stride; i += kBlock) cp[i] = plan[copy * stride + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d0 = cp[0];
    const int64_t plane_elems = static_cast<int64_t>(H) * W;
    const float* xp = x + static_cast<int64_t>(plane0) * plane_elems;
    float* yp = y + (static_cast<int64_t>(copy) * planes + plane0) * plane_elems;
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
            // the four taps: inside the strip? byte offset in the plane (0 when outside: read, then replaced by 0)
            bool in[4];
            unsigned boff[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cy = s.y0 + (c >> 1), cx = s.x0 + (c & 1);
                in[c] = cy >= 0 && cy < h && cx >= 0 && cx < w;
                boff[c] = in[c] ? static_cast<unsigned>((oy + cy) * W + ox + cx) * 4u : 0u;
            }
            const unsigned out = static_cast<unsigned>(Y * W + X);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const char* base = reinterpret_cast<const char*>(xp + p * plane_elems);
                float t[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float val = *reinterpret_cast<const float*>(base + boff[c]);
                    t[c] = in[c] ? val : 0.0f;
                }
                yp[p * plane_elems + out] = fmaf(t[3], s.se, fmaf(t[2], s.sw, fmaf(t[1], s.ne, t[0] * s.nw)));
            }
        }
    }
}

// --------------------------------------------------------------------------------------------- backward
// The geometry of a copy -- which rotated pixels reach a source pixel, with which weight, and where they went in the
// output -- is the same for every image and channel.  A thread owns one source pixel of P planes: per copy it finds the
// (at most 9) contributions ONCE and then runs the P planes over them (one load + one product + one add each), so the
// search, which is all the arithmetic of this kernel, is paid once per P planes.  Per plane the products are added in
// the same order as before (raster order of the rotated pixels / ATen's visiting order, copies descending).
struct BsrHit {
    int off;          // Y * W + X of the rotated pixel in its output plane
    int key;          // ATen's visiting order (reference-order mode)
    float wgt;
    bool valid;
};

template <int P, bool ORDERED>
__global__ __launch_bounds__(kBlock) void bsr_bwd_kernel(const float* __restrict__ gy, const int* __restrict__ plan,
                                                         float* __restrict__ gx, float* __restrict__ ws, int planes, int H,
                                                         int W, int copies, int nb, int row_tiles, int lanes) {
    extern __shared__ __attribute__((aligned(16))) int plans[];      // copies * stride ints
    __shared__ float red[kBlock / kWave];
    const int stride = bsr_plan_stride(nb);
    for (int i = threadIdx.x; i < copies * stride; i += kBlock) plans[i] = plan[i];
    __syncthreads();
    const int tile = blockIdx.x % row_tiles;
    const int plane0 = (blockIdx.x / row_tiles) * P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t plane_elems = static_cast<int64_t>(H) * W;
    float asum[P];
#pragma unroll
    for (int p = 0; p < P; ++p) asum[p] = 0.0f;
    const int chunks_x = (W + 63) / 64;                  // the tile in 64-pixel row segments, dealt to the waves in turn
    for (int chunk = wave; chunk < kBsrBwdRows * chunks_x; chunk += kBlock / 64) {
        {
            const int sy = tile * kBsrBwdRows + chunk / chunks_x, sx = (chunk % chunks_x) * 64 + lane;
            if (sy >= H || sx >= W) continue;
            float acc[P];
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p] = 0.0f;
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
                BsrHit hit[9];
#pragma unroll
                for (int slot = 0; slot < 9; ++slot) {           // raster order of the candidates
                    const int py = pyc + slot / 3 - 1, px = pxc + slot % 3 - 1;
                    hit[slot].valid = false;
                    hit[slot].off = 0;
                    hit[slot].key = 0;
                    hit[slot].wgt = 0.0f;
                    if (py < 0 || py >= h || px < 0 || px >= w) continue;
                    const BsrSample smp = bsr_sample(strip, py, px, h, w);
                    const int dy = cy - smp.y0, dx = cx - smp.x0;
                    if (dy < 0 || dy > 1 || dx < 0 || dx > 1) continue;
                    // where rotated pixel (py, px) of this strip went in the output
                    const int pu = d0 == 0 ? py : px, pv = d0 == 0 ? px : py;
                    int j = 0;
                    for (int k = 1; k < nb; ++k) j = (pv >= blocks[3 * k] && pv < blocks[3 * k] + blocks[3 * k + 1]) ? k : j;
                    const int ou = strip[2] + pu, ov = blocks[3 * j + 2] + (pv - blocks[3 * j]);
                    const int flat = py * w + px, corner = 2 * dy + dx;
                    hit[slot].valid = true;
                    hit[slot].off = (d0 == 0 ? ou : ov) * W + (d0 == 0 ? ov : ou);
                    hit[slot].wgt = dy == 0 ? (dx == 0 ? smp.nw : smp.ne) : (dx == 0 ? smp.sw : smp.se);
                    hit[slot].key = ORDERED ? ((flat / lanes) * 4 + corner) * lanes + flat % lanes : 0;
                }
                const float* gyc = gy + (static_cast<int64_t>(copy) * planes + plane0) * plane_elems;
                if constexpr (ORDERED) {
                    // verification mode: compact, then insertion sort of <= 9 entries by ATen's visiting order
                    int off[9], key[9], hits = 0;
                    float wgt[9];
                    for (int slot = 0; slot < 9; ++slot)
                        if (hit[slot].valid) {
                            off[hits] = hit[slot].off;
                            key[hits] = hit[slot].key;
                            wgt[hits++] = hit[slot].wgt;
                        }
                    for (int a = 1; a < hits; ++a) {
                        const int ka = key[a], oa = off[a];
                        const float wa = wgt[a];
                        int b = a - 1;
                        while (b >= 0 && key[b] > ka) { key[b + 1] = key[b]; off[b + 1] = off[b]; wgt[b + 1] = wgt[b]; --b; }
                        key[b + 1] = ka;
                        off[b + 1] = oa;
                        wgt[b + 1] = wa;
                    }
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        float ck = 0.0f;
                        for (int a = 0; a < hits; ++a) ck += wgt[a] * gyc[p * plane_elems + off[a]];
                        acc[p] = copy == copies - 1 ? ck : acc[p] + ck;
                    }
                } else {
                    // slot by slot over the planes: uniform plane base + one 32-bit byte offset per slot (scalar-base
                    // addressing), a candidate that does not reach this pixel adds an exact zero (ck is never -0, so
                    // the sum is unchanged; the select keeps a non-finite gy of a foreign pixel out)
                    float ck[P];
#pragma unroll
                    for (int p = 0; p < P; ++p) ck[p] = 0.0f;
#pragma unroll
                    for (int slot = 0; slot < 9; ++slot) {
                        const unsigned boff = static_cast<unsigned>(hit[slot].off) * 4u;
#pragma unroll
                        for (int p = 0; p < P; ++p) {
                            const char* base = reinterpret_cast<const char*>(gyc + p * plane_elems);
                            const float g = *reinterpret_cast<const float*>(base + boff);
                            ck[p] += hit[slot].wgt * (hit[slot].valid ? g : 0.0f);
                        }
                    }
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[p] = copy == copies - 1 ? ck[p] : acc[p] + ck[p];
                }
            }
#pragma unroll
            for (int p = 0; p < P; ++p) {
                gx[static_cast<int64_t>(plane0 + p) * plane_elems + static_cast<int64_t>(sy) * W + sx] = acc[p];
                asum[p] += fabsf(acc[p]);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float total = block_sum(asum[p], red);
        if (ws != nullptr && threadIdx.x == 0) ws[static_cast<int64_t>(plane0 + p) * row_tiles + tile] = total;
    }
}

}  // namespace ta

using namespace ta;

static int check_bsr(const void* a, const void* b, const void* plan, int64_t planes, int h, int w, int copies, int nb) {
    TA_REQUIRE(a && b && plan && a != b, "null or aliased pointers");
    TA_REQUIRE(planes > 0 && h > 0 && w > 0 && copies > 0, "bad shape");
    TA_REQUIRE(nb >= 1 && nb <= kBsrMaxBlocks, "num_block %d outside 1..%d", nb, kBsrMaxBlocks);
    TA_REQUIRE(planes * ceil_div(h, kBsrBwdRows) * copies < (1ll << 31), "too many tiles");
    return 0;
}

extern "C" int64_t ta_bsr_tiles(int h) { return h > 0 ? ceil_div(h, kBsrBwdRows) : 0; }

// planes per thread: as many as divide the plane count while the launch still has >= `floor` workgroups
static int bsr_planes_per_thread(int64_t planes, int64_t tiles, int64_t floor, int most) {
    for (int p : {12, 6, 3, 2})
        if (p <= most && planes % p == 0 && planes / p * tiles >= floor) return p;
    return 1;
}

extern "C" int ta_bsr_fwd(const float* x, const int32_t* plan, float* y, int64_t planes, int h, int w, int copies, int nb,
                          void* stream) {
    if (int rc = check_bsr(x, y, plan, planes, h, w, copies, nb)) return rc;
    TA_REQUIRE(static_cast<int64_t>(h) * w < (1ll << 29), "plane too large");
    const int row_tiles = static_cast<int>(ceil_div(h, kBsrRows));
    const int pp = bsr_planes_per_thread(planes, static_cast<int64_t>(row_tiles) * copies, 2048, 12);
    const dim3 grid(static_cast<unsigned>(copies * (planes / pp) * row_tiles));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define TA_BSR_FWD(P) \
    hipLaunchKernelGGL((bsr_fwd_kernel<P>), grid, dim3(kBlock), 0, st, x, plan, y, static_cast<int>(planes), h, w, nb, row_tiles)
    if (pp == 12) TA_BSR_FWD(12);
    else if (pp == 6) TA_BSR_FWD(6);
    else if (pp == 3) TA_BSR_FWD(3);
    else if (pp == 2) TA_BSR_FWD(2);
    else TA_BSR_FWD(1);
#undef TA_BSR_FWD
    return check_launch("bsr_fwd");
}

extern "C" int ta_bsr_bwd(const float* gy, const int32_t* plan, float* gx, float* ws, int64_t planes, int h, int w, int copies,
                          int nb, void* stream) {
    if (int rc = check_bsr(gy, gx, plan, planes, h, w, copies, nb)) return rc;
    const int plan_ints = copies * bsr_plan_stride(nb);
    TA_REQUIRE(plan_ints <= kBsrMaxPlanInts, "copies * plan stride exceeds the %d ints staged in LDS", kBsrMaxPlanInts);
    TA_REQUIRE(static_cast<int64_t>(h) * w < (1ll << 29), "plane too large");
    const int row_tiles = static_cast<int>(ceil_div(h, kBsrBwdRows));
    const int lanes = sum_order_lanes();                          // verification mode (ta_set_sum_order): ATen's visiting order
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = static_cast<size_t>(plan_ints) * sizeof(int);
    const int pp = lanes ? 1 : bsr_planes_per_thread(planes, row_tiles, 768, 6);     // 12 planes: no faster than 6 (r2k)
    const dim3 grid(static_cast<unsigned>(planes / pp * row_tiles));
#define TA_BSR_BWD(P, ORD)                                                                                            \
    hipLaunchKernelGGL((bsr_bwd_kernel<P, ORD>), grid, dim3(kBlock), lds, st, gy, plan, gx, ws, static_cast<int>(planes), h, w, \
                       copies, nb, row_tiles, lanes)
    if (lanes) TA_BSR_BWD(1, true);
    else if (pp == 6) TA_BSR_BWD(6, false);
    else if (pp == 3) TA_BSR_BWD(3, false);
    else if (pp == 2) TA_BSR_BWD(2, false);
    else TA_BSR_BWD(1, false);
#undef TA_BSR_BWD
    return check_launch("bsr_bwd");
}
