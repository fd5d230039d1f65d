// Update stack of the iterative FGSM loop for gfx950 (MI355X):
//   K1  per-image sum|g|            (reference: grad.abs().mean(dim=(1,2,3)), attack.py:128)
//   K2  fused momentum + sign + alpha-step + eps-ball projection + image-box clamp (+ x_adv = x + delta' for the
//       next iteration)             (reference: attack.py:124-128 and 145-153, utils.py:68-69, attack.py:88)
// Both kernels tile an image into 3072-element workgroup tiles (150528 = 49 tiles): every lane keeps
// three 16-byte accesses per operand in flight, consecutive lanes touch consecutive 16 B (1 KiB per
// wave instruction).  K1 and K2 use the same (tile, image) -> blockIdx map so the second read of g
// comes from the same XCD's L2 / the Infinity Cache.  No float atomics: the partial sums are written
// per tile and re-added in tile order by every wave of K2, so results do not depend on scheduling.
//
// Arithmetic is kept in the reference's order with contraction off (-ffp-contract=off):
//   q = g / (sum|g| / E);  m' = m*decay + q;  d' = d + alpha*sign(m');  d' = min(max(d',-eps),eps);
//   d' = min(max(d', 0 - x), 1 - x)
#include <stdlib.h>
#include "update_common.h"
#include "philox.h"

namespace ta {

// ------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------
// G_STD: g is the gradient with respect to the NORMALISED input (the backbone's own input gradient gy); the sums are of
// |gy / std[c]| -- what ta_normalize_bwd would have summed while writing gx = gy / std[c] -- in the same per-thread order,
// so they carry ta_normalize_bwd's bits and the pass that stores gx disappears (ta_mi_update_std divides inline).
template <int VEC, bool HAS_V, bool SQUARE, bool G_STD = false>
__global__ __launch_bounds__(kBlock) void abs_sum_partials_kernel(const float* __restrict__ g,
                                                                  const float* __restrict__ v,
                                                                  float* __restrict__ ws, int64_t e,
                                                                  int tiles, const float* __restrict__ stdv = nullptr,
                                                                  int64_t hw = 1) {
    __shared__ float lds[kBlock / kWave];
    const int64_t img = blockIdx.y;
    const int64_t tile0 = static_cast<int64_t>(blockIdx.x) * kTile;
    const float* gi = g + img * e;
    const float* vi = HAS_V ? v + img * e : nullptr;
    constexpr int S = Slots<VEC>::n;
    Pack<VEC> a[S], b[S];
    bool full[S];
    const TileChannels<kTile> ch(G_STD ? tile0 : 0, G_STD ? hw : 1, e);
    const float s0 = G_STD ? stdv[ch.c_lo] : 1.0f, s1 = G_STD ? stdv[ch.c_hi] : 1.0f;
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int64_t off = tile0 + (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
        full[u] = off + VEC <= e;
        if (full[u]) {
            a[u].load(gi + off);
            if (HAS_V) b[u].load(vi + off);
        }
    }
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < S; ++u) {
        if (full[u]) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float x = a[u][k];
                if (G_STD) {
                    const int64_t off = tile0 + (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
                    x = x / ch.pick(stdv, s0, s1, off + k);
                }
                if (HAS_V) x = x + b[u][k];
                acc += SQUARE ? x * x : fabsf(x);
            }
        } else if (VEC > 1) {   // ragged end of an image whose size is not a multiple of VEC
            const int64_t off = tile0 + (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
            for (int64_t i = off; i < e && i < off + VEC; ++i) {
                float x = G_STD ? gi[i] / stdv[static_cast<int>(i / hw)] : gi[i];
                if (HAS_V) x = x + vi[i];
                acc += SQUARE ? x * x : fabsf(x);
            }
        }
    }
    const float total = block_sum(acc, lds);
    if (threadIdx.x == 0) ws[img * tiles + blockIdx.x] = total;
}

// ------------------------------------------------------------------------------------------------
// K1 in the REFERENCE's summation order (opt-in: ta_set_sum_order(8 | 16); the binding reads TA_ATEN_SUM_LANES).
// grad.abs().mean(dim=(1,2,3)) on the reference's CPU path is ATen's vectorised cascade sum (SumKernel.cpp; restated
// in oracle/ta_oracle.c: ta_oracle_aten_row_sum): `lanes` SIMD lanes x 4 interleaved accumulators = 4*lanes columns,
// each column summed in blocks of 16 steps, 16 block sums into a level-1 sum, and so on for 4 levels, then the levels,
// the 4 accumulators and the lanes folded in a fixed order.  Which `lanes` applies is a property of the CPU the
// reference runs on (8 = AVX2, 16 = AVX-512).  This kernel evaluates that very expression tree -- every partial sum
// has the same operands in the same order, only independent subtrees run in parallel -- so sum|g|, hence g / mean|g|,
// the momentum and every later iterate carry the reference's bits.  One workgroup per image; the result goes to slot 0
// of the image's partial-sum row and zeros to the other slots, so K2 / the momentum kernel are untouched.
// ------------------------------------------------------------------------------------------------
constexpr int kAtenIlp = 4, kAtenLevelPower = 4, kAtenStep = 1 << kAtenLevelPower;
constexpr int kAtenMaxLdsFloats = 16 * 1024;          // 64 KB of dynamic LDS: level-0 block sums are E / 16 floats (150528 -> 9408)

template <bool HAS_V>
__global__ __launch_bounds__(kBlock) void aten_order_abs_sum_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                                    float* __restrict__ ws, int64_t e, int tiles,
                                                                    int lanes, const float* __restrict__ stdv = nullptr,
                                                                    int64_t hw = 1) {
    __shared__ __attribute__((aligned(16))) float aten_lds[kAtenMaxLdsFloats];     // 64 KB; one workgroup per image
    const int64_t img = blockIdx.x;
    const float* gi = g + img * e;
    const float* vi = HAS_V ? v + img * e : nullptr;
    auto mag = [&](int64_t i) {          // stdv: the operand is gy, the gradient is gy / std[c] (ta_mi_update_std)
        const float gv = stdv != nullptr ? gi[i] / stdv[static_cast<int>(i / hw)] : gi[i];
        return fabsf(HAS_V ? gv + vi[i] : gv);
    };
    const int cols = lanes * kAtenIlp;
    const int64_t steps = e / cols;
    const int nb1 = static_cast<int>(steps >> kAtenLevelPower);                  // full level-0 blocks
    const int nb2 = nb1 >> kAtenLevelPower, nb3 = nb2 >> kAtenLevelPower;
    float* s1 = aten_lds;                                // [nb1][cols]
    float* s2 = s1 + static_cast<int64_t>(nb1) * cols;   // [nb2][cols]
    float* s3 = s2 + static_cast<int64_t>(nb2) * cols;   // [nb3][cols]
    float* tot = s3 + static_cast<int64_t>(nb3) * cols;  // [cols]
    // level 0: 16 consecutive steps of one column, left to right, starting from 0
    for (int idx = threadIdx.x; idx < nb1 * cols; idx += kBlock) {
        const int b = idx / cols, c = idx - b * cols;
        float acc = 0.0f;
        for (int j = 0; j < kAtenStep; ++j) acc += mag((static_cast<int64_t>(b) * kAtenStep + j) * cols + c);
        s1[idx] = acc;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < nb2 * cols; idx += kBlock) {              // level 1: 16 block sums
        const int d = idx / cols, c = idx - d * cols;
        float acc = 0.0f;
        for (int j = 0; j < kAtenStep; ++j) acc += s1[(d * kAtenStep + j) * cols + c];
        s2[idx] = acc;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < nb3 * cols; idx += kBlock) {              // level 2
        const int q = idx / cols, c = idx - q * cols;
        float acc = 0.0f;
        for (int j = 0; j < kAtenStep; ++j) acc += s2[(q * kAtenStep + j) * cols + c];
        s3[idx] = acc;
    }
    __syncthreads();
    if (threadIdx.x < cols) {
        const int c = threadIdx.x;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (int64_t i = static_cast<int64_t>(nb1) * kAtenStep; i < steps; ++i) a0 += mag(i * cols + c);   // steps past the last full block
        for (int b = nb2 * kAtenStep; b < nb1; ++b) a1 += s1[b * cols + c];      // block sums not yet passed up
        for (int d = nb3 * kAtenStep; d < nb2; ++d) a2 += s2[d * cols + c];
        for (int q = 0; q < nb3; ++q) a3 += s3[q * cols + c];                    // the top level is never passed on
        tot[c] = ((a0 + a1) + a2) + a3;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t vec_size = e / lanes;
        for (int64_t vv = steps * kAtenIlp; vv < vec_size; ++vv)                 // whole vectors beyond the last ILP group
            for (int l = 0; l < lanes; ++l) tot[l] += mag(vv * lanes + l);
        for (int k = 1; k < kAtenIlp; ++k)
            for (int l = 0; l < lanes; ++l) tot[l] += tot[k * lanes + l];
        float total = 0.0f;
        for (int64_t i = vec_size * lanes; i < e; ++i) total += mag(i);         // scalar tail first, then the lanes in order
        for (int l = 0; l < lanes; ++l) total += tot[l];
        ws[img * tiles] = total;
        for (int t = 1; t < tiles; ++t) ws[img * tiles + t] = 0.0f;
    }
}

// 0 = off (the kernels' own fixed order); 8 / 16 = ATen's cascade order for that SIMD width: ta_set_sum_order (runtime.hip)
static int aten_sum_lanes() { return sum_order_lanes(); }

// ------------------------------------------------------------------------------------------------
// Surrogate pre-processing Normalize (reference: transforms.Normalize inside PreprocessingModel,
// utils.py:72-79): y = (x - mean[c]) / std[c].  Its backward gx = gy / std[c] is the LAST kernel of the
// surrogate's backward pass, i.e. the producer of the gradient the update stack consumes -- so it also emits the
// per-tile sums of |gx| in K1's layout, and the separate K1 pass over g disappears (g is read once, by K2).
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void normalize_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ stdv, int64_t e, int64_t hw) {
    const int64_t img = blockIdx.y;
    const int64_t tile0 = static_cast<int64_t>(blockIdx.x) * kTile;
    constexpr int S = Slots<VEC>::n;
    const TileChannels<kTile> ch(tile0, hw, e);
    const float m0 = mean[ch.c_lo], m1 = mean[ch.c_hi], s0 = stdv[ch.c_lo], s1 = stdv[ch.c_hi];
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int64_t off = tile0 + (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
        if (off + VEC <= e) {
            Pack<VEC> a, o;
            a.load(x + img * e + off);
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                o[k] = (a[k] - ch.pick(mean, m0, m1, off + k)) / ch.pick(stdv, s0, s1, off + k);
            o.store(y + img * e + off);
        } else if (VEC > 1) {
            for (int64_t i = off; i < e && i < off + VEC; ++i) {
                const int c = static_cast<int>(i / hw);
                y[img * e + i] = (x[img * e + i] - mean[c]) / stdv[c];
            }
        }
    }
}

// The first kernel of an iteration when nothing but the surrogate's Normalize sits between `data + delta` (attack.py:88)
// and the backbone: y = ((x + d) - mean[c]) / std[c] straight from the perturbation and the image -- the image read as the
// PNG byte it was decoded from when the probe's flag says so (1 B instead of 4 B per element, as in the fused update).  The
// fused update then no longer stores x + d' for this kernel's sake (28 -> 24 B/element algorithmic, 25 -> 21 executed).
// Same rounding points as the add of attack.py:88 followed by Normalize: same bits.
template <int VEC, bool X_U8>
__global__ __launch_bounds__(kBlock) void normalize_adv_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ x_u8,
                                                                   const int* __restrict__ u8_mismatch,
                                                                   const float* __restrict__ delta, float* __restrict__ y,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ stdv, int64_t e, int64_t hw) {
    static_assert(!X_U8 || VEC == 4, "the byte source is read four at a time");
    const int64_t img = blockIdx.y;
    const int64_t tile0 = static_cast<int64_t>(blockIdx.x) * kTile;
    constexpr int S = Slots<VEC>::n;
    const bool bytes = X_U8 && uniform_int(*u8_mismatch) == 0;
    const TileChannels<kTile> ch(tile0, hw, e);
    const float m0 = mean[ch.c_lo], m1 = mean[ch.c_hi], s0 = stdv[ch.c_lo], s1 = stdv[ch.c_hi];
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int64_t off = tile0 + (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
        if (off + VEC <= e) {
            Pack<VEC> a, d, o;
            d.load(delta + img * e + off);
            if (X_U8 && bytes) {
                const uint32_t pb = *reinterpret_cast<const uint32_t*>(x_u8 + img * e + off);
#pragma unroll
                for (int k = 0; k < VEC; ++k) a[k] = u8_to_unit((pb >> (8 * k)) & 0xffu);
            } else {
                a.load(x + img * e + off);
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                o[k] = ((a[k] + d[k]) - ch.pick(mean, m0, m1, off + k)) / ch.pick(stdv, s0, s1, off + k);
            o.store(y + img * e + off);
        } else if (VEC > 1) {
            for (int64_t i = off; i < e && i < off + VEC; ++i) {
                const int c = static_cast<int>(i / hw);
                y[img * e + i] = ((x[img * e + i] + delta[img * e + i]) - mean[c]) / stdv[c];
            }
        }
    }
}

// The same pass for a surrogate that runs in NHWC memory (three-channel images): y[n][p][c] -- the backbone's first convolution
// reads it as it is, the NCHW -> NHWC copy torch would insert in front of it (146 us per iteration at batch 125, r6e) disappears.
// A lane owns four pixels: three 16-byte loads of delta, three 4-byte (or 16-byte) loads of the image, three 16-byte stores of
// twelve consecutive floats.  Same expression per element as above: same bits, other layout.
template <bool X_U8>
__global__ __launch_bounds__(kBlock) void normalize_adv_fwd_nhwc3_kernel(const float* __restrict__ x, const uint8_t* __restrict__ x_u8,
                                                                         const int* __restrict__ u8_mismatch,
                                                                         const float* __restrict__ delta, float* __restrict__ y,
                                                                         const float* __restrict__ mean,
                                                                         const float* __restrict__ stdv, int64_t hw) {
    const int64_t img = blockIdx.y;
    const int64_t p = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * 4;
    if (p >= hw) return;
    const bool bytes = X_U8 && uniform_int(*u8_mismatch) == 0;
    float o[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int64_t off = (img * 3 + c) * hw + p;
        Pack<4> a, d;
        d.load(delta + off);
        if (X_U8 && bytes) {
            const uint32_t pb = *reinterpret_cast<const uint32_t*>(x_u8 + off);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = u8_to_unit((pb >> (8 * k)) & 0xffu);
        } else {
            a.load(x + off);
        }
        const float m = mean[c], sd = stdv[c];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[c][k] = ((a[k] + d[k]) - m) / sd;
    }
    float* out = y + (img * hw + p) * 3;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        Pack<4> v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 4 * q + k;                     // i = 3 * pixel + channel
            v[k] = o[i % 3][i / 3];
        }
        v.store(out + 4 * q);
    }
}

template <int VEC, bool HAS_V>      // HAS_V: the tile sums are of |gx + v| (VMI-FGSM: the momentum normalises grad + variance)
__global__ __launch_bounds__(kBlock) void normalize_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                               const float* __restrict__ stdv,
                                                               const float* __restrict__ v,
                                                               float* __restrict__ ws, int64_t e, int64_t hw,
                                                               int tiles) {
    __shared__ float lds[kBlock / kWave];
    const int64_t img = blockIdx.y;
    const int64_t tile0 = static_cast<int64_t>(blockIdx.x) * kTile;
    constexpr int S = Slots<VEC>::n;
    float acc = 0.0f;
    const TileChannels<kTile> ch(tile0, hw, e);
    const float s0 = stdv[ch.c_lo], s1 = stdv[ch.c_hi];
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int64_t off = tile0 + (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
        if (off + VEC <= e) {
            Pack<VEC> a, o, b;
            a.load(gy + img * e + off);
            if (HAS_V) b.load(v + img * e + off);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                o[k] = a[k] / ch.pick(stdv, s0, s1, off + k);
                acc += fabsf(HAS_V ? o[k] + b[k] : o[k]);    // same per-thread order as abs_sum_partials_kernel
            }
            o.store(gx + img * e + off);
        } else if (VEC > 1) {
            for (int64_t i = off; i < e && i < off + VEC; ++i) {
                const float o = gy[img * e + i] / stdv[static_cast<int>(i / hw)];
                gx[img * e + i] = o;
                acc += fabsf(HAS_V ? o + v[img * e + i] : o);
            }
        }
    }
    const float total = block_sum(acc, lds);
    if (threadIdx.x == 0) ws[img * tiles + blockIdx.x] = total;
}

// VMI-FGSM's neighbour chain (vmifgsm.py:46-58) with the surrogate's Normalize folded into both ends: per neighbour
//   forward   y   = (((x + delta) + noise) - mean[c]) / std[c]         one pass, 12 B/element (noise from Philox)
//   backward  acc = acc + gy / std[c]          (acc = gy / std[c] for the first neighbour)      one pass, 12 (8)
// instead of sample (12) -> normalize (8) ... normalize backward (8) -> accumulate (12).  Every rounding point of the
// unfused chain is kept (add, add, subtract, divide; divide, add), so the bits are the same.
template <int VEC, bool HAS_NOISE>
__global__ __launch_bounds__(kBlock) void vmi_neighbor_norm_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ delta,
                                                                   const float* __restrict__ noise,
                                                                   float* __restrict__ y,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ stdv, float radius,
                                                                   uint64_t seed, uint64_t offset, int64_t e, int64_t hw) {
    const int64_t img = blockIdx.y;
    const int64_t tile0 = static_cast<int64_t>(blockIdx.x) * kTile;
    constexpr int S = Slots<VEC>::n;
    const TileChannels<kTile> ch(tile0, hw, e);
    const float m0 = mean[ch.c_lo], m1 = mean[ch.c_hi], s0 = stdv[ch.c_lo], s1 = stdv[ch.c_hi];
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int64_t off = tile0 + (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
        const int64_t at = img * e + off;                      // flat element index: the Philox counter of ta_vmi_neighbor
        if (off + VEC <= e) {
            Pack<VEC> a, d, r, o;
            a.load(x + at);
            d.load(delta + at);
            if (HAS_NOISE) {
                r.load(noise + at);
            } else if (VEC == 4) {
                const float4 q = uniform4(static_cast<uint64_t>(at) >> 2, seed, offset, radius);
                r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w;
            } else {
                r[0] = uniform1(static_cast<uint64_t>(at), seed, offset, radius);
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                o[k] = (((a[k] + d[k]) + r[k]) - ch.pick(mean, m0, m1, off + k)) / ch.pick(stdv, s0, s1, off + k);
            o.store(y + at);
        } else if (VEC > 1) {
            for (int64_t i = off; i < e && i < off + VEC; ++i) {
                const int c = static_cast<int>(i / hw);
                const float r = HAS_NOISE ? noise[img * e + i] : uniform1(static_cast<uint64_t>(img * e + i), seed, offset, radius);
                y[img * e + i] = (((x[img * e + i] + delta[img * e + i]) + r) - mean[c]) / stdv[c];
            }
        }
    }
}

template <int VEC, bool FIRST>
__global__ __launch_bounds__(kBlock) void normalize_bwd_accumulate_kernel(const float* __restrict__ gy, float* acc,
                                                                          const float* __restrict__ stdv, int64_t e,
                                                                          int64_t hw) {
    const int64_t img = blockIdx.y;
    const int64_t tile0 = static_cast<int64_t>(blockIdx.x) * kTile;
    constexpr int S = Slots<VEC>::n;
    const TileChannels<kTile> ch(tile0, hw, e);
    const float s0 = stdv[ch.c_lo], s1 = stdv[ch.c_hi];
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int64_t off = tile0 + (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
        if (off + VEC <= e) {
            Pack<VEC> a, o;
            a.load(gy + img * e + off);
            if (!FIRST) o.load(acc + img * e + off);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float g = a[k] / ch.pick(stdv, s0, s1, off + k);
                o[k] = FIRST ? g : o[k] + g;
            }
            o.store(acc + img * e + off);
        } else if (VEC > 1) {
            for (int64_t i = off; i < e && i < off + VEC; ++i) {
                const float g = gy[img * e + i] / stdv[static_cast<int>(i / hw)];
                acc[img * e + i] = FIRST ? g : acc[img * e + i] + g;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2 (fused) and the two hook-level halves
// ------------------------------------------------------------------------------------------------
// K2 has its own tiling (it only SUMS the K1-layout partials): on MI355X one 16-byte access per operand and lane
// with 512-lane workgroups streams fastest (tools/k2_sweep.hip: 6.19 TB/s at N=32, vs 5.72 for 256 x 3), and
// non-temporal loads/stores win once a launch moves more than the 256 MiB Infinity Cache holds (6.35 vs 5.45 TB/s
// at N=125) but lose below it -- NT is therefore a launch-time choice.
constexpr int kK2Block = 512;

//
// X_U8: the images of this path are PNG-decoded (utils.py:136: `image.astype(np.float32) / 255`), i.e. every x is
// float(k) / 255 for a byte k.  When the caller has verified that (ta_u8_source_probe: one pass per batch, a device-side
// flag, no host round trip) the kernel reads the byte instead of the float -- 1 B instead of 4 B per element of the
// 24 (+4) -- and rebuilds x with the bits of the IEEE division (u8_to_unit).  The flag is read by the kernel itself, so
// a batch that is NOT byte-valued (any other caller of the plug-in API) silently takes the fp32 operand: same launch,
// same result, no synchronisation.
//
// G_STD (round 5): g is gy, the gradient with respect to the NORMALISED input -- what the backbone's backward itself
// produces -- and the kernel forms the gradient of attack.py:118-122 inline, gy / std[c] (the IEEE division
// ta_normalize_bwd performed, utils.py:72-79's Normalize backward), so the 8 B/element pass that stored gx = gy / std[c]
// disappears; ws then holds sums of |gy / std[c]| (ta_stem7s2_input_grad leaves them, or K1 with G_STD).  Same rounding
// points, same bits as ta_normalize_bwd + ta_mi_update.
template <int VEC, int BLOCK, int SLOTS, bool NT, bool HAS_V, bool HAS_MIN, bool HAS_MOUT, bool HAS_XADV, bool X_U8 = false,
          bool G_STD = false>
__global__ __launch_bounds__(BLOCK) void mi_update_kernel(
    const float* __restrict__ g, const float* __restrict__ v, const float* m_in, float* m_out, float* delta,
    const float* __restrict__ x, float* __restrict__ x_adv, const float* __restrict__ ws, StepParams p, int64_t e,
    int tiles_ws, const uint8_t* __restrict__ x_u8 = nullptr, const int* __restrict__ u8_mismatch = nullptr,
    const float* __restrict__ stdv = nullptr, int hw = 1) {
    constexpr int TILE = BLOCK * VEC * SLOTS;
    static_assert(!X_U8 || VEC == 4, "the byte source is read four at a time");
    const int64_t img = blockIdx.y;
    const int64_t base = img * e + static_cast<int64_t>(blockIdx.x) * TILE;
    const int64_t left = e - static_cast<int64_t>(blockIdx.x) * TILE;   // elements of this image from tile start
    Pack<VEC> pg[SLOTS], pv[SLOTS], pm[SLOTS], pd[SLOTS], px[SLOTS];
    uint32_t pb[SLOTS];
    bool full[SLOTS];
    const bool bytes = X_U8 && uniform_int(*u8_mismatch) == 0;           // one scalar load, a scalar branch
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int64_t off = (static_cast<int64_t>(u) * BLOCK + threadIdx.x) * VEC;
        full[u] = off + VEC <= left;
        if (full[u]) {
            if (NT) {
                pg[u].load_nt(g + base + off);
                if (HAS_V) pv[u].load_nt(v + base + off);
                if (HAS_MIN) pm[u].load_nt(m_in + base + off);
                pd[u].load_nt(delta + base + off);
                if (X_U8 && bytes) pb[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(x_u8 + base + off));
                else px[u].load_nt(x + base + off);
            } else {
                pg[u].load(g + base + off);
                if (HAS_V) pv[u].load(v + base + off);
                if (HAS_MIN) pm[u].load(m_in + base + off);
                pd[u].load(delta + base + off);
                if (X_U8 && bytes) pb[u] = *reinterpret_cast<const uint32_t*>(x_u8 + base + off);
                else px[u].load(x + base + off);
            }
        }
    }
    // G_STD: the tile's (at most two) channels and their std -- one 32-bit scalar division and two scalar loads, in the
    // shadow of the operands' loads issued above
    const TileChannels<TILE> ch(G_STD ? static_cast<int64_t>(blockIdx.x) * TILE : 0, G_STD ? hw : 1, e);
    const float sd0 = G_STD ? stdv[ch.c_lo] : 1.0f, sd1 = G_STD ? stdv[ch.c_hi] : 1.0f;
    if (X_U8 && bytes) {
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
            if (full[u])
#pragma unroll
                for (int k = 0; k < VEC; ++k) px[u][k] = u8_to_unit((pb[u] >> (8 * k)) & 0xffu);
    }
    const float mean = image_total(ws, img, tiles_ws) / static_cast<float>(e);   // sum then div_ (ATen mean)
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int64_t off = (static_cast<int64_t>(u) * BLOCK + threadIdx.x) * VEC;
        if (full[u]) {
            Pack<VEC> om, od, oa;
            // G_STD: one channel per 16-byte access (hw % VEC == 0 is the launcher's condition for the vector form)
            const float sd = G_STD ? ch.pick(stdv, sd0, sd1, static_cast<int64_t>(blockIdx.x) * TILE + off) : 1.0f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float g0 = G_STD ? pg[u][k] / sd : pg[u][k];
                const float gg = HAS_V ? g0 + pv[u][k] : g0;
                const float q = gg / mean;
                const float mprev = HAS_MIN ? pm[u][k] : 0.0f;
                const float mn = mprev * p.decay + q;
                const float d = project(pd[u][k] + p.alpha * sign_of(mn), px[u][k], p.neg_eps, p.eps);
                om[k] = mn;
                od[k] = d;
                oa[k] = px[u][k] + d;
            }
            if (NT) {
                if (HAS_MOUT) om.store_nt(m_out + base + off);
                od.store_nt(delta + base + off);
            } else {
                if (HAS_MOUT) om.store(m_out + base + off);
                od.store(delta + base + off);
            }
            if (HAS_XADV) oa.store(x_adv + base + off);      // read again by the next kernel: keep cacheable
        } else if (VEC > 1) {
            for (int64_t i = off; i < left && i < off + VEC; ++i) {
                const float g0 = G_STD ? g[base + i] / stdv[static_cast<int>((static_cast<int64_t>(blockIdx.x) * TILE + i) / hw)]
                                       : g[base + i];
                const float gg = HAS_V ? g0 + v[base + i] : g0;
                const float q = gg / mean;
                const float mprev = HAS_MIN ? m_in[base + i] : 0.0f;
                const float mn = mprev * p.decay + q;
                const float xx = x[base + i];
                const float d = project(delta[base + i] + p.alpha * sign_of(mn), xx, p.neg_eps, p.eps);
                if (HAS_MOUT) m_out[base + i] = mn;
                delta[base + i] = d;
                if (HAS_XADV) x_adv[base + i] = xx + d;
            }
        }
    }
}

template <int VEC, bool HAS_V, bool HAS_MIN>
__global__ __launch_bounds__(kBlock) void momentum_kernel(const float* __restrict__ g,
                                                          const float* __restrict__ v,
                                                          const float* m_in, float* m_out,
                                                          const float* __restrict__ ws, float decay,
                                                          int64_t e, int tiles) {
    const int64_t img = blockIdx.y;
    const int64_t base = img * e + static_cast<int64_t>(blockIdx.x) * kTile;
    const int64_t left = e - static_cast<int64_t>(blockIdx.x) * kTile;
    const float mean = image_total(ws, img, tiles) / static_cast<float>(e);
    constexpr int S = Slots<VEC>::n;
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int64_t off = (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
        if (off + VEC <= left) {
            Pack<VEC> pg, pv, pm, om;
            pg.load(g + base + off);
            if (HAS_V) pv.load(v + base + off);
            if (HAS_MIN) pm.load(m_in + base + off);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float gg = HAS_V ? pg[k] + pv[k] : pg[k];
                om[k] = (HAS_MIN ? pm[k] : 0.0f) * decay + gg / mean;
            }
            om.store(m_out + base + off);
        } else if (VEC > 1) {
            for (int64_t i = off; i < left && i < off + VEC; ++i) {
                const float gg = HAS_V ? g[base + i] + v[base + i] : g[base + i];
                m_out[base + i] = (HAS_MIN ? m_in[base + i] : 0.0f) * decay + gg / mean;
            }
        }
    }
}

template <int VEC, bool ALPHA_T, bool HAS_XADV>
__global__ __launch_bounds__(kBlock) void update_delta_linf_kernel(
    const float* delta_in, const float* __restrict__ x, const float* __restrict__ m,
    const float* __restrict__ alpha_t, float* delta_out, float* __restrict__ x_adv, float alpha,
    float neg_eps, float eps, int64_t numel) {
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
    constexpr int S = Slots<VEC>::n;
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int64_t off = base + (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
        if (off + VEC <= numel) {
            Pack<VEC> pd, px, pm, pa, od, oa;
            pd.load(delta_in + off);
            px.load(x + off);
            pm.load(m + off);
            if (ALPHA_T) pa.load(alpha_t + off);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float a = ALPHA_T ? pa[k] : alpha;
                od[k] = project(pd[k] + a * sign_of(pm[k]), px[k], neg_eps, eps);
                oa[k] = px[k] + od[k];
            }
            od.store(delta_out + off);
            if (HAS_XADV) oa.store(x_adv + off);
        } else if (VEC > 1) {
            for (int64_t i = off; i < numel && i < off + VEC; ++i) {
                const float a = ALPHA_T ? alpha_t[i] : alpha;
                const float d = project(delta_in[i] + a * sign_of(m[i]), x[i], neg_eps, eps);
                delta_out[i] = d;
                if (HAS_XADV) x_adv[i] = x[i] + d;
            }
        }
    }
}

// L2 branch, pass B: d' = d + (g / (|g|_2 + 1e-20)) * alpha, plus per-tile sum of d'^2 into ws2
template <int VEC>
__global__ __launch_bounds__(kBlock) void l2_step_kernel(const float* delta_in,
                                                         const float* __restrict__ g, float* delta_out,
                                                         const float* __restrict__ ws_g2,
                                                         float* __restrict__ ws_d2, float alpha,
                                                         int64_t e, int tiles) {
    __shared__ float lds[kBlock / kWave];
    const int64_t img = blockIdx.y;
    const int64_t base = img * e + static_cast<int64_t>(blockIdx.x) * kTile;
    const int64_t left = e - static_cast<int64_t>(blockIdx.x) * kTile;
    const float gnorm = sqrtf(image_total(ws_g2, img, tiles)) + 1e-20f;
    constexpr int S = Slots<VEC>::n;
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int64_t off = (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
        if (off + VEC <= left) {
            Pack<VEC> pd, pg, od;
            pd.load(delta_in + base + off);
            pg.load(g + base + off);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                od[k] = pd[k] + (pg[k] / gnorm) * alpha;
                acc += od[k] * od[k];
            }
            od.store(delta_out + base + off);
        } else if (VEC > 1) {
            for (int64_t i = off; i < left && i < off + VEC; ++i) {
                const float d = delta_in[base + i] + (g[base + i] / gnorm) * alpha;
                delta_out[base + i] = d;
                acc += d * d;
            }
        }
    }
    const float total = block_sum(acc, lds);
    if (threadIdx.x == 0) ws_d2[img * tiles + blockIdx.x] = total;
}

// L2 branch, pass C: renorm(p=2, maxnorm=eps) then the image box
template <int VEC>
__global__ __launch_bounds__(kBlock) void l2_renorm_kernel(float* delta, const float* __restrict__ x,
                                                           const float* __restrict__ ws_d2, float eps,
                                                           int64_t e, int tiles) {
    const int64_t img = blockIdx.y;
    const int64_t base = img * e + static_cast<int64_t>(blockIdx.x) * kTile;
    const int64_t left = e - static_cast<int64_t>(blockIdx.x) * kTile;
    const float norm = sqrtf(image_total(ws_d2, img, tiles));
    const float scale = norm > eps ? eps / (norm + 1e-7f) : 1.0f;     // at::renorm
    constexpr int S = Slots<VEC>::n;
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int64_t off = (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
        if (off + VEC <= left) {
            Pack<VEC> pd, px, od;
            pd.load(delta + base + off);
            px.load(x + base + off);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float d = pd[k];
                if (norm > eps) d = d * scale;
                od[k] = fminf(fmaxf(d, 0.0f - px[k]), 1.0f - px[k]);
            }
            od.store(delta + base + off);
        } else if (VEC > 1) {
            for (int64_t i = off; i < left && i < off + VEC; ++i) {
                float d = delta[base + i];
                if (norm > eps) d = d * scale;
                const float xx = x[base + i];
                delta[base + i] = fminf(fmaxf(d, 0.0f - xx), 1.0f - xx);
            }
        }
    }
}

template <typename F> int dispatch_bool(bool b, F&& f) { return b ? f(std::true_type{}) : f(std::false_type{}); }

static bool vec_ok(int64_t e, std::initializer_list<const void*> ptrs) {
    if (e % kVec != 0) return false;
    for (const void* p : ptrs)
        if (p != nullptr && !aligned16(p)) return false;
    return true;
}

static int launch_partials(const float* g, const float* v, float* ws, int64_t n, int64_t e, bool square,
                           hipStream_t st, hipEvent_t ev_start = nullptr, const float* stdv = nullptr, int64_t hw = 1) {
    const hipEvent_t no_event = nullptr;
    const int tiles = static_cast<int>(ceil_div(e, kTile));
    if (const int lanes = square ? 0 : aten_sum_lanes()) {
        const int cols = lanes * kAtenIlp;
        const int64_t steps = e / cols, nb1 = steps >> kAtenLevelPower;
        const int64_t floats = (nb1 + (nb1 >> kAtenLevelPower) + (nb1 >> (2 * kAtenLevelPower)) + 1) * cols;
        // the cascade uses 16-step levels as long as ceil(log2(steps)) / 4 <= 4, i.e. up to 2^19 steps
        TA_REQUIRE(floats <= kAtenMaxLdsFloats && steps <= (1ll << 19),
                   "ta_set_sum_order: images of %lld elements exceed what the reference-order sum stages in LDS", (long long)e);
        if (v)
            TA_LAUNCH_TIMED(aten_order_abs_sum_kernel<true>, dim3(static_cast<unsigned>(n)), dim3(kBlock), st, ev_start,
                            no_event, g, v, ws, e, tiles, lanes, stdv, hw);
        else
            TA_LAUNCH_TIMED(aten_order_abs_sum_kernel<false>, dim3(static_cast<unsigned>(n)), dim3(kBlock), st, ev_start,
                            no_event, g, v, ws, e, tiles, lanes, stdv, hw);
        return check_launch("aten_order_abs_sum");
    }
    const dim3 grid(tiles, static_cast<unsigned>(n));
    const bool vec = vec_ok(e, {g, v}) && (stdv == nullptr || hw % kVec == 0);
    if (stdv != nullptr) {                      // sums of |g / std[c]| (never squared, never with a variance term)
        if (vec)
            TA_LAUNCH_TIMED((abs_sum_partials_kernel<4, false, false, true>), grid, dim3(kBlock), st, ev_start, no_event, g, v, ws,
                            e, tiles, stdv, hw);
        else
            TA_LAUNCH_TIMED((abs_sum_partials_kernel<1, false, false, true>), grid, dim3(kBlock), st, ev_start, no_event, g, v, ws,
                            e, tiles, stdv, hw);
        return check_launch("abs_sum_partials (std)");
    }
#define TA_K1(VEC, HV, SQ) \
    TA_LAUNCH_TIMED((abs_sum_partials_kernel<VEC, HV, SQ>), grid, dim3(kBlock), st, ev_start, no_event, g, v, ws, e, tiles, \
                    static_cast<const float*>(nullptr), static_cast<int64_t>(1))
    if (vec) {
        if (square) { TA_K1(4, false, true); }
        else if (v) { TA_K1(4, true, false); }
        else { TA_K1(4, false, false); }
    } else {
        if (square) { TA_K1(1, false, true); }
        else if (v) { TA_K1(1, true, false); }
        else { TA_K1(1, false, false); }
    }
#undef TA_K1
    return check_launch("abs_sum_partials");
}

}  // namespace ta

using namespace ta;

static int check_batch(int64_t n, int64_t e) {
    TA_REQUIRE(n > 0 && e > 0, "batch (n=%lld, e=%lld) must be positive", (long long)n, (long long)e);
    TA_REQUIRE(n <= 65535, "n=%lld exceeds the 65535 images one launch addresses", (long long)n);
    TA_REQUIRE(ceil_div(e, kTile) < (1ll << 31), "image too large");
    return 0;
}

// the kernels that index per-channel constants address an image with 32-bit offsets
static int check_planes(int64_t n, int64_t e) {
    if (int rc = check_batch(n, e)) return rc;
    TA_REQUIRE(e < (1ll << 31), "images of %lld elements exceed the 2^31 the per-channel kernels address", (long long)e);
    return 0;
}

extern "C" int64_t ta_update_tiles(int64_t e) { return e > 0 ? ceil_div(e, kTile) : 0; }

extern "C" int64_t ta_l1_workspace_floats(int64_t n, int64_t e) {
    if (n <= 0 || e <= 0) return 0;
    return 2 * n * ceil_div(e, kTile);    // two regions: the L2 branch needs |g|^2 and |d'|^2 partials
}

extern "C" int ta_abs_sum_partials(const float* g, const float* v, float* ws, int64_t n, int64_t e,
                                   void* stream) {
    if (int rc = check_batch(n, e)) return rc;
    TA_REQUIRE(g && ws, "null pointer");
    return launch_partials(g, v, ws, n, e, false, static_cast<hipStream_t>(stream));
}

extern "C" int ta_momentum(const float* g, const float* v, const float* m_in, float* m_out, float* ws,
                           float decay, int64_t n, int64_t e, void* stream) {
    if (int rc = check_batch(n, e)) return rc;
    TA_REQUIRE(g && m_out && ws, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = launch_partials(g, v, ws, n, e, false, st)) return rc;
    const int tiles = static_cast<int>(ceil_div(e, kTile));
    const dim3 grid(tiles, static_cast<unsigned>(n));
    const bool vec = vec_ok(e, {g, v, m_in, m_out});
#define TA_MOM(VEC, HV, HM) \
    hipLaunchKernelGGL((momentum_kernel<VEC, HV, HM>), grid, dim3(kBlock), 0, st, g, v, m_in, m_out, ws, decay, e, tiles)
    if (vec) {
        if (v && m_in) { TA_MOM(4, true, true); } else if (v) { TA_MOM(4, true, false); }
        else if (m_in) { TA_MOM(4, false, true); } else { TA_MOM(4, false, false); }
    } else {
        if (v && m_in) { TA_MOM(1, true, true); } else if (v) { TA_MOM(1, true, false); }
        else if (m_in) { TA_MOM(1, false, true); } else { TA_MOM(1, false, false); }
    }
#undef TA_MOM
    return check_launch("momentum");
}

extern "C" int ta_update_delta_linf(const float* delta_in, const float* x, const float* m, float alpha,
                                    const float* alpha_t, float eps, float* delta_out, float* x_adv,
                                    int64_t numel, void* stream) {
    TA_REQUIRE(numel > 0, "numel must be positive");
    TA_REQUIRE(delta_in && x && m && delta_out, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(static_cast<unsigned>(ceil_div(numel, kTile)));
    const bool vec = vec_ok(0, {delta_in, x, m, alpha_t, delta_out, x_adv});
#define TA_UPD(VEC, AT, XA)                                                                          \
    hipLaunchKernelGGL((update_delta_linf_kernel<VEC, AT, XA>), grid, dim3(kBlock), 0, st, delta_in, x, \
                       m, alpha_t, delta_out, x_adv, alpha, -eps, eps, numel)
    if (vec) {
        if (alpha_t && x_adv) { TA_UPD(4, true, true); } else if (alpha_t) { TA_UPD(4, true, false); }
        else if (x_adv) { TA_UPD(4, false, true); } else { TA_UPD(4, false, false); }
    } else {
        if (alpha_t && x_adv) { TA_UPD(1, true, true); } else if (alpha_t) { TA_UPD(1, true, false); }
        else if (x_adv) { TA_UPD(1, false, true); } else { TA_UPD(1, false, false); }
    }
#undef TA_UPD
    return check_launch("update_delta_linf");
}

extern "C" int ta_update_delta_l2(const float* delta_in, const float* x, const float* g, float alpha,
                                  float eps, float* delta_out, float* ws, int64_t n, int64_t e,
                                  void* stream) {
    if (int rc = check_batch(n, e)) return rc;
    TA_REQUIRE(delta_in && x && g && delta_out && ws, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tiles = static_cast<int>(ceil_div(e, kTile));
    float* ws_g2 = ws;
    float* ws_d2 = ws + n * tiles;
    if (int rc = launch_partials(g, nullptr, ws_g2, n, e, true, st)) return rc;
    const dim3 grid(tiles, static_cast<unsigned>(n));
    const bool vec = vec_ok(e, {delta_in, x, g, delta_out});
    if (vec)
        hipLaunchKernelGGL((l2_step_kernel<4>), grid, dim3(kBlock), 0, st, delta_in, g, delta_out, ws_g2, ws_d2, alpha, e, tiles);
    else
        hipLaunchKernelGGL((l2_step_kernel<1>), grid, dim3(kBlock), 0, st, delta_in, g, delta_out, ws_g2, ws_d2, alpha, e, tiles);
    if (int rc = check_launch("l2_step")) return rc;
    if (vec)
        hipLaunchKernelGGL((l2_renorm_kernel<4>), grid, dim3(kBlock), 0, st, delta_out, x, ws_d2, eps, e, tiles);
    else
        hipLaunchKernelGGL((l2_renorm_kernel<1>), grid, dim3(kBlock), 0, st, delta_out, x, ws_d2, eps, e, tiles);
    return check_launch("l2_renorm");
}

extern "C" int ta_normalize_fwd(const float* x, float* y, const float* mean, const float* stdv, int64_t n, int c,
                                int64_t hw, void* stream) {
    const int64_t e = static_cast<int64_t>(c) * hw;
    if (int rc = check_planes(n, e)) return rc;
    TA_REQUIRE(x && y && mean && stdv && c > 0, "null pointer");
    const dim3 grid(static_cast<unsigned>(ceil_div(e, kTile)), static_cast<unsigned>(n));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (vec_ok(e, {x, y}))
        hipLaunchKernelGGL(normalize_fwd_kernel<4>, grid, dim3(kBlock), 0, st, x, y, mean, stdv, e, hw);
    else
        hipLaunchKernelGGL(normalize_fwd_kernel<1>, grid, dim3(kBlock), 0, st, x, y, mean, stdv, e, hw);
    return check_launch("normalize_fwd");
}

extern "C" int ta_normalize_bwd(const float* gy, float* gx, const float* stdv, const float* v, float* ws, int64_t n, int c,
                                int64_t hw, void* stream) {
    const int64_t e = static_cast<int64_t>(c) * hw;
    if (int rc = check_planes(n, e)) return rc;
    TA_REQUIRE(gy && gx && stdv && ws && c > 0, "null pointer");
    const int tiles = static_cast<int>(ceil_div(e, kTile));
    const dim3 grid(tiles, static_cast<unsigned>(n));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define TA_NB(VEC, HV) hipLaunchKernelGGL((normalize_bwd_kernel<VEC, HV>), grid, dim3(kBlock), 0, st, gy, gx, stdv, v, ws, e, hw, tiles)
    if (vec_ok(e, {gy, gx, v})) { if (v) { TA_NB(4, true); } else { TA_NB(4, false); } }
    else { if (v) { TA_NB(1, true); } else { TA_NB(1, false); } }
#undef TA_NB
    return check_launch("normalize_bwd");
}

extern "C" int ta_vmi_neighbor_normalized(const float* x, const float* delta, const float* noise, float* y, const float* mean,
                                          const float* stdv, float radius, uint64_t seed, uint64_t offset, int64_t n, int c,
                                          int64_t hw, void* stream) {
    const int64_t e = static_cast<int64_t>(c) * hw;
    if (int rc = check_planes(n, e)) return rc;
    TA_REQUIRE(x && delta && y && mean && stdv && c > 0, "null pointer");
    const dim3 grid(static_cast<unsigned>(ceil_div(e, kTile)), static_cast<unsigned>(n));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define TA_VNN(VEC, HN) \
    hipLaunchKernelGGL((vmi_neighbor_norm_kernel<VEC, HN>), grid, dim3(kBlock), 0, st, x, delta, noise, y, mean, stdv, radius, seed, offset, e, hw)
    if (vec_ok(e, {x, delta, noise, y})) { if (noise) { TA_VNN(4, true); } else { TA_VNN(4, false); } }
    else { if (noise) { TA_VNN(1, true); } else { TA_VNN(1, false); } }
#undef TA_VNN
    return check_launch("vmi_neighbor_normalized");
}

extern "C" int ta_normalize_bwd_accumulate(const float* gy, float* acc, const float* stdv, int first, int64_t n, int c,
                                           int64_t hw, void* stream) {
    const int64_t e = static_cast<int64_t>(c) * hw;
    if (int rc = check_planes(n, e)) return rc;
    TA_REQUIRE(gy && acc && stdv && c > 0 && gy != acc, "null or aliased pointers");
    const dim3 grid(static_cast<unsigned>(ceil_div(e, kTile)), static_cast<unsigned>(n));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define TA_NBA(VEC, F) hipLaunchKernelGGL((normalize_bwd_accumulate_kernel<VEC, F>), grid, dim3(kBlock), 0, st, gy, acc, stdv, e, hw)
    if (vec_ok(e, {gy, acc})) { if (first) { TA_NBA(4, true); } else { TA_NBA(4, false); } }
    else { if (first) { TA_NBA(1, true); } else { TA_NBA(1, false); } }
#undef TA_NBA
    return check_launch("normalize_bwd_accumulate");
}

static int mi_update_impl(const float* g, const float* v, const float* m_in, float* m_out, float* delta,
                          const float* x, const uint8_t* x_u8, const int* u8_mismatch, float* x_adv, float* ws, int ws_slots,
                          float decay, float alpha, float eps, int64_t n, int64_t e, void* stream,
                          const float* stdv = nullptr, int64_t hw = 1) {
    if (int rc = check_batch(n, e)) return rc;
    TA_REQUIRE(stdv == nullptr || (v == nullptr && x_adv == nullptr && hw > 0 && e % hw == 0 && hw < (1ll << 31) && e < (1ll << 31)),
               "the std form of the update takes neither a variance term nor an x_adv buffer; e = channels * hw");
    TA_REQUIRE(g && delta && x && ws && ws_slots >= 0, "null pointer");
    TA_REQUIRE((x_u8 == nullptr) == (u8_mismatch == nullptr), "x_u8 and its probe flag come together");
    // ws_slots > 0 with a variance term: the producer summed |g + v| (ta_normalize_bwd with v) -- the caller's contract
    TA_REQUIRE(!(ws_slots && aten_sum_lanes() != 0),
               "ta_set_sum_order: the reference-order sum is never taken from a producer (pass ws_slots = 0)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const LaunchEvents timed = claim_launch_events();          // null unless ta_timing_begin armed them
    if (ws_slots == 0)
        if (int rc = launch_partials(g, v, ws, n, e, false, st, timed.start, stdv, hw)) return rc;
    const hipEvent_t k2_start = ws_slots == 0 ? nullptr : timed.start;
    const int tiles_ws = ws_slots > 0 ? ws_slots : static_cast<int>(ceil_div(e, kTile));
    const StepParams p{decay, alpha, -eps, eps};
    const bool vec = vec_ok(e, {g, v, m_in, m_out, delta, x, x_adv});
    // stream past the caches only when one launch moves more than the Infinity Cache can hold
    bool nt = vec && static_cast<double>(n) * static_cast<double>(e) * 24.0 > 256.0 * 1024 * 1024;
    const int key = (v ? 8 : 0) | (m_in ? 4 : 0) | (m_out ? 2 : 0) | (x_adv ? 1 : 0);
    // the byte source: 16-byte aligned floats, 4-byte aligned bytes, whole images of a multiple of 4 elements
    const bool u8 = x_u8 != nullptr && vec && (reinterpret_cast<uintptr_t>(x_u8) & 3u) == 0;
    if (stdv != nullptr) {
        // g = gy / std[c] formed inline.  Vector form: one channel per 16-byte access (hw % 4 == 0)
        const bool vec_std = vec && hw % kVec == 0;
        const int ihw = static_cast<int>(hw);
#define TA_MIS(VEC, BLOCK, SLOTS, NT, HMI, HMO, U8)                                                              \
    TA_LAUNCH_TIMED((mi_update_kernel<VEC, BLOCK, SLOTS, NT, false, HMI, HMO, false, U8, true>),                  \
                    dim3(static_cast<unsigned>(ceil_div(e, BLOCK * VEC * SLOTS)), static_cast<unsigned>(n)),      \
                    dim3(BLOCK), st, k2_start, timed.stop, g, v, m_in, m_out, delta, x, x_adv, ws, p, e, tiles_ws, \
                    x_u8, u8_mismatch, stdv, ihw)
#define TA_MIS_CASES(VEC, BLOCK, SLOTS, NT, U8)                                    \
    switch ((m_in ? 2 : 0) | (m_out ? 1 : 0)) {                                     \
        case 0: TA_MIS(VEC, BLOCK, SLOTS, NT, false, false, U8); break;             \
        case 1: TA_MIS(VEC, BLOCK, SLOTS, NT, false, true, U8); break;              \
        case 2: TA_MIS(VEC, BLOCK, SLOTS, NT, true, false, U8); break;              \
        default: TA_MIS(VEC, BLOCK, SLOTS, NT, true, true, U8); break;              \
    }
        if (vec_std && u8) { if (nt) { TA_MIS_CASES(4, kK2Block, 1, true, true) } else { TA_MIS_CASES(4, kK2Block, 1, false, true) } }
        else if (vec_std) { if (nt) { TA_MIS_CASES(4, kK2Block, 1, true, false) } else { TA_MIS_CASES(4, kK2Block, 1, false, false) } }
        else { TA_MIS_CASES(1, kBlock, kTile / kBlock, false, false) }
#undef TA_MIS_CASES
#undef TA_MIS
        return check_launch("mi_update (std form)");
    }
    if (u8) {
#define TA_MI8(NT, HV, HMI, HMO, HXA)                                                                            \
    TA_LAUNCH_TIMED((mi_update_kernel<4, kK2Block, 1, NT, HV, HMI, HMO, HXA, true>),                             \
                    dim3(static_cast<unsigned>(ceil_div(e, kK2Block * 4)), static_cast<unsigned>(n)),             \
                    dim3(kK2Block), st, k2_start, timed.stop, g, v, m_in, m_out, delta, x, x_adv, ws, p, e, tiles_ws, \
                    x_u8, u8_mismatch, static_cast<const float*>(nullptr), 1)
#define TA_MI8_CASES(NT)                                                  \
    switch (key) {                                                        \
        case 0: TA_MI8(NT, false, false, false, false); break;            \
        case 1: TA_MI8(NT, false, false, false, true); break;             \
        case 2: TA_MI8(NT, false, false, true, false); break;             \
        case 3: TA_MI8(NT, false, false, true, true); break;              \
        case 4: TA_MI8(NT, false, true, false, false); break;             \
        case 5: TA_MI8(NT, false, true, false, true); break;              \
        case 6: TA_MI8(NT, false, true, true, false); break;              \
        case 7: TA_MI8(NT, false, true, true, true); break;               \
        case 8: TA_MI8(NT, true, false, false, false); break;             \
        case 9: TA_MI8(NT, true, false, false, true); break;              \
        case 10: TA_MI8(NT, true, false, true, false); break;             \
        case 11: TA_MI8(NT, true, false, true, true); break;              \
        case 12: TA_MI8(NT, true, true, false, false); break;             \
        case 13: TA_MI8(NT, true, true, false, true); break;              \
        case 14: TA_MI8(NT, true, true, true, false); break;              \
        default: TA_MI8(NT, true, true, true, true); break;               \
    }
        if (nt) { TA_MI8_CASES(true) } else { TA_MI8_CASES(false) }
#undef TA_MI8_CASES
#undef TA_MI8
        return check_launch("mi_update (byte source)");
    }
#define TA_MI(VEC, BLOCK, SLOTS, NT, HV, HMI, HMO, HXA)                                                         \
    TA_LAUNCH_TIMED((mi_update_kernel<VEC, BLOCK, SLOTS, NT, HV, HMI, HMO, HXA>),                               \
                    dim3(static_cast<unsigned>(ceil_div(e, BLOCK * VEC * SLOTS)), static_cast<unsigned>(n)),     \
                    dim3(BLOCK), st, k2_start, timed.stop, g, v, m_in, m_out, delta, x, x_adv, ws, p, e, tiles_ws,    \
                    static_cast<const uint8_t*>(nullptr), static_cast<const int*>(nullptr),                      \
                    static_cast<const float*>(nullptr), 1)
#define TA_MI_CASES(VEC, BLOCK, SLOTS, NT)                                   \
    switch (key) {                                                           \
        case 0: TA_MI(VEC, BLOCK, SLOTS, NT, false, false, false, false); break; \
        case 1: TA_MI(VEC, BLOCK, SLOTS, NT, false, false, false, true); break;  \
        case 2: TA_MI(VEC, BLOCK, SLOTS, NT, false, false, true, false); break;  \
        case 3: TA_MI(VEC, BLOCK, SLOTS, NT, false, false, true, true); break;   \
        case 4: TA_MI(VEC, BLOCK, SLOTS, NT, false, true, false, false); break;  \
        case 5: TA_MI(VEC, BLOCK, SLOTS, NT, false, true, false, true); break;   \
        case 6: TA_MI(VEC, BLOCK, SLOTS, NT, false, true, true, false); break;   \
        case 7: TA_MI(VEC, BLOCK, SLOTS, NT, false, true, true, true); break;    \
        case 8: TA_MI(VEC, BLOCK, SLOTS, NT, true, false, false, false); break;  \
        case 9: TA_MI(VEC, BLOCK, SLOTS, NT, true, false, false, true); break;   \
        case 10: TA_MI(VEC, BLOCK, SLOTS, NT, true, false, true, false); break;  \
        case 11: TA_MI(VEC, BLOCK, SLOTS, NT, true, false, true, true); break;   \
        case 12: TA_MI(VEC, BLOCK, SLOTS, NT, true, true, false, false); break;  \
        case 13: TA_MI(VEC, BLOCK, SLOTS, NT, true, true, false, true); break;   \
        case 14: TA_MI(VEC, BLOCK, SLOTS, NT, true, true, true, false); break;   \
        default: TA_MI(VEC, BLOCK, SLOTS, NT, true, true, true, true); break;    \
    }
    if (vec && nt) { TA_MI_CASES(4, kK2Block, 1, true) }
    else if (vec) { TA_MI_CASES(4, kK2Block, 1, false) }
    else { TA_MI_CASES(1, kBlock, kTile / kBlock, false) }
#undef TA_MI_CASES
#undef TA_MI
    return check_launch("mi_update");
}

extern "C" int ta_mi_update(const float* g, const float* v, const float* m_in, float* m_out, float* delta,
                            const float* x, float* x_adv, float* ws, int ws_slots, float decay, float alpha,
                            float eps, int64_t n, int64_t e, void* stream) {
    return mi_update_impl(g, v, m_in, m_out, delta, x, nullptr, nullptr, x_adv, ws, ws_slots, decay, alpha, eps, n, e, stream);
}

extern "C" int ta_mi_update_u8(const float* g, const float* v, const float* m_in, float* m_out, float* delta,
                               const float* x, const uint8_t* x_u8, const int* u8_mismatch, float* x_adv, float* ws,
                               int ws_slots, float decay, float alpha, float eps, int64_t n, int64_t e, void* stream) {
    TA_REQUIRE(x_u8 && u8_mismatch, "null byte source (use ta_mi_update)");
    return mi_update_impl(g, v, m_in, m_out, delta, x, x_u8, u8_mismatch, x_adv, ws, ws_slots, decay, alpha, eps, n, e, stream);
}

extern "C" int ta_mi_update_std(const float* gy, const float* stdv, const float* m_in, float* m_out, float* delta, const float* x,
                                const uint8_t* x_u8, const int* u8_mismatch, float* ws, int ws_slots, float decay, float alpha,
                                float eps, int64_t n, int c, int64_t hw, void* stream) {
    TA_REQUIRE(stdv && c > 0 && hw > 0, "null std or empty planes");
    return mi_update_impl(gy, nullptr, m_in, m_out, delta, x, x_u8, u8_mismatch, nullptr, ws, ws_slots, decay, alpha, eps, n,
                          static_cast<int64_t>(c) * hw, stream, stdv, hw);
}

extern "C" int ta_abs_sum_partials_std(const float* gy, const float* stdv, float* ws, int64_t n, int c, int64_t hw, void* stream) {
    const int64_t e = static_cast<int64_t>(c) * hw;
    if (int rc = check_planes(n, e)) return rc;
    TA_REQUIRE(gy && stdv && ws && c > 0 && hw > 0, "null pointer");
    return launch_partials(gy, nullptr, ws, n, e, false, static_cast<hipStream_t>(stream), nullptr, stdv, hw);
}

extern "C" int ta_normalize_adv_fwd(const float* x, const uint8_t* x_u8, const int* u8_mismatch, const float* delta, float* y,
                                    const float* mean, const float* stdv, int64_t n, int c, int64_t hw, void* stream) {
    const int64_t e = static_cast<int64_t>(c) * hw;
    if (int rc = check_planes(n, e)) return rc;
    TA_REQUIRE(x && delta && y && mean && stdv && c > 0, "null pointer");
    TA_REQUIRE((x_u8 == nullptr) == (u8_mismatch == nullptr), "x_u8 and its probe flag come together");
    const dim3 grid(static_cast<unsigned>(ceil_div(e, kTile)), static_cast<unsigned>(n));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = vec_ok(e, {x, delta, y});
    if (vec && x_u8 != nullptr && (reinterpret_cast<uintptr_t>(x_u8) & 3u) == 0)
        hipLaunchKernelGGL((normalize_adv_fwd_kernel<4, true>), grid, dim3(kBlock), 0, st, x, x_u8, u8_mismatch, delta, y, mean, stdv, e, hw);
    else if (vec)
        hipLaunchKernelGGL((normalize_adv_fwd_kernel<4, false>), grid, dim3(kBlock), 0, st, x, x_u8, u8_mismatch, delta, y, mean, stdv, e, hw);
    else
        hipLaunchKernelGGL((normalize_adv_fwd_kernel<1, false>), grid, dim3(kBlock), 0, st, x, x_u8, u8_mismatch, delta, y, mean, stdv, e, hw);
    return check_launch("normalize_adv_fwd");
}

extern "C" int ta_normalize_adv_fwd_nhwc(const float* x, const uint8_t* x_u8, const int* u8_mismatch, const float* delta, float* y,
                                         const float* mean, const float* stdv, int64_t n, int c, int64_t hw, void* stream) {
    const int64_t e = static_cast<int64_t>(c) * hw;
    if (int rc = check_planes(n, e)) return rc;
    TA_REQUIRE(x && delta && y && mean && stdv, "null pointer");
    TA_REQUIRE((x_u8 == nullptr) == (u8_mismatch == nullptr), "x_u8 and its probe flag come together");
    TA_REQUIRE(c == 3 && hw % 4 == 0 && aligned16(x) && aligned16(delta) && aligned16(y),
               "the NHWC form takes three-channel images with hw %% 4 == 0 and 16-byte aligned operands (c=%d, hw=%lld)", c, (long long)hw);
    const dim3 grid(static_cast<unsigned>(ceil_div(hw / 4, kBlock)), static_cast<unsigned>(n));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (x_u8 != nullptr && (reinterpret_cast<uintptr_t>(x_u8) & 3u) == 0)
        hipLaunchKernelGGL(normalize_adv_fwd_nhwc3_kernel<true>, grid, dim3(kBlock), 0, st, x, x_u8, u8_mismatch, delta, y, mean, stdv, hw);
    else
        hipLaunchKernelGGL(normalize_adv_fwd_nhwc3_kernel<false>, grid, dim3(kBlock), 0, st, x, x_u8, u8_mismatch, delta, y, mean, stdv, hw);
    return check_launch("normalize_adv_fwd_nhwc");
}

// x_u8[i] = round(x[i] * 255) and *mismatch |= (float(x_u8[i]) / 255 != x[i]) -- the caller zeroes *mismatch first
// (ta_u8_source_probe does).  Grid-stride, 16 B in / 4 B out per lane.
__global__ __launch_bounds__(256) void u8_probe_kernel(const float* __restrict__ x, uint8_t* __restrict__ x_u8,
                                                       int* __restrict__ mismatch, int64_t numel) {
    const int64_t quads = numel >> 2;
    int bad = 0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < quads; i += static_cast<int64_t>(gridDim.x) * 256) {
        Pack<4> px;
        px.load(x + i * 4);
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float s = px[k] * 255.0f;
            const uint32_t b = (s >= 0.0f && s <= 255.0f) ? static_cast<uint32_t>(rintf(s)) : 0u;       // NaN -> 0 -> mismatch
            bad |= !(u8_to_unit(b) == px[k]);
            packed |= b << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(x_u8 + i * 4) = packed;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = quads * 4; i < numel; ++i) {
            const float s = x[i] * 255.0f;
            const uint32_t b = (s >= 0.0f && s <= 255.0f) ? static_cast<uint32_t>(rintf(s)) : 0u;
            bad |= !(u8_to_unit(b) == x[i]);
            x_u8[i] = static_cast<uint8_t>(b);
        }
    if (bad) atomicOr(mismatch, 1);
}

extern "C" int ta_u8_source_probe(const float* x, uint8_t* x_u8, int* mismatch, int64_t numel, void* stream) {
    TA_REQUIRE(x && x_u8 && mismatch && numel > 0, "null pointer or empty batch");
    TA_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (reinterpret_cast<uintptr_t>(x_u8) & 3u) == 0, "unaligned operand");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipError_t err = hipMemsetAsync(mismatch, 0, sizeof(int), st)) return static_cast<int>(err);
    const int64_t quads = numel >> 2;
    const unsigned blocks = static_cast<unsigned>(quads / 256 < 1 ? 1 : (quads / 256 > 8192 ? 8192 : quads / 256));
    hipLaunchKernelGGL(u8_probe_kernel, dim3(blocks), dim3(256), 0, st, x, x_u8, mismatch, numel);
    return check_launch("u8_source_probe");
}
