// Glue of a convolutional surrogate between its MIOpen convolutions, fused (round 3).
//
// With eval-mode BatchNorm folded into the convolutions, a ResNet iteration on PyTorch-ROCm spends 30 % of its GPU time in
// memory-bound ATen passes around the convolutions (profiles/r03/steady_state_b125_r3a.json: per convolution a bias add
// and a ReLU clamp, per block a residual add; in the backward a threshold pass per ReLU and an add per junction -- each a
// full read + write of an activation map).  These kernels do the same arithmetic in fewer passes; every rounding point of
// the module path is kept (add bias, add identity, clamp; add, mask), so logits and gradients are the module path's bits:
//   forward   y = relu(y + b[c])                                       one pass instead of two        (in place)
//             y = relu((y + b[c]) + (o [+ bo[c]]))                     one pass instead of three / four
//   backward  out = y <= 0 ? 0 : g                                     threshold_backward              (in place allowed)
//             out = y <= 0 ? 0 : ga + gb                               junction add + threshold: one pass instead of two
// Layout-agnostic: element i of a contiguous buffer has channel (i / inner) % C -- inner = 1 for NHWC (channels_last), H*W
// for NCHW.  HBM-bound streaming kernels, 16-byte accesses, same tiling as elementwise.hip.
#include "ta_common.h"

namespace ta {

__device__ __forceinline__ float relu_like_aten(float v) { return v < 0.0f ? 0.0f : v; }   // clamp_min(v, 0): NaN stays NaN

// ---- ReLU masks as bits (round 4).  The input-gradient backward of a frozen network needs the saved activations for ONE
// thing: the sign test of threshold_backward (y <= 0 ? 0 : g) -- 4 bytes read per element to learn one bit.  The forward
// kernels below can leave that bit behind (bit i % 8 of byte i / 8 for element i of the dense buffer; set where the
// gradient passes: !(y <= 0), so a NaN activation lets it pass as ATen does), and relu_mask_kernel can take it instead of
// y: 12.1 instead of 16 B/element at a junction, 8.1 instead of 12 after a convolution.  Two neighbouring lanes hold the
// two halves of a byte (16-byte groups of four elements); the even lane fetches its neighbour's half with one DPP move.
__device__ __forceinline__ unsigned pass_bits(const float4& v) {
    return (v.x <= 0.0f ? 0u : 1u) | (v.y <= 0.0f ? 0u : 2u) | (v.z <= 0.0f ? 0u : 4u) | (v.w <= 0.0f ? 0u : 8u);
}
// the caller guarantees numel % 8 == 0, so the lanes (2k, 2k + 1) of a pair are both inside or both outside the buffer
__device__ __forceinline__ void store_pass_bits(uint8_t* __restrict__ mask, unsigned i, unsigned nibble) {
    const unsigned other = __float_as_uint(__shfl_xor(__uint_as_float(nibble), 1, kWave));
    if ((threadIdx.x & 1u) == 0u) mask[i >> 3] = static_cast<uint8_t>(nibble | (other << 4));
}
__device__ __forceinline__ unsigned load_pass_bits(const uint8_t* __restrict__ mask, unsigned i) {
    return (static_cast<unsigned>(mask[i >> 3]) >> (i & 4u)) & 15u;
}

struct ChannelOf {
    unsigned inner, channels, mask;      // mask = channels - 1 when channels is a power of two, else 0
    __device__ __forceinline__ unsigned operator()(unsigned i) const {
        const unsigned q = inner == 1u ? i : i / inner;
        return mask != 0u ? (q & mask) : q % channels;
    }
};

// the four bias values of the 16-byte group starting at element i.  mode 0: NHWC, four consecutive channels (C % 4 == 0);
// mode 1: NCHW with H*W % 4 == 0, one channel for the whole group; mode 2: NCHW with any H*W (a 7 x 7 map), per element
__device__ __forceinline__ float4 bias4(const float* __restrict__ bias, const ChannelOf& ch, unsigned i, int mode) {
    if (mode == 0) return *reinterpret_cast<const float4*>(bias + ch(i));
    if (mode == 1) {
        const float s = bias[ch(i)];
        return make_float4(s, s, s, s);
    }
    return make_float4(bias[ch(i)], bias[ch(i + 1)], bias[ch(i + 2)], bias[ch(i + 3)]);
}

template <bool RELU, bool BITS = false>
__global__ __launch_bounds__(kBlock) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias, ChannelOf ch,
                                                          unsigned numel, int mode, uint8_t* __restrict__ mask = nullptr) {
    const unsigned base = blockIdx.x * kTile;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
        const unsigned i = base + (u * kBlock + threadIdx.x) * kVec;
        if (i + kVec <= numel) {
            float4 a = *reinterpret_cast<const float4*>(y + i);
            const float4 b = bias4(bias, ch, i, mode);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            if (RELU) { a.x = relu_like_aten(a.x); a.y = relu_like_aten(a.y); a.z = relu_like_aten(a.z); a.w = relu_like_aten(a.w); }
            *reinterpret_cast<float4*>(y + i) = a;
            if (BITS) store_pass_bits(mask, i, pass_bits(a));
        } else {
            for (unsigned j = i; j < numel; ++j) {
                const float v = y[j] + bias[ch(j)];
                y[j] = RELU ? relu_like_aten(v) : v;
            }
        }
    }
}

template <bool HAS_BO, bool BITS = false>
__global__ __launch_bounds__(kBlock) void bias_add_relu_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                               const float* __restrict__ other,
                                                               const float* __restrict__ bias_other, ChannelOf ch,
                                                               unsigned numel, int mode, uint8_t* __restrict__ mask = nullptr) {
    const unsigned base = blockIdx.x * kTile;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
        const unsigned i = base + (u * kBlock + threadIdx.x) * kVec;
        if (i + kVec <= numel) {
            float4 a = *reinterpret_cast<const float4*>(y + i);
            float4 o = *reinterpret_cast<const float4*>(other + i);
            const float4 b = bias4(bias, ch, i, mode);
            if (HAS_BO) {                                                               // the shortcut's own bias first
                const float4 bo = bias4(bias_other, ch, i, mode);
                o.x += bo.x; o.y += bo.y; o.z += bo.z; o.w += bo.w;
            }
            a.x = relu_like_aten((a.x + b.x) + o.x);
            a.y = relu_like_aten((a.y + b.y) + o.y);
            a.z = relu_like_aten((a.z + b.z) + o.z);
            a.w = relu_like_aten((a.w + b.w) + o.w);
            *reinterpret_cast<float4*>(y + i) = a;
            if (BITS) store_pass_bits(mask, i, pass_bits(a));
        } else {
            for (unsigned j = i; j < numel; ++j) {
                const unsigned c = ch(j);
                const float o = HAS_BO ? other[j] + bias_other[c] : other[j];
                y[j] = relu_like_aten((y[j] + bias[c]) + o);
            }
        }
    }
}

template <bool HAS_B, bool BITS = false>   // out = y <= 0 ? 0 : ga (+ gb)   threshold_backward(grad, result, 0): NaN in y lets the gradient pass
__global__ __launch_bounds__(kBlock) void relu_mask_kernel(const float* ga, const float* gb, const float* __restrict__ y,
                                                           float* out, unsigned numel, const uint8_t* __restrict__ mask = nullptr) {
    const unsigned base = blockIdx.x * kTile;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
        const unsigned i = base + (u * kBlock + threadIdx.x) * kVec;
        if (i + kVec <= numel) {
            float4 g = *reinterpret_cast<const float4*>(ga + i);
            if (HAS_B) {
                const float4 h = *reinterpret_cast<const float4*>(gb + i);
                g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w;
            }
            if (BITS) {
                const unsigned pass = load_pass_bits(mask, i);
                g.x = (pass & 1u) ? g.x : 0.0f;
                g.y = (pass & 2u) ? g.y : 0.0f;
                g.z = (pass & 4u) ? g.z : 0.0f;
                g.w = (pass & 8u) ? g.w : 0.0f;
            } else {
                const float4 r = *reinterpret_cast<const float4*>(y + i);
                g.x = r.x <= 0.0f ? 0.0f : g.x;
                g.y = r.y <= 0.0f ? 0.0f : g.y;
                g.z = r.z <= 0.0f ? 0.0f : g.z;
                g.w = r.w <= 0.0f ? 0.0f : g.w;
            }
            *reinterpret_cast<float4*>(out + i) = g;
        } else {
            for (unsigned j = i; j < numel; ++j) {
                const float g = HAS_B ? ga[j] + gb[j] : ga[j];
                out[j] = y[j] <= 0.0f ? 0.0f : g;
            }
        }
    }
}

}  // namespace ta

using namespace ta;

static int glue_shape(int64_t numel, int channels, int64_t inner, ChannelOf* ch, int* mode) {
    TA_REQUIRE(numel > 0 && numel < (1ll << 32) - kTile && channels > 0 && inner > 0 && inner < (1ll << 31),
               "shape (numel=%lld, channels=%d, inner=%lld)", (long long)numel, channels, (long long)inner);
    TA_REQUIRE(numel % 4 == 0, "numel=%lld is not a multiple of 4", (long long)numel);
    ch->inner = static_cast<unsigned>(inner);
    ch->channels = static_cast<unsigned>(channels);
    ch->mask = (channels & (channels - 1)) == 0 ? static_cast<unsigned>(channels - 1) : 0u;
    *mode = (inner == 1 && channels % 4 == 0) ? 0 : (inner > 1 && inner % 4 == 0) ? 1 : 2;
    return 0;
}

#define TA_GLUE_GRID(numel) dim3(static_cast<unsigned>(ceil_div((numel), kTile)))

extern "C" int ta_bias_act(float* y, const float* bias, int relu, uint8_t* mask, int64_t numel, int channels, int64_t inner,
                           void* stream) {
    TA_REQUIRE(y && bias && aligned16(y) && aligned16(bias), "null or unaligned pointer");
    TA_REQUIRE(mask == nullptr || (relu && numel % 8 == 0), "pass bits need the ReLU and numel %% 8 == 0");
    ChannelOf ch;
    int vc;
    if (int rc = glue_shape(numel, channels, inner, &ch, &vc)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned n = static_cast<unsigned>(numel);
    if (mask)
        hipLaunchKernelGGL((bias_act_kernel<true, true>), TA_GLUE_GRID(numel), dim3(kBlock), 0, st, y, bias, ch, n, vc, mask);
    else if (relu)
        hipLaunchKernelGGL((bias_act_kernel<true, false>), TA_GLUE_GRID(numel), dim3(kBlock), 0, st, y, bias, ch, n, vc, mask);
    else
        hipLaunchKernelGGL((bias_act_kernel<false, false>), TA_GLUE_GRID(numel), dim3(kBlock), 0, st, y, bias, ch, n, vc, mask);
    return check_launch("bias_act");
}

extern "C" int ta_bias_add_relu(float* y, const float* bias, const float* other, const float* bias_other, uint8_t* mask,
                                int64_t numel, int channels, int64_t inner, void* stream) {
    TA_REQUIRE(y && bias && other && y != other && aligned16(y) && aligned16(bias) && aligned16(other) &&
               (bias_other == nullptr || aligned16(bias_other)), "null, aliased or unaligned pointer");
    TA_REQUIRE(mask == nullptr || numel % 8 == 0, "pass bits need numel %% 8 == 0");
    ChannelOf ch;
    int vc;
    if (int rc = glue_shape(numel, channels, inner, &ch, &vc)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned n = static_cast<unsigned>(numel);
#define TA_BAR(BO, BITS) hipLaunchKernelGGL((bias_add_relu_kernel<BO, BITS>), TA_GLUE_GRID(numel), dim3(kBlock), 0, st, y, bias, other, bias_other, ch, n, vc, mask)
    if (bias_other) { if (mask) TA_BAR(true, true); else TA_BAR(true, false); }
    else { if (mask) TA_BAR(false, true); else TA_BAR(false, false); }
#undef TA_BAR
    return check_launch("bias_add_relu");
}

// exactly one of y (the activation itself) and mask (its pass bits, from ta_bias_act / ta_bias_add_relu) is given
extern "C" int ta_relu_mask(const float* ga, const float* gb, const float* y, const uint8_t* mask, float* out, int64_t numel,
                            void* stream) {
    TA_REQUIRE(ga && out && (y != nullptr) != (mask != nullptr) && aligned16(ga) && aligned16(out) &&
               (y == nullptr || aligned16(y)) && (gb == nullptr || aligned16(gb)), "null or unaligned pointer");
    TA_REQUIRE(numel > 0 && numel % 4 == 0 && numel < (1ll << 32) - kTile, "numel=%lld", (long long)numel);
    TA_REQUIRE(mask == nullptr || numel % 8 == 0, "pass bits need numel %% 8 == 0");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned n = static_cast<unsigned>(numel);
#define TA_RM(B, BITS) hipLaunchKernelGGL((relu_mask_kernel<B, BITS>), TA_GLUE_GRID(numel), dim3(kBlock), 0, st, ga, gb, y, out, n, mask)
    if (gb) { if (mask) TA_RM(true, true); else TA_RM(true, false); }
    else { if (mask) TA_RM(false, true); else TA_RM(false, false); }
#undef TA_RM
    return check_launch("relu_mask");
}

// ---- max-pool backward + junction add + the ReLU threshold in front of the pool, one pass ---------------------------------
// The stem of the ResNets is conv -> ReLU -> max-pool(3, stride 2, padding 1).  Its backward on PyTorch-ROCm is four passes:
// the add that joins the two branches of the first block's input (pooled size), a zero fill, max_pool2d_with_indices_backward
// (channels_last: atomic adds into the filled buffer) and threshold_backward over the full-size map.  Gather form, no
// atomics, deterministic: every input pixel looks at the <= ceil(K/S)^2 windows that cover it and takes the gradient of those
// whose recorded argmax (ATen's index h * W + w within the plane) is this pixel, windows in row-major order; then the ReLU
// threshold (out = y <= 0 ? 0 : sum).  channels_last only: a lane owns four consecutive channels of one pixel.
namespace ta {

__global__ __launch_bounds__(kBlock) void maxpool_bwd_relu_kernel(const float* __restrict__ ga, const float* __restrict__ gb,
                                                                  const int64_t* __restrict__ idx, const float* __restrict__ y,
                                                                  float* __restrict__ out, int channels, int h, int w, int ph,
                                                                  int pw, int k, int s, int p, unsigned total4) {
    const unsigned q = blockIdx.x * kBlock + threadIdx.x;              // one float4 group: (n, hh, ww, c4)
    if (q >= total4) return;
    const unsigned c4n = static_cast<unsigned>(channels) / 4u;
    const unsigned c4 = q % c4n, pix = q / c4n;
    const int ww = static_cast<int>(pix % static_cast<unsigned>(w));
    const unsigned t = pix / static_cast<unsigned>(w);
    const int hh = static_cast<int>(t % static_cast<unsigned>(h));
    const unsigned n = t / static_cast<unsigned>(h);
    // windows (i, j) with i*s - p <= hh <= i*s - p + k - 1
    const int i_hi = min((hh + p) / s, ph - 1), j_hi = min((ww + p) / s, pw - 1);
    const int i_lo = max(0, (hh + p - k + s) / s), j_lo = max(0, (ww + p - k + s) / s);     // ceil((hh + p - k + 1) / s), >= 0 operands
    const int64_t here = static_cast<int64_t>(hh) * w + ww;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = i_lo; i <= i_hi; ++i)
        for (int j = j_lo; j <= j_hi; ++j) {
            const int64_t o = ((static_cast<int64_t>(n) * ph + i) * pw + j) * channels + 4 * c4;
            float4 g = *reinterpret_cast<const float4*>(ga + o);
            if (gb != nullptr) {
                const float4 g2 = *reinterpret_cast<const float4*>(gb + o);
                g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
            }
            const int64_t* ip = idx + o;
            acc.x += ip[0] == here ? g.x : 0.0f;
            acc.y += ip[1] == here ? g.y : 0.0f;
            acc.z += ip[2] == here ? g.z : 0.0f;
            acc.w += ip[3] == here ? g.w : 0.0f;
        }
    const int64_t at = ((static_cast<int64_t>(n) * h + hh) * w + ww) * channels + 4 * c4;
    const float4 r = *reinterpret_cast<const float4*>(y + at);
    acc.x = r.x <= 0.0f ? 0.0f : acc.x;
    acc.y = r.y <= 0.0f ? 0.0f : acc.y;
    acc.z = r.z <= 0.0f ? 0.0f : acc.z;
    acc.w = r.w <= 0.0f ? 0.0f : acc.w;
    *reinterpret_cast<float4*>(out + at) = acc;
}


// ---- the stem's max-pool as the backward needs it (round 6) ----
// 3 x 3 / stride 2 / padding 1 over an even-sized NHWC map (every torchvision-style ResNet stem).  ATen's max_pool2d_with_indices
// leaves an int64 argmax per pooled element (8 B where the pooled value itself is 4) and the backward then reads the stem's
// activation again for the ReLU's sign test: 200 + 401 MB of a 1.2 GB pass at batch 125.  Here the forward leaves
//   arg    one BYTE per pooled element: the tap kh * 3 + kw that won -- ATen's scan order and update rule (strictly greater, or
//          NaN), so the same element wins a tie as in max_pool2d_with_indices;
//   mask   the pass bits of the ACTIVATION it pools (layout of the kernels above): a window (i, j) owns the pixels
//          (2i + {0, 1}, 2j + {0, 1}) -- taps kh, kw >= 1 -- and a lane handles eight channels, one whole byte per owned pixel.
// and the backward reads 1 + 1/8 byte where it read 8 + 4; the activation is not kept for the backward at all.
__global__ __launch_bounds__(kBlock) void maxpool3s2_fwd_kernel(const float* __restrict__ y, float* __restrict__ pooled,
                                                                uint8_t* __restrict__ arg, uint8_t* __restrict__ mask, int channels,
                                                                int h, int w, int ph, int pw, unsigned total8) {
    const unsigned q = blockIdx.x * kBlock + threadIdx.x;              // (n, i, j, c8): eight channels of one window
    if (q >= total8) return;
    const unsigned c8n = static_cast<unsigned>(channels) / 8u;
    const unsigned c8 = q % c8n, win = q / c8n;
    const int j = static_cast<int>(win % static_cast<unsigned>(pw));
    const unsigned t = win / static_cast<unsigned>(pw);
    const int i = static_cast<int>(t % static_cast<unsigned>(ph));
    const unsigned n = t / static_cast<unsigned>(ph);
    const unsigned first = (i == 0 ? 3u : 0u) + (j == 0 ? 1u : 0u);   // ATen starts the argmax at the window's first valid tap
    float4 lo = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), hi = lo;
    unsigned alo = first * 0x01010101u, ahi = alo;                     // four argmax bytes each
#define TA_POOL_TAKE(best, v, word, shift, code) \
    if ((v) > (best) || (v) != (v)) { (best) = (v); (word) = ((word) & ~(0xffu << (shift))) | ((code) << (shift)); }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = 2 * i - 1 + kh;
        if (ih < 0) continue;                                          // ih <= h - 1 always: h is even, ph = h / 2
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int iw = 2 * j - 1 + kw;
            if (iw < 0) continue;
            const int64_t o = ((static_cast<int64_t>(n) * h + ih) * w + iw) * channels + 8 * c8;
            const float4 a = *reinterpret_cast<const float4*>(y + o), b = *reinterpret_cast<const float4*>(y + o + 4);
            const unsigned code = static_cast<unsigned>(kh * 3 + kw);
            TA_POOL_TAKE(lo.x, a.x, alo, 0, code) TA_POOL_TAKE(lo.y, a.y, alo, 8, code)
            TA_POOL_TAKE(lo.z, a.z, alo, 16, code) TA_POOL_TAKE(lo.w, a.w, alo, 24, code)
            TA_POOL_TAKE(hi.x, b.x, ahi, 0, code) TA_POOL_TAKE(hi.y, b.y, ahi, 8, code)
            TA_POOL_TAKE(hi.z, b.z, ahi, 16, code) TA_POOL_TAKE(hi.w, b.w, ahi, 24, code)
            if (kh >= 1 && kw >= 1) mask[o >> 3] = static_cast<uint8_t>(pass_bits(a) | (pass_bits(b) << 4));
        }
    }
#undef TA_POOL_TAKE
    const int64_t po = ((static_cast<int64_t>(n) * ph + i) * pw + j) * channels + 8 * c8;
    *reinterpret_cast<float4*>(pooled + po) = lo;
    *reinterpret_cast<float4*>(pooled + po + 4) = hi;
    *reinterpret_cast<uint2*>(arg + po) = make_uint2(alo, ahi);
}

// The backward on what the kernel above left.  maxpool_bwd_relu_kernel gathers per input pixel: the <= 4 windows over it are
// loaded by each of the pixels they cover (2.25 loads of every pooled element, 81 B of cache traffic per 16 B written -- the
// kernel ran at 2.4 TB/s of HBM traffic, bound by L2 -> L1).  Here a lane owns the 2 x 2 pixels (2i + {0,1}, 2j + {0,1}) x four
// channels: the windows over them are (i, j), (i, j+1), (i+1, j), (i+1, j+1) -- four loads serve four pixels.  Same windows
// in the same order and the same sums per pixel as the gather (a window that does not exist adds 0.0f to a sum that is never -0).
__device__ __forceinline__ void pool_take(float4& acc, const float4& g, unsigned won, unsigned here) {
    acc.x += (won & 0xffu) == here ? g.x : 0.0f;
    acc.y += ((won >> 8) & 0xffu) == here ? g.y : 0.0f;
    acc.z += ((won >> 16) & 0xffu) == here ? g.z : 0.0f;
    acc.w += (won >> 24) == here ? g.w : 0.0f;
}

__global__ __launch_bounds__(kBlock) void maxpool3s2_bwd_relu_kernel(const float* __restrict__ ga, const float* __restrict__ gb,
                                                                     const uint8_t* __restrict__ arg, const uint8_t* __restrict__ mask,
                                                                     float* __restrict__ out, int channels, int h, int w, int ph,
                                                                     int pw, unsigned total4) {
    const unsigned q = blockIdx.x * kBlock + threadIdx.x;              // (n, i, j, c4)
    if (q >= total4) return;
    const unsigned c4n = static_cast<unsigned>(channels) / 4u;
    const unsigned c4 = q % c4n, win = q / c4n;
    const int j = static_cast<int>(win % static_cast<unsigned>(pw));
    const unsigned t = win / static_cast<unsigned>(pw);
    const int i = static_cast<int>(t % static_cast<unsigned>(ph));
    const unsigned n = t / static_cast<unsigned>(ph);
    const bool right = j + 1 < pw, down = i + 1 < ph;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 g[4];
    unsigned won[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {                                      // v = 2 * di + dj: window (i + di, j + dj)
        const bool there = ((v & 1) == 0 || right) && ((v & 2) == 0 || down);
        g[v] = zero;
        won[v] = 0xffffffffu;                                          // no tap has this code
        if (there) {
            const int64_t o = ((static_cast<int64_t>(n) * ph + i + (v >> 1)) * pw + j + (v & 1)) * channels + 4 * c4;
            g[v] = *reinterpret_cast<const float4*>(ga + o);
            if (gb != nullptr) {
                const float4 g2 = *reinterpret_cast<const float4*>(gb + o);
                g[v].x += g2.x; g[v].y += g2.y; g[v].z += g2.z; g[v].w += g2.w;
            }
            won[v] = *reinterpret_cast<const unsigned*>(arg + o);
        }
    }
    // the tap of window v that pixel (2i + a, 2j + b) is: rows 2(i + di) - 1 + kh, columns alike
    float4 px[4] = {zero, zero, zero, zero};                            // pixel 2 * a + b
    pool_take(px[0], g[0], won[0], 4u);
    pool_take(px[1], g[0], won[0], 5u); pool_take(px[1], g[1], won[1], 3u);
    pool_take(px[2], g[0], won[0], 7u); pool_take(px[2], g[2], won[2], 1u);
    pool_take(px[3], g[0], won[0], 8u); pool_take(px[3], g[1], won[1], 6u); pool_take(px[3], g[2], won[2], 2u); pool_take(px[3], g[3], won[3], 0u);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int64_t at = ((static_cast<int64_t>(n) * h + 2 * i + (v >> 1)) * w + 2 * j + (v & 1)) * channels + 4 * c4;
        const unsigned pass = load_pass_bits(mask, static_cast<unsigned>(at));
        float4 r = px[v];
        r.x = (pass & 1u) ? r.x : 0.0f;
        r.y = (pass & 2u) ? r.y : 0.0f;
        r.z = (pass & 4u) ? r.z : 0.0f;
        r.w = (pass & 8u) ? r.w : 0.0f;
        *reinterpret_cast<float4*>(out + at) = r;
    }
}

}  // namespace ta

extern "C" int ta_maxpool_bwd_relu(const float* ga, const float* gb, const int64_t* idx, const float* y, float* out, int64_t n,
                                   int channels, int h, int w, int ph, int pw, int k, int s, int p, void* stream) {
    TA_REQUIRE(ga && idx && y && out && aligned16(ga) && aligned16(y) && aligned16(out) && aligned16(idx) &&
               (gb == nullptr || aligned16(gb)), "null or unaligned pointer");
    TA_REQUIRE(n > 0 && channels > 0 && channels % 4 == 0 && h > 0 && w > 0 && ph > 0 && pw > 0 && k > 0 && s > 0 && p >= 0 && p < k,
               "shape (n=%lld, c=%d, %dx%d <- %dx%d, k=%d s=%d p=%d)", (long long)n, channels, h, w, ph, pw, k, s, p);
    TA_REQUIRE((ph - 1) * s - p < h && (pw - 1) * s - p < w, "pooled size does not belong to this input size");
    const int64_t total4 = n * h * w * (channels / 4);
    TA_REQUIRE(total4 < (1ll << 32) - kBlock, "too many elements for one launch");
    hipLaunchKernelGGL(maxpool_bwd_relu_kernel, dim3(static_cast<unsigned>(ceil_div(total4, kBlock))), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), ga, gb, idx, y, out, channels, h, w, ph, pw, k, s, p,
                       static_cast<unsigned>(total4));
    return check_launch("maxpool_bwd_relu");
}

extern "C" int ta_maxpool3s2_fwd(const float* y, float* pooled, uint8_t* arg, uint8_t* mask, int64_t n, int channels, int h, int w,
                                 void* stream) {
    TA_REQUIRE(y && pooled && arg && mask && aligned16(y) && aligned16(pooled) && (reinterpret_cast<uintptr_t>(arg) & 7u) == 0,
               "null or unaligned pointer");
    TA_REQUIRE(n > 0 && channels > 0 && channels % 8 == 0 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0,
               "shape (n=%lld, c=%d, %dx%d): channels %% 8 == 0 and an even map", (long long)n, channels, h, w);
    const int64_t total8 = n * (h / 2) * (w / 2) * (channels / 8);
    TA_REQUIRE(n * h * w * channels < (1ll << 32) - 4 * kBlock, "too many elements for one launch");
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(static_cast<unsigned>(ceil_div(total8, kBlock))), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), y, pooled, arg, mask, channels, h, w, h / 2, w / 2,
                       static_cast<unsigned>(total8));
    return check_launch("maxpool3s2_fwd");
}

extern "C" int ta_maxpool3s2_bwd_relu(const float* ga, const float* gb, const uint8_t* arg, const uint8_t* mask, float* out, int64_t n,
                                      int channels, int h, int w, void* stream) {
    TA_REQUIRE(ga && arg && mask && out && aligned16(ga) && aligned16(out) && (reinterpret_cast<uintptr_t>(arg) & 3u) == 0 &&
               (gb == nullptr || aligned16(gb)), "null or unaligned pointer");
    TA_REQUIRE(n > 0 && channels > 0 && channels % 8 == 0 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0,
               "shape (n=%lld, c=%d, %dx%d): channels %% 8 == 0 and an even map", (long long)n, channels, h, w);
    const int64_t total4 = n * (h / 2) * (w / 2) * (channels / 4);
    TA_REQUIRE(n * h * w * channels < (1ll << 32) - 4 * kBlock, "too many elements for one launch");
    hipLaunchKernelGGL(maxpool3s2_bwd_relu_kernel, dim3(static_cast<unsigned>(ceil_div(total4, kBlock))), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), ga, gb, arg, mask, out, channels, h, w, h / 2, w / 2,
                       static_cast<unsigned>(total4));
    return check_launch("maxpool3s2_bwd_relu");
}
