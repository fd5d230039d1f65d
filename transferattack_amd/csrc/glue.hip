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

struct ChannelOf {
    unsigned inner, channels, mask;      // mask = channels - 1 when channels is a power of two, else 0
    __device__ __forceinline__ unsigned operator()(unsigned i) const {
        const unsigned q = inner == 1u ? i : i / inner;
        return mask != 0u ? (q & mask) : q % channels;
    }
};

template <bool RELU>
__global__ __launch_bounds__(kBlock) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias, ChannelOf ch,
                                                          unsigned numel, bool vec_channels) {
    const unsigned base = blockIdx.x * kTile;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
        const unsigned i = base + (u * kBlock + threadIdx.x) * kVec;
        if (i + kVec <= numel) {
            float4 a = *reinterpret_cast<const float4*>(y + i);
            float4 b;
            if (vec_channels) {                       // NHWC: four consecutive channels (C % 4 == 0)
                b = *reinterpret_cast<const float4*>(bias + ch(i));
            } else {                                  // NCHW: one channel for the whole group (inner % 4 == 0)
                const float s = bias[ch(i)];
                b = make_float4(s, s, s, s);
            }
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            if (RELU) { a.x = relu_like_aten(a.x); a.y = relu_like_aten(a.y); a.z = relu_like_aten(a.z); a.w = relu_like_aten(a.w); }
            *reinterpret_cast<float4*>(y + i) = a;
        } else {
            for (unsigned j = i; j < numel; ++j) {
                const float v = y[j] + bias[ch(j)];
                y[j] = RELU ? relu_like_aten(v) : v;
            }
        }
    }
}

template <bool HAS_BO>
__global__ __launch_bounds__(kBlock) void bias_add_relu_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                               const float* __restrict__ other,
                                                               const float* __restrict__ bias_other, ChannelOf ch,
                                                               unsigned numel, bool vec_channels) {
    const unsigned base = blockIdx.x * kTile;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
        const unsigned i = base + (u * kBlock + threadIdx.x) * kVec;
        if (i + kVec <= numel) {
            float4 a = *reinterpret_cast<const float4*>(y + i);
            float4 o = *reinterpret_cast<const float4*>(other + i);
            float4 b, bo = make_float4(0.f, 0.f, 0.f, 0.f);
            const unsigned c = ch(i);
            if (vec_channels) {
                b = *reinterpret_cast<const float4*>(bias + c);
                if (HAS_BO) bo = *reinterpret_cast<const float4*>(bias_other + c);
            } else {
                const float s = bias[c];
                b = make_float4(s, s, s, s);
                if (HAS_BO) { const float t = bias_other[c]; bo = make_float4(t, t, t, t); }
            }
            if (HAS_BO) { o.x += bo.x; o.y += bo.y; o.z += bo.z; o.w += bo.w; }        // the shortcut's own bias first
            a.x = relu_like_aten((a.x + b.x) + o.x);
            a.y = relu_like_aten((a.y + b.y) + o.y);
            a.z = relu_like_aten((a.z + b.z) + o.z);
            a.w = relu_like_aten((a.w + b.w) + o.w);
            *reinterpret_cast<float4*>(y + i) = a;
        } else {
            for (unsigned j = i; j < numel; ++j) {
                const unsigned c = ch(j);
                const float o = HAS_BO ? other[j] + bias_other[c] : other[j];
                y[j] = relu_like_aten((y[j] + bias[c]) + o);
            }
        }
    }
}

template <bool HAS_B>      // out = y <= 0 ? 0 : ga (+ gb)      threshold_backward(grad, result, 0): NaN in y lets the gradient pass
__global__ __launch_bounds__(kBlock) void relu_mask_kernel(const float* ga, const float* gb, const float* __restrict__ y,
                                                           float* out, unsigned numel) {
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
            const float4 r = *reinterpret_cast<const float4*>(y + i);
            g.x = r.x <= 0.0f ? 0.0f : g.x;
            g.y = r.y <= 0.0f ? 0.0f : g.y;
            g.z = r.z <= 0.0f ? 0.0f : g.z;
            g.w = r.w <= 0.0f ? 0.0f : g.w;
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

static int glue_shape(int64_t numel, int channels, int64_t inner, ChannelOf* ch, bool* vec_channels) {
    TA_REQUIRE(numel > 0 && numel < (1ll << 32) - kTile && channels > 0 && inner > 0 && inner < (1ll << 31),
               "shape (numel=%lld, channels=%d, inner=%lld)", (long long)numel, channels, (long long)inner);
    TA_REQUIRE(numel % 4 == 0 && ((inner == 1 && channels % 4 == 0) || (inner > 1 && inner % 4 == 0)),
               "16-byte groups must not straddle channels (numel=%lld, channels=%d, inner=%lld)", (long long)numel, channels,
               (long long)inner);
    ch->inner = static_cast<unsigned>(inner);
    ch->channels = static_cast<unsigned>(channels);
    ch->mask = (channels & (channels - 1)) == 0 ? static_cast<unsigned>(channels - 1) : 0u;
    *vec_channels = inner == 1;
    return 0;
}

#define TA_GLUE_GRID(numel) dim3(static_cast<unsigned>(ceil_div((numel), kTile)))

extern "C" int ta_bias_act(float* y, const float* bias, int relu, int64_t numel, int channels, int64_t inner, void* stream) {
    TA_REQUIRE(y && bias && aligned16(y) && aligned16(bias), "null or unaligned pointer");
    ChannelOf ch;
    bool vc;
    if (int rc = glue_shape(numel, channels, inner, &ch, &vc)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (relu)
        hipLaunchKernelGGL(bias_act_kernel<true>, TA_GLUE_GRID(numel), dim3(kBlock), 0, st, y, bias, ch, static_cast<unsigned>(numel), vc);
    else
        hipLaunchKernelGGL(bias_act_kernel<false>, TA_GLUE_GRID(numel), dim3(kBlock), 0, st, y, bias, ch, static_cast<unsigned>(numel), vc);
    return check_launch("bias_act");
}

extern "C" int ta_bias_add_relu(float* y, const float* bias, const float* other, const float* bias_other, int64_t numel,
                                int channels, int64_t inner, void* stream) {
    TA_REQUIRE(y && bias && other && y != other && aligned16(y) && aligned16(bias) && aligned16(other) &&
               (bias_other == nullptr || aligned16(bias_other)), "null, aliased or unaligned pointer");
    ChannelOf ch;
    bool vc;
    if (int rc = glue_shape(numel, channels, inner, &ch, &vc)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bias_other)
        hipLaunchKernelGGL(bias_add_relu_kernel<true>, TA_GLUE_GRID(numel), dim3(kBlock), 0, st, y, bias, other, bias_other, ch,
                           static_cast<unsigned>(numel), vc);
    else
        hipLaunchKernelGGL(bias_add_relu_kernel<false>, TA_GLUE_GRID(numel), dim3(kBlock), 0, st, y, bias, other, bias_other, ch,
                           static_cast<unsigned>(numel), vc);
    return check_launch("bias_add_relu");
}

extern "C" int ta_relu_mask(const float* ga, const float* gb, const float* y, float* out, int64_t numel, void* stream) {
    TA_REQUIRE(ga && y && out && aligned16(ga) && aligned16(y) && aligned16(out) && (gb == nullptr || aligned16(gb)),
               "null or unaligned pointer");
    TA_REQUIRE(numel > 0 && numel % 4 == 0 && numel < (1ll << 32) - kTile, "numel=%lld", (long long)numel);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (gb)
        hipLaunchKernelGGL(relu_mask_kernel<true>, TA_GLUE_GRID(numel), dim3(kBlock), 0, st, ga, gb, y, out, static_cast<unsigned>(numel));
    else
        hipLaunchKernelGGL(relu_mask_kernel<false>, TA_GLUE_GRID(numel), dim3(kBlock), 0, st, ga, gb, y, out, static_cast<unsigned>(numel));
    return check_launch("relu_mask");
}
