// Probe (round 6, review item 4): convolution + epilogue instances of composable_kernel against MIOpen's convolution + this
// repository's glue kernels.  Built by tools/ck_probe/build.sh into tools/bin/libck_probe.so; driven by tools/ck_conv_probe.py.
// Not part of libta_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <array>
#include <memory>
#include <string>
#include <tuple>
#include <vector>
#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"
#include "ck/tensor_operation/gpu/device/device_grouped_conv_fwd_multiple_abd.hpp"
#include "ck/tensor_operation/gpu/device/device_grouped_conv_bwd_data_multiple_d.hpp"
#include "ck/library/tensor_operation_instance/add_device_operation_instance.hpp"

namespace probe {
using F32 = float;
using PassThrough = ck::tensor_operation::element_wise::PassThrough;
using namespace ck::tensor_layout::convolution;

// y = clamp_min(acc + bias[k], 0): csrc/glue.hip bias_act (NaN passes, as torch's clamp_min_)
struct BiasRelu {
    template <typename Y, typename X0, typename X1>
    __host__ __device__ constexpr void operator()(Y& y, const X0& acc, const X1& b) const {
        const float a = acc + b;
        y = a < 0.0f ? 0.0f : a;
    }
};
// y = clamp_min((acc + bias[k]) + shortcut, 0): csrc/glue.hip bias_add_relu, same rounding order
struct BiasAddRelu {
    template <typename Y, typename X0, typename X1, typename X2>
    __host__ __device__ constexpr void operator()(Y& y, const X0& acc, const X1& b, const X2& o) const {
        const float a = (acc + b) + o;
        y = a < 0.0f ? 0.0f : a;
    }
};
// dx = threshold_backward(acc, act, 0): the gradient passes where !(act <= 0)
struct Mask {
    template <typename Y, typename X0, typename X1>
    __host__ __device__ constexpr void operator()(Y& y, const X0& acc, const X1& act) const {
        y = act <= 0.0f ? 0.0f : static_cast<float>(acc);
    }
};
// dx = threshold_backward(acc + other, act, 0): the junction of a residual block
struct AddMask {
    template <typename Y, typename X0, typename X1, typename X2>
    __host__ __device__ constexpr void operator()(Y& y, const X0& acc, const X1& other, const X2& act) const {
        const float a = acc + other;
        y = act <= 0.0f ? 0.0f : a;
    }
};

template <typename DsLayout, typename DsData, typename Op>
using FwdBase = ck::tensor_operation::device::DeviceGroupedConvFwdMultipleABD<2, NHWGC, GKYXC, DsLayout, NHWGK, F32, F32, DsData, F32,
                                                                            PassThrough, PassThrough, Op>;
using FwdPlain = FwdBase<ck::Tuple<>, ck::Tuple<>, PassThrough>;
using FwdBias = FwdBase<ck::Tuple<G_K>, ck::Tuple<F32>, BiasRelu>;
using FwdBiasAdd = FwdBase<ck::Tuple<G_K, NHWGK>, ck::Tuple<F32, F32>, BiasAddRelu>;

// backward data: A = output gradient (NHWGK), B = weight (GKYXC), E = input gradient (NHWGC)
template <typename DsLayout, typename DsData, typename Op>
using BwdBase = ck::tensor_operation::device::DeviceGroupedConvBwdDataMultipleD<2, NHWGK, GKYXC, DsLayout, NHWGC, F32, F32, DsData, F32,
                                                                              PassThrough, PassThrough, Op>;
using BwdPlain = BwdBase<ck::Tuple<>, ck::Tuple<>, PassThrough>;
using BwdMask = BwdBase<ck::Tuple<NHWGC>, ck::Tuple<F32>, Mask>;
using BwdAddMask = BwdBase<ck::Tuple<NHWGC, NHWGC>, ck::Tuple<F32, F32>, AddMask>;

// filled by the instance translation units: _0 = any filter (Default specialization), _1 = 1x1 / stride 1 / no padding
void add_fwd_plain_0(std::vector<std::unique_ptr<FwdPlain>>& v);
void add_fwd_plain_1(std::vector<std::unique_ptr<FwdPlain>>& v);
void add_fwd_bias_0(std::vector<std::unique_ptr<FwdBias>>& v);
void add_fwd_bias_1(std::vector<std::unique_ptr<FwdBias>>& v);
void add_fwd_bias_add_1(std::vector<std::unique_ptr<FwdBiasAdd>>& v);
void add_bwd_plain_0(std::vector<std::unique_ptr<BwdPlain>>& v);
void add_bwd_plain_1(std::vector<std::unique_ptr<BwdPlain>>& v);
void add_bwd_mask_0(std::vector<std::unique_ptr<BwdMask>>& v);
void add_bwd_mask_1(std::vector<std::unique_ptr<BwdMask>>& v);
void add_bwd_add_mask_0(std::vector<std::unique_ptr<BwdAddMask>>& v);
void add_bwd_add_mask_1(std::vector<std::unique_ptr<BwdAddMask>>& v);
}  // namespace probe
