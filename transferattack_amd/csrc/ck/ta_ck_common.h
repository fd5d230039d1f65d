// libta_ck.so (include/ta_ck.h): composable_kernel convolutions with the surrogate's glue pass as CDE operation.
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
#include "ck/library/tensor_operation_instance/add_device_operation_instance.hpp"
#include "../../../include/ta_ck.h"

namespace ta_ck {
using F32 = float;
using PassThrough = ck::tensor_operation::element_wise::PassThrough;
using namespace ck::tensor_layout::convolution;

// The epilogues keep the rounding order and the NaN behaviour of csrc/glue.hip (torch's clamp_min_ / threshold_backward):
// a NaN sum stays NaN, a NaN activation lets the gradient pass.
struct BiasRelu {                       // glue.hip bias_act:      clamp_min(acc + b, 0)
    template <typename Y, typename X0, typename X1>
    __host__ __device__ constexpr void operator()(Y& y, const X0& acc, const X1& b) const {
        const float a = acc + b;
        y = a < 0.0f ? 0.0f : a;
    }
};
struct BiasAddRelu {                    // glue.hip bias_add_relu: clamp_min((acc + b) + o, 0)
    template <typename Y, typename X0, typename X1, typename X2>
    __host__ __device__ constexpr void operator()(Y& y, const X0& acc, const X1& b, const X2& o) const {
        const float a = (acc + b) + o;
        y = a < 0.0f ? 0.0f : a;
    }
};
struct BiasAddBiasRelu {                // ... with a projection shortcut: clamp_min((acc + b) + (o + bo), 0)
    template <typename Y, typename X0, typename X1, typename X2, typename X3>
    __host__ __device__ constexpr void operator()(Y& y, const X0& acc, const X1& b, const X2& o, const X3& bo) const {
        const float a = (acc + b) + (o + bo);
        y = a < 0.0f ? 0.0f : a;
    }
};
struct Mask {                           // glue.hip relu_mask:     threshold_backward(acc, act, 0)
    template <typename Y, typename X0, typename X1>
    __host__ __device__ constexpr void operator()(Y& y, const X0& acc, const X1& act) const {
        y = act <= 0.0f ? 0.0f : static_cast<float>(acc);
    }
};
struct AddMask {                        // ... with the junction:  threshold_backward(acc + other, act, 0)
    template <typename Y, typename X0, typename X1, typename X2>
    __host__ __device__ constexpr void operator()(Y& y, const X0& acc, const X1& other, const X2& act) const {
        const float a = acc + other;
        y = act <= 0.0f ? 0.0f : a;
    }
};

template <typename DsLayout, typename DsData, typename Op>
using FwdBase = ck::tensor_operation::device::DeviceGroupedConvFwdMultipleABD<2, NHWGC, GKYXC, DsLayout, NHWGK, F32, F32, DsData, F32,
                                                                            PassThrough, PassThrough, Op>;
using FwdBias = FwdBase<ck::Tuple<G_K>, ck::Tuple<F32>, BiasRelu>;
using FwdBiasAdd = FwdBase<ck::Tuple<G_K, NHWGK>, ck::Tuple<F32, F32>, BiasAddRelu>;
using FwdBiasAddBias = FwdBase<ck::Tuple<G_K, NHWGK, G_K>, ck::Tuple<F32, F32, F32>, BiasAddBiasRelu>;
// a stride-1 convolution's input gradient as a FORWARD convolution of the output gradient with the flipped, transposed filter:
// the forward kernels with the backward glue as epilogue (their D operands have the layout of this "forward" output)
using FwdMask = FwdBase<ck::Tuple<NHWGK>, ck::Tuple<F32>, Mask>;
using FwdAddMask = FwdBase<ck::Tuple<NHWGK, NHWGK>, ck::Tuple<F32, F32>, AddMask>;
// one translation unit each (the instantiation of ~12 kernels takes a minute):
void add_fwd_bias_any(std::vector<std::unique_ptr<FwdBias>>& v);              // any filter
void add_fwd_bias_1x1(std::vector<std::unique_ptr<FwdBias>>& v);              // 1x1 / stride 1 / no padding
void add_fwd_bias_add_1x1(std::vector<std::unique_ptr<FwdBiasAdd>>& v);
void add_fwd_bias_add_bias_1x1(std::vector<std::unique_ptr<FwdBiasAddBias>>& v);
void add_fwd_mask_any(std::vector<std::unique_ptr<FwdMask>>& v);
void add_fwd_mask_1x1(std::vector<std::unique_ptr<FwdMask>>& v);
void add_fwd_add_mask_any(std::vector<std::unique_ptr<FwdAddMask>>& v);
void add_fwd_add_mask_1x1(std::vector<std::unique_ptr<FwdAddMask>>& v);
}  // namespace ta_ck
