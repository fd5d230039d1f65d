// one translation unit per (epilogue, convolution specialization): -DPROBE_KIND=..., -DPROBE_1X1=0|1
#include "ck_probe_common.h"
#include "ck/tensor_operation/gpu/device/impl/device_grouped_conv_fwd_multiple_abd_xdl_cshuffle.hpp"
#include "ck/library/tensor_operation_instance/gpu/grouped_conv_fwd/device_grouped_conv_fwd_xdl_instance.hpp"

namespace probe {
using namespace ck::tensor_operation::device::instance;
template <typename DsLayout, typename DsData, typename Op, ck::tensor_operation::device::ConvolutionForwardSpecialization Spec, typename Base>
void add_fwd(std::vector<std::unique_ptr<Base>>& v) {
    add_device_operation_instances(v, device_grouped_conv_fwd_xdl_f32_instances<2, NHWGC, GKYXC, DsLayout, NHWGK, Spec, DsData, Op>{});
    add_device_operation_instances(v, device_grouped_conv_fwd_xdl_f32_16x16_instances<2, NHWGC, GKYXC, DsLayout, NHWGK, Spec, DsData, Op>{});
}
}  // namespace probe
