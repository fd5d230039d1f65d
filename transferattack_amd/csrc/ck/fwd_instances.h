#include "ta_ck_common.h"
#include "ck/tensor_operation/gpu/device/convolution_forward_specialization.hpp"
#include "ck/tensor_operation/gpu/device/gemm_specialization.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_grouped_conv_fwd_multiple_abd_xdl_cshuffle.hpp"
#include "_gen/ta_ck_lists.inc"

namespace ta_ck {
using namespace ck::tensor_operation::device;
template <ck::index_t... Is> using S = ck::Sequence<Is...>;
static constexpr auto GemmMNKPadding = GemmSpecialization::MNKPadding;
template <ck::index_t NDimSpatial, typename ALayout, typename BLayout, typename DsLayout, typename ELayout,
          ConvolutionForwardSpecialization ConvSpec, typename DsDataTypes, typename OutElementOp>
using fwd_list = std::tuple<TA_CK_FWD_ROWS>;

template <typename DsLayout, typename DsData, typename Op, ConvolutionForwardSpecialization Spec, typename Base>
void add_fwd(std::vector<std::unique_ptr<Base>>& v) {
    ck::tensor_operation::device::instance::add_device_operation_instances(v, fwd_list<2, NHWGC, GKYXC, DsLayout, NHWGK, Spec, DsData, Op>{});
}
}  // namespace ta_ck
