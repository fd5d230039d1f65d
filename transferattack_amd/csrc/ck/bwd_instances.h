#include "ta_ck_common.h"
#include "ck/tensor_operation/gpu/device/convolution_backward_data_specialization.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_grouped_conv_bwd_data_multiple_d_xdl_cshuffle_v1.hpp"
#include "_gen/ta_ck_lists.inc"

namespace ta_ck {
using namespace ck::tensor_operation::device;
template <ck::index_t... Is> using S = ck::Sequence<Is...>;
template <ck::index_t NDimSpatial, typename ALayout, typename BLayout, typename DsLayout, typename ELayout,
          ConvolutionBackwardDataSpecialization ConvSpec, typename DsData, typename Op>
using bwd_list = std::tuple<TA_CK_BWD_ROWS(DsData, Op)>;

template <typename DsLayout, typename DsData, typename Op, ConvolutionBackwardDataSpecialization Spec, typename Base>
void add_bwd(std::vector<std::unique_ptr<Base>>& v) {
    ck::tensor_operation::device::instance::add_device_operation_instances(v, bwd_list<2, NHWGK, GKYXC, DsLayout, NHWGC, Spec, DsData, Op>{});
}
}  // namespace ta_ck
