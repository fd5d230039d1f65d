#include "fwd_instances.h"
namespace ta_ck {
void add_fwd_bias_add_1x1(std::vector<std::unique_ptr<FwdBiasAdd>>& v) { add_fwd<ck::Tuple<G_K, NHWGK>, ck::Tuple<F32, F32>, BiasAddRelu, ConvolutionForwardSpecialization::Filter1x1Stride1Pad0, FwdBiasAdd>(v); }
}
