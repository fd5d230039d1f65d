#include "fwd_instances.h"
namespace ta_ck {
void add_fwd_bias_add_bias_1x1(std::vector<std::unique_ptr<FwdBiasAddBias>>& v) { add_fwd<ck::Tuple<G_K, NHWGK, G_K>, ck::Tuple<F32, F32, F32>, BiasAddBiasRelu, ConvolutionForwardSpecialization::Filter1x1Stride1Pad0, FwdBiasAddBias>(v); }
}
