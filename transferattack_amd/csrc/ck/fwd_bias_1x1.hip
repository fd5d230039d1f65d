#include "fwd_instances.h"
namespace ta_ck {
void add_fwd_bias_1x1(std::vector<std::unique_ptr<FwdBias>>& v) { add_fwd<ck::Tuple<G_K>, ck::Tuple<F32>, BiasRelu, ConvolutionForwardSpecialization::Filter1x1Stride1Pad0, FwdBias>(v); }
}
