#include "fwd_instances.h"
namespace ta_ck {
void add_fwd_bias_any(std::vector<std::unique_ptr<FwdBias>>& v) { add_fwd<ck::Tuple<G_K>, ck::Tuple<F32>, BiasRelu, ConvolutionForwardSpecialization::Default, FwdBias>(v); }
}
