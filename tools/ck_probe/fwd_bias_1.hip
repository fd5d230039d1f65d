#include "fwd_instances.h"
namespace probe {
void add_fwd_bias_1(std::vector<std::unique_ptr<FwdBias>>& v) { add_fwd<ck::Tuple<G_K>, ck::Tuple<F32>, BiasRelu, ConvFwd1x1S1P0, FwdBias>(v); }
}
