#include "fwd_instances.h"
namespace probe {
void add_fwd_bias_add_1(std::vector<std::unique_ptr<FwdBiasAdd>>& v) { add_fwd<ck::Tuple<G_K, NHWGK>, ck::Tuple<F32, F32>, BiasAddRelu, ConvFwd1x1S1P0, FwdBiasAdd>(v); }
}
