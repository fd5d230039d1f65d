#include "bwd_instances.h"
namespace probe {
void add_bwd_add_mask_0(std::vector<std::unique_ptr<BwdAddMask>>& v) { add_bwd<ck::Tuple<NHWGC, NHWGC>, ck::Tuple<F32, F32>, AddMask, ConvolutionBackwardDataSpecialization::Default, BwdAddMask>(v); }
}
