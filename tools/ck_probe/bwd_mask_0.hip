#include "bwd_instances.h"
namespace probe {
void add_bwd_mask_0(std::vector<std::unique_ptr<BwdMask>>& v) { add_bwd<ck::Tuple<NHWGC>, ck::Tuple<F32>, Mask, ConvolutionBackwardDataSpecialization::Default, BwdMask>(v); }
}
