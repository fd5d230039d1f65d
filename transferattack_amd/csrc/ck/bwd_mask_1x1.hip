#include "bwd_instances.h"
namespace ta_ck {
void add_bwd_mask_1x1(std::vector<std::unique_ptr<BwdMask>>& v) { add_bwd<ck::Tuple<NHWGC>, ck::Tuple<F32>, Mask, ConvolutionBackwardDataSpecialization::Filter1x1Stride1Pad0, BwdMask>(v); }
}
