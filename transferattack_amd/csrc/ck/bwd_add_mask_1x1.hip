#include "bwd_instances.h"
namespace ta_ck {
void add_bwd_add_mask_1x1(std::vector<std::unique_ptr<BwdAddMask>>& v) { add_bwd<ck::Tuple<NHWGC, NHWGC>, ck::Tuple<F32, F32>, AddMask, ConvolutionBackwardDataSpecialization::Filter1x1Stride1Pad0, BwdAddMask>(v); }
}
