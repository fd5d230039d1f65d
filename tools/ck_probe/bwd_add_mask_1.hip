#include "bwd_instances.h"
namespace probe {
void add_bwd_add_mask_1(std::vector<std::unique_ptr<BwdAddMask>>& v) { add_bwd<ck::Tuple<NHWGC, NHWGC>, ck::Tuple<F32, F32>, AddMask, ConvolutionBackwardDataSpecialization::Filter1x1Stride1Pad0, BwdAddMask>(v); }
}
