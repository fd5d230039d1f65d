#include "fwd_instances.h"
namespace ta_ck {
void add_fwd_mask_1x1(std::vector<std::unique_ptr<FwdMask>>& v) { add_fwd<ck::Tuple<NHWGK>, ck::Tuple<F32>, Mask, ConvolutionForwardSpecialization::Filter1x1Stride1Pad0, FwdMask>(v); }
}
