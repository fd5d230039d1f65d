#include "fwd_instances.h"
namespace ta_ck {
void add_fwd_add_mask_1x1(std::vector<std::unique_ptr<FwdAddMask>>& v) { add_fwd<ck::Tuple<NHWGK, NHWGK>, ck::Tuple<F32, F32>, AddMask, ConvolutionForwardSpecialization::Filter1x1Stride1Pad0, FwdAddMask>(v); }
}
