#include "fwd_instances.h"
namespace ta_ck {
void add_fwd_mask_any(std::vector<std::unique_ptr<FwdMask>>& v) { add_fwd<ck::Tuple<NHWGK>, ck::Tuple<F32>, Mask, ConvolutionForwardSpecialization::Default, FwdMask>(v); }
}
