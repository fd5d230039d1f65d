#include "bwd_instances.h"
namespace probe {
void add_bwd_plain_1(std::vector<std::unique_ptr<BwdPlain>>& v) { add_bwd<ck::Tuple<>, ck::Tuple<>, PassThrough, ConvolutionBackwardDataSpecialization::Filter1x1Stride1Pad0, BwdPlain>(v); }
}
