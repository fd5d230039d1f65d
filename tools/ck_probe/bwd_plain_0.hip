#include "bwd_instances.h"
namespace probe {
void add_bwd_plain_0(std::vector<std::unique_ptr<BwdPlain>>& v) { add_bwd<ck::Tuple<>, ck::Tuple<>, PassThrough, ConvolutionBackwardDataSpecialization::Default, BwdPlain>(v); }
}
