#include "fwd_instances.h"
namespace probe {
void add_fwd_plain_1(std::vector<std::unique_ptr<FwdPlain>>& v) { add_fwd<ck::Tuple<>, ck::Tuple<>, PassThrough, ConvFwd1x1S1P0, FwdPlain>(v); }
}
