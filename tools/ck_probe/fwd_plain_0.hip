#include "fwd_instances.h"
namespace probe {
void add_fwd_plain_0(std::vector<std::unique_ptr<FwdPlain>>& v) { add_fwd<ck::Tuple<>, ck::Tuple<>, PassThrough, ConvFwdDefault, FwdPlain>(v); }
}
