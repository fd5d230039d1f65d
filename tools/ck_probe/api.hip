// C entry points of the probe library (ctypes: tools/ck_conv_probe.py).  All tensors NHWC / KYXC (torch channels_last memory), fp32.
#include "ck_probe_common.h"
#include "ck/stream_config.hpp"

namespace probe {
struct Geom { int n, c, hi, wi, k, y, x, stride, pad, ho, wo; };

static Geom geom(int n, int c, int hi, int wi, int k, int y, int x, int stride, int pad) {
    return Geom{n, c, hi, wi, k, y, x, stride, pad, (hi + 2 * pad - y) / stride + 1, (wi + 2 * pad - x) / stride + 1};
}
using A5 = std::array<ck::index_t, 5>;
using A2 = std::array<ck::index_t, 2>;
static A5 in_len(const Geom& g) { return {1, g.n, g.c, g.hi, g.wi}; }
static A5 in_str(const Geom& g) { return {g.c, g.hi * g.wi * g.c, 1, g.wi * g.c, g.c}; }
static A5 w_len(const Geom& g) { return {1, g.k, g.c, g.y, g.x}; }
static A5 w_str(const Geom& g) { return {g.k * g.y * g.x * g.c, g.y * g.x * g.c, 1, g.x * g.c, g.c}; }
static A5 out_len(const Geom& g) { return {1, g.n, g.k, g.ho, g.wo}; }
static A5 out_str(const Geom& g) { return {g.k, g.ho * g.wo * g.k, 1, g.wo * g.k, g.k}; }
static A5 bias_str(const Geom& g) { return {g.k, 0, 1, 0, 0}; }

template <typename Base> struct Registry {
    std::vector<std::unique_ptr<Base>> ops[2];
    bool filled = false;
};
static Registry<FwdPlain> r_fwd_plain;
static Registry<FwdBias> r_fwd_bias;
static Registry<FwdBiasAdd> r_fwd_bias_add;
static Registry<BwdPlain> r_bwd_plain;
static Registry<BwdMask> r_bwd_mask;
static Registry<BwdAddMask> r_bwd_add_mask;

static void fill() {
    static bool done = false;
    if (done) return;
    done = true;
    add_fwd_plain_0(r_fwd_plain.ops[0]); add_fwd_plain_1(r_fwd_plain.ops[1]);
    add_fwd_bias_0(r_fwd_bias.ops[0]); add_fwd_bias_1(r_fwd_bias.ops[1]);
    add_fwd_bias_add_1(r_fwd_bias_add.ops[1]);
    add_bwd_plain_0(r_bwd_plain.ops[0]); add_bwd_plain_1(r_bwd_plain.ops[1]);
    add_bwd_mask_0(r_bwd_mask.ops[0]); add_bwd_mask_1(r_bwd_mask.ops[1]);
    add_bwd_add_mask_0(r_bwd_add_mask.ops[0]); add_bwd_add_mask_1(r_bwd_add_mask.ops[1]);
}

template <typename Op, typename Arg> static int launch(Op& op, Arg& arg, void* stream) {
    if (!op.IsSupportedArgument(arg.get())) return 1;
    auto invoker = op.MakeInvokerPointer();
    invoker->Run(arg.get(), StreamConfig{static_cast<hipStream_t>(stream), false});
    return 0;
}
}  // namespace probe

using namespace probe;

// kind: 0 fwd plain, 1 fwd bias+relu, 2 fwd bias+shortcut+relu, 3 bwd-data plain, 4 bwd-data mask, 5 bwd-data add+mask
extern "C" int ckp_count(int kind, int one) {
    fill();
    switch (kind) {
    case 0: return static_cast<int>(r_fwd_plain.ops[one].size());
    case 1: return static_cast<int>(r_fwd_bias.ops[one].size());
    case 2: return static_cast<int>(r_fwd_bias_add.ops[one].size());
    case 3: return static_cast<int>(r_bwd_plain.ops[one].size());
    case 4: return static_cast<int>(r_bwd_mask.ops[one].size());
    case 5: return static_cast<int>(r_bwd_add_mask.ops[one].size());
    }
    return 0;
}

static std::string g_name;
extern "C" const char* ckp_name(int kind, int one, int idx) {
    fill();
    switch (kind) {
    case 0: g_name = r_fwd_plain.ops[one][idx]->GetTypeString(); break;
    case 1: g_name = r_fwd_bias.ops[one][idx]->GetTypeString(); break;
    case 2: g_name = r_fwd_bias_add.ops[one][idx]->GetTypeString(); break;
    case 3: g_name = r_bwd_plain.ops[one][idx]->GetTypeString(); break;
    case 4: g_name = r_bwd_mask.ops[one][idx]->GetTypeString(); break;
    case 5: g_name = r_bwd_add_mask.ops[one][idx]->GetTypeString(); break;
    default: g_name = "?";
    }
    return g_name.c_str();
}

// forward: a = input [n, hi, wi, c], w = weight [k, y, x, c], d0 = bias [k], d1 = shortcut [n, ho, wo, k], e = output
// backward data: a = output gradient [n, ho, wo, k], w = weight, d0 = (other addend | activation), d1 = activation, e = input gradient
// returns 0 = launched, 1 = this instance does not take the problem, -1 = bad arguments
extern "C" int ckp_run(int kind, int one, int idx, const float* a, const float* w, const float* d0, const float* d1, float* e, int n,
                       int c, int hi, int wi, int k, int y, int x, int stride, int pad, void* stream) {
    fill();
    const Geom g = geom(n, c, hi, wi, k, y, x, stride, pad);
    const A2 st{stride, stride}, dil{1, 1}, pl{pad, pad}, pr{pad, pad};
    if (idx < 0 || idx >= ckp_count(kind, one)) return -1;
    switch (kind) {
    case 0: {
        auto& op = *r_fwd_plain.ops[one][idx];
        auto arg = op.MakeArgumentPointer(a, w, {}, e, in_len(g), in_str(g), w_len(g), w_str(g), {}, {}, out_len(g), out_str(g), st, dil, pl, pr,
                                          PassThrough{}, PassThrough{}, PassThrough{});
        return launch(op, arg, stream);
    }
    case 1: {
        auto& op = *r_fwd_bias.ops[one][idx];
        auto arg = op.MakeArgumentPointer(a, w, {d0}, e, in_len(g), in_str(g), w_len(g), w_str(g), {out_len(g)}, {bias_str(g)}, out_len(g),
                                          out_str(g), st, dil, pl, pr, PassThrough{}, PassThrough{}, BiasRelu{});
        return launch(op, arg, stream);
    }
    case 2: {
        auto& op = *r_fwd_bias_add.ops[one][idx];
        auto arg = op.MakeArgumentPointer(a, w, {d0, d1}, e, in_len(g), in_str(g), w_len(g), w_str(g), {out_len(g), out_len(g)},
                                          {bias_str(g), out_str(g)}, out_len(g), out_str(g), st, dil, pl, pr, PassThrough{}, PassThrough{},
                                          BiasAddRelu{});
        return launch(op, arg, stream);
    }
    case 3: {
        auto& op = *r_bwd_plain.ops[one][idx];
        auto arg = op.MakeArgumentPointer(a, w, {}, e, out_len(g), out_str(g), w_len(g), w_str(g), {}, {}, in_len(g), in_str(g), st, dil, pl, pr,
                                          PassThrough{}, PassThrough{}, PassThrough{});
        return launch(op, arg, stream);
    }
    case 4: {
        auto& op = *r_bwd_mask.ops[one][idx];
        auto arg = op.MakeArgumentPointer(a, w, {d0}, e, out_len(g), out_str(g), w_len(g), w_str(g), {in_len(g)}, {in_str(g)}, in_len(g),
                                          in_str(g), st, dil, pl, pr, PassThrough{}, PassThrough{}, Mask{});
        return launch(op, arg, stream);
    }
    case 5: {
        auto& op = *r_bwd_add_mask.ops[one][idx];
        auto arg = op.MakeArgumentPointer(a, w, {d0, d1}, e, out_len(g), out_str(g), w_len(g), w_str(g), {in_len(g), in_len(g)},
                                          {in_str(g), in_str(g)}, in_len(g), in_str(g), st, dil, pl, pr, PassThrough{}, PassThrough{}, AddMask{});
        return launch(op, arg, stream);
    }
    }
    return -1;
}
