// C entry points of libta_ck.so (include/ta_ck.h).
#include <stdarg.h>
#include <mutex>
#include "ta_ck_common.h"
#include "ck/stream_config.hpp"

namespace ta_ck {
static thread_local char g_error[384] = "";
static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return TA_CK_EINVAL;
}

struct Geom { int n, c, hi, wi, k, y, x, stride, pad, ho, wo; };
using A5 = std::array<ck::index_t, 5>;
using A2 = std::array<ck::index_t, 2>;
// [G, N, C | K, H, W] lengths and strides of NHWGC / GKYXC / NHWGK memory with one group
static A5 in_len(const Geom& g) { return {1, g.n, g.c, g.hi, g.wi}; }
static A5 in_str(const Geom& g) { return {g.c, g.hi * g.wi * g.c, 1, g.wi * g.c, g.c}; }
static A5 w_len(const Geom& g) { return {1, g.k, g.c, g.y, g.x}; }
static A5 w_str(const Geom& g) { return {g.k * g.y * g.x * g.c, g.y * g.x * g.c, 1, g.x * g.c, g.c}; }
static A5 out_len(const Geom& g) { return {1, g.n, g.k, g.ho, g.wo}; }
static A5 out_str(const Geom& g) { return {g.k, g.ho * g.wo * g.k, 1, g.wo * g.k, g.k}; }
static A5 bias_str(const Geom& g) { return {g.k, 0, 1, 0, 0}; }

static std::vector<std::unique_ptr<FwdBias>> r_fwd_bias_any, r_fwd_bias_1x1;
static std::vector<std::unique_ptr<FwdBiasAdd>> r_fwd_bias_add_1x1;
static std::vector<std::unique_ptr<FwdBiasAddBias>> r_fwd_bias_add_bias_1x1;
static std::vector<std::unique_ptr<FwdMask>> r_fwd_mask_any, r_fwd_mask_1x1;
static std::vector<std::unique_ptr<FwdAddMask>> r_fwd_add_mask_any, r_fwd_add_mask_1x1;
static std::once_flag g_once;

static void fill() {
    std::call_once(g_once, [] {
        add_fwd_bias_any(r_fwd_bias_any);
        add_fwd_bias_1x1(r_fwd_bias_1x1);
        add_fwd_bias_add_1x1(r_fwd_bias_add_1x1);
        add_fwd_bias_add_bias_1x1(r_fwd_bias_add_bias_1x1);
        add_fwd_mask_any(r_fwd_mask_any);
        add_fwd_mask_1x1(r_fwd_mask_1x1);
        add_fwd_add_mask_any(r_fwd_add_mask_any);
        add_fwd_add_mask_1x1(r_fwd_add_mask_1x1);
    });
}

static bool is_1x1(int ksize, int stride, int pad) { return ksize == 1 && stride == 1 && pad == 0; }

template <typename Op, typename Arg> static int launch(Op& op, Arg& arg, void* stream) {
    if (!op.IsSupportedArgument(arg.get())) return TA_CK_UNSUPPORTED;
    op.MakeInvokerPointer()->Run(arg.get(), StreamConfig{static_cast<hipStream_t>(stream), false});
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        fail("launch: %s", hipGetErrorString(err));
        return static_cast<int>(err);
    }
    return 0;
}

// the registry a (kind, geometry) pair addresses: its size, and op(index) as a type-erased name
template <typename F> static int with_registry(int kind, int ksize, int stride, int pad, F&& f) {
    fill();
    const bool one = is_1x1(ksize, stride, pad);
    switch (kind) {
    case TA_CK_FWD_BIAS_RELU: return one ? f(r_fwd_bias_1x1) : f(r_fwd_bias_any);
    case TA_CK_FWD_BIAS_ADD_RELU: return one ? f(r_fwd_bias_add_1x1) : 0;
    case TA_CK_FWD_BIAS_ADD_BIAS_RELU: return one ? f(r_fwd_bias_add_bias_1x1) : 0;
    case TA_CK_FWD_MASK: return one ? f(r_fwd_mask_1x1) : f(r_fwd_mask_any);
    case TA_CK_FWD_ADD_MASK: return one ? f(r_fwd_add_mask_1x1) : f(r_fwd_add_mask_any);
    }
    return 0;
}
}  // namespace ta_ck

using namespace ta_ck;

extern "C" int ta_ck_abi_version(void) { return TA_CK_ABI_VERSION; }
extern "C" const char* ta_ck_last_error(void) { return g_error; }

extern "C" int ta_ck_instances(int kind, int ksize, int stride, int pad) {
    return with_registry(kind, ksize, stride, pad, [](auto& reg) { return static_cast<int>(reg.size()); });
}

extern "C" const char* ta_ck_instance_name(int kind, int ksize, int stride, int pad, int index) {
    static thread_local std::string name;
    name = "?";
    with_registry(kind, ksize, stride, pad, [&](auto& reg) {
        if (index >= 0 && index < static_cast<int>(reg.size())) name = reg[index]->GetTypeString();
        return 0;
    });
    return name.c_str();
}

extern "C" int ta_ck_conv(int kind, int index, const float* a, const float* w, const float* d0, const float* d1, const float* d2, float* e,
                          int n, int c, int hi, int wi, int k, int ksize, int stride, int pad, void* stream) {
    if (!a || !w || !e || !d0) return fail("null pointer");
    if (n <= 0 || c <= 0 || hi <= 0 || wi <= 0 || k <= 0 || ksize <= 0 || stride <= 0 || pad < 0 || hi + 2 * pad < ksize || wi + 2 * pad < ksize)
        return fail("bad shape (n=%d c=%d h=%d w=%d k=%d filter=%d stride=%d pad=%d)", n, c, hi, wi, k, ksize, stride, pad);
    const Geom g{n, c, hi, wi, k, ksize, ksize, stride, pad, (hi + 2 * pad - ksize) / stride + 1, (wi + 2 * pad - ksize) / stride + 1};
    if (static_cast<int64_t>(n) * hi * wi * c * 4 >= (1ll << 31) || static_cast<int64_t>(n) * g.ho * g.wo * k * 4 >= (1ll << 31))
        return fail("tensor of 2 GiB or more (32-bit byte offsets)");
    if (index < 0 || index >= ta_ck_instances(kind, ksize, stride, pad)) return fail("no configuration %d for kind %d and this filter", index, kind);
    const bool one = is_1x1(ksize, stride, pad);
    const A2 st{stride, stride}, dil{1, 1}, pl{pad, pad}, pr{pad, pad};
    switch (kind) {
    case TA_CK_FWD_BIAS_RELU: {
        auto& op = *(one ? r_fwd_bias_1x1 : r_fwd_bias_any)[index];
        auto arg = op.MakeArgumentPointer(a, w, {d0}, e, in_len(g), in_str(g), w_len(g), w_str(g), {out_len(g)}, {bias_str(g)}, out_len(g),
                                          out_str(g), st, dil, pl, pr, PassThrough{}, PassThrough{}, BiasRelu{});
        return launch(op, arg, stream);
    }
    case TA_CK_FWD_BIAS_ADD_RELU: {
        if (!d1) return fail("the shortcut is missing");
        auto& op = *r_fwd_bias_add_1x1[index];
        auto arg = op.MakeArgumentPointer(a, w, {d0, d1}, e, in_len(g), in_str(g), w_len(g), w_str(g), {out_len(g), out_len(g)},
                                          {bias_str(g), out_str(g)}, out_len(g), out_str(g), st, dil, pl, pr, PassThrough{}, PassThrough{},
                                          BiasAddRelu{});
        return launch(op, arg, stream);
    }
    case TA_CK_FWD_BIAS_ADD_BIAS_RELU: {
        if (!d1 || !d2) return fail("the shortcut or its bias is missing");
        auto& op = *r_fwd_bias_add_bias_1x1[index];
        auto arg = op.MakeArgumentPointer(a, w, {d0, d1, d2}, e, in_len(g), in_str(g), w_len(g), w_str(g), {out_len(g), out_len(g), out_len(g)},
                                          {bias_str(g), out_str(g), bias_str(g)}, out_len(g), out_str(g), st, dil, pl, pr, PassThrough{},
                                          PassThrough{}, BiasAddBiasRelu{});
        return launch(op, arg, stream);
    }
    case TA_CK_FWD_MASK: {
        auto& op = *(one ? r_fwd_mask_1x1 : r_fwd_mask_any)[index];
        auto arg = op.MakeArgumentPointer(a, w, {d0}, e, in_len(g), in_str(g), w_len(g), w_str(g), {out_len(g)}, {out_str(g)}, out_len(g),
                                          out_str(g), st, dil, pl, pr, PassThrough{}, PassThrough{}, Mask{});
        return launch(op, arg, stream);
    }
    case TA_CK_FWD_ADD_MASK: {
        if (!d1) return fail("the activation is missing");
        auto& op = *(one ? r_fwd_add_mask_1x1 : r_fwd_add_mask_any)[index];
        auto arg = op.MakeArgumentPointer(a, w, {d0, d1}, e, in_len(g), in_str(g), w_len(g), w_str(g), {out_len(g), out_len(g)},
                                          {out_str(g), out_str(g)}, out_len(g), out_str(g), st, dil, pl, pr, PassThrough{}, PassThrough{},
                                          AddMask{});
        return launch(op, arg, stream);
    }
    }
    return fail("unknown kind %d", kind);
}
