// Streaming (HBM-bound) elementwise kernels of the FGSM-family loop for gfx950:
//   SIM / Admix scale-copy transforms + their backward   (sim.py:36-40, admix.py:40-45)
//   VMI-FGSM neighbour sampling / gradient accumulation  (vmifgsm.py:42-58)
//   NI look-ahead axpy                                    (nifgsm.py:35-39)
//   random-start init                                     (attack.py:130-143)
//   output quantiser  NCHW fp32 -> NHWC uint8             (utils.py:63-66)
// All use 16-byte lane accesses, three in flight per operand, one 3072-element tile per workgroup.
#include "philox.h"

namespace ta {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Generic flat driver: `body(i, vec)` is called with the element index of a full 4-group
// (vec=true) or of a single trailing / unaligned element (vec=false).
template <bool VEC, typename Body>
__device__ __forceinline__ void for_tile(int64_t numel, Body&& body) {
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
    if (VEC) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t i = base + (static_cast<int64_t>(u) * kBlock + threadIdx.x) * kVec;
            if (i + kVec <= numel) body(i, std::true_type{});
            else for (int64_t j = i; j < numel; ++j) body(j, std::false_type{});
        }
    } else {
#pragma unroll
        for (int u = 0; u < kUnroll * kVec; ++u) {
            const int64_t i = base + static_cast<int64_t>(u) * kBlock + threadIdx.x;
            if (i < numel) body(i, std::false_type{});
        }
    }
}

// ---- SIM ----------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kBlock) void scale_copies_fwd_kernel(const float* __restrict__ x,
                                                                  float* __restrict__ y, int64_t ne,
                                                                  int num_scale) {
    for_tile<VEC>(ne, [&](int64_t i, auto vec) {
        if constexpr (decltype(vec)::value) {
            const float4 a = ld4(x + i);
            float s = 1.0f;
            for (int c = 0; c < num_scale; ++c, s *= 0.5f) st4(y + c * ne + i, scale4(a, s));
        } else {
            const float a = x[i];
            float s = 1.0f;
            for (int c = 0; c < num_scale; ++c, s *= 0.5f) y[c * ne + i] = a * s;
        }
    });
}

// The backward kernels below run one image per blockIdx.y (tiles of 3072 elements, K1's layout) and also emit the
// per-tile sums of |gx| (ws, nullable): when one of them is the last kernel that writes the input gradient, the fused
// update takes mean|g| from these sums and reads g once.
// same per-thread order as abs_sum_partials_kernel (K1): the sums carry K1's bits
__device__ __forceinline__ void add_abs4(float& asum, float4 a) {
    asum += fabsf(a.x); asum += fabsf(a.y); asum += fabsf(a.z); asum += fabsf(a.w);
}
__device__ __forceinline__ void emit_tile_sum(float asum, float* __restrict__ ws, float* red) {
    const float total = block_sum(asum, red);
    if (ws != nullptr && threadIdx.x == 0) ws[static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x] = total;
}

// gx = sum_i gy_i / 2^i accumulated i = num_scale-1 .. 0 (the order autograd's input buffer sees)
template <bool VEC>
__global__ __launch_bounds__(kBlock) void scale_copies_bwd_kernel(const float* __restrict__ gy,
                                                                  float* __restrict__ gx, float* __restrict__ ws,
                                                                  int64_t ne, int64_t e, int num_scale) {
    __shared__ float red[kBlock / kWave];
    const int64_t img0 = static_cast<int64_t>(blockIdx.y) * e;
    float asum = 0.0f;
    for_tile<VEC>(e, [&](int64_t ii, auto vec) {
        const int64_t i = img0 + ii;
        const float top = ldexpf(1.0f, -(num_scale - 1));
        if constexpr (decltype(vec)::value) {
            float s = top;
            float4 acc = scale4(ld4(gy + (num_scale - 1) * ne + i), s);
            for (int c = num_scale - 2; c >= 0; --c) {
                s *= 2.0f;
                acc = add4(acc, scale4(ld4(gy + c * ne + i), s));
            }
            st4(gx + i, acc);
            add_abs4(asum, acc);
        } else {
            float s = top;
            float acc = gy[(num_scale - 1) * ne + i] * s;
            for (int c = num_scale - 2; c >= 0; --c) {
                s *= 2.0f;
                acc += gy[c * ne + i] * s;
            }
            gx[i] = acc;
            asum += fabsf(acc);
        }
    });
    emit_tile_sum(asum, ws, red);
}

// gx = sum_i gy_i accumulated i = copies-1 .. 0: backward of a stack of `copies` unit-gain views of x
// (EMI-FGSM's sample stack, emifgsm.py:57-58), in the order autograd's input buffer adds them
template <bool VEC>
__global__ __launch_bounds__(kBlock) void sum_copies_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                float* __restrict__ ws, int64_t ne, int64_t e, int copies) {
    __shared__ float red[kBlock / kWave];
    const int64_t img0 = static_cast<int64_t>(blockIdx.y) * e;
    float asum = 0.0f;
    for_tile<VEC>(e, [&](int64_t ii, auto vec) {
        const int64_t i = img0 + ii;
        if constexpr (decltype(vec)::value) {
            float4 acc = ld4(gy + (copies - 1) * ne + i);
            for (int c = copies - 2; c >= 0; --c) acc = add4(acc, ld4(gy + c * ne + i));
            st4(gx + i, acc);
            add_abs4(asum, acc);
        } else {
            float acc = gy[(copies - 1) * ne + i];
            for (int c = copies - 2; c >= 0; --c) acc += gy[c * ne + i];
            gx[i] = acc;
            asum += fabsf(acc);
        }
    });
    emit_tile_sum(asum, ws, red);
}

// gx = ((g[m-1] + g[m-2]) + ...) + g[0]: the members' input gradients of an EnsembleModel (utils.py:98-99: the same x
// feeds every member), added in the order autograd's input buffer receives them (last member first)
constexpr int kMaxMembers = 8;
struct MemberPtrs {
    const float* p[kMaxMembers];
};

template <bool VEC>
__global__ __launch_bounds__(kBlock) void sum_members_kernel(MemberPtrs gs, int m, float* __restrict__ gx,
                                                             float* __restrict__ ws, int64_t e) {
    __shared__ float red[kBlock / kWave];
    const int64_t img0 = static_cast<int64_t>(blockIdx.y) * e;
    float asum = 0.0f;
    for_tile<VEC>(e, [&](int64_t ii, auto vec) {
        const int64_t i = img0 + ii;
        if constexpr (decltype(vec)::value) {
            float4 acc = ld4(gs.p[m - 1] + i);
            for (int c = m - 2; c >= 0; --c) acc = add4(acc, ld4(gs.p[c] + i));
            st4(gx + i, acc);
            add_abs4(asum, acc);
        } else {
            float acc = gs.p[m - 1][i];
            for (int c = m - 2; c >= 0; --c) acc += gs.p[c][i];
            gx[i] = acc;
            asum += fabsf(acc);
        }
    });
    emit_tile_sum(asum, ws, red);
}

// ---- Admix --------------------------------------------------------------------------------------
// grid (tiles, n): one image per blockIdx.y so the permuted partner row is a single indirection
template <bool VEC>
__global__ __launch_bounds__(kBlock) void admix_fwd_kernel(const float* __restrict__ x,
                                                           const int64_t* __restrict__ perm,
                                                           float* __restrict__ y, int64_t n, int64_t e,
                                                           int num_admix, int num_scale, float strength) {
    const int64_t b = blockIdx.y;
    const float* xb = x + b * e;
    for_tile<VEC>(e, [&](int64_t i, auto vec) {
        for (int j = 0; j < num_admix; ++j) {
            const float* xp = x + perm[j * n + b] * e;
            if constexpr (decltype(vec)::value) {
                const float4 a = ld4(xb + i), p = ld4(xp + i);
                const float4 mixed = add4(a, scale4(p, strength));
                float s = 1.0f;
                for (int c = 0; c < num_scale; ++c, s *= 0.5f)
                    st4(y + ((static_cast<int64_t>(c) * num_admix + j) * n + b) * e + i, scale4(mixed, s));
            } else {
                const float mixed = xb[i] + xp[i] * strength;
                float s = 1.0f;
                for (int c = 0; c < num_scale; ++c, s *= 0.5f)
                    y[((static_cast<int64_t>(c) * num_admix + j) * n + b) * e + i] = mixed * s;
            }
        }
    });
}

// gx[b] = sum over j = num_admix-1..0 of ( sum over i = num_scale-1..0 of gy[i][j][b] / 2^i )
template <bool VEC>
__global__ __launch_bounds__(kBlock) void admix_bwd_kernel(const float* __restrict__ gy,
                                                           float* __restrict__ gx, float* __restrict__ ws,
                                                           int64_t n, int64_t e, int num_admix, int num_scale) {
    __shared__ float red[kBlock / kWave];
    const int64_t b = blockIdx.y;
    const float top = ldexpf(1.0f, -(num_scale - 1));
    float asum = 0.0f;
    for_tile<VEC>(e, [&](int64_t i, auto vec) {
        if constexpr (decltype(vec)::value) {
            float4 total;
            for (int j = num_admix - 1; j >= 0; --j) {
                float s = top;
                float4 acc = scale4(ld4(gy + ((static_cast<int64_t>(num_scale - 1) * num_admix + j) * n + b) * e + i), s);
                for (int c = num_scale - 2; c >= 0; --c) {
                    s *= 2.0f;
                    acc = add4(acc, scale4(ld4(gy + ((static_cast<int64_t>(c) * num_admix + j) * n + b) * e + i), s));
                }
                total = (j == num_admix - 1) ? acc : add4(total, acc);
            }
            st4(gx + b * e + i, total);
            add_abs4(asum, total);
        } else {
            float total = 0.0f;
            for (int j = num_admix - 1; j >= 0; --j) {
                float s = top;
                float acc = gy[((static_cast<int64_t>(num_scale - 1) * num_admix + j) * n + b) * e + i] * s;
                for (int c = num_scale - 2; c >= 0; --c) {
                    s *= 2.0f;
                    acc += gy[((static_cast<int64_t>(c) * num_admix + j) * n + b) * e + i] * s;
                }
                total = (j == num_admix - 1) ? acc : total + acc;
            }
            gx[b * e + i] = total;
            asum += fabsf(total);
        }
    });
    emit_tile_sum(asum, ws, red);
}

// ---- VMI / NI / init ----------------------------------------------------------------------------
template <bool VEC, bool HAS_NOISE>
__global__ __launch_bounds__(kBlock) void vmi_neighbor_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ delta,
                                                              const float* __restrict__ noise,
                                                              float* __restrict__ out, float radius,
                                                              uint64_t seed, uint64_t offset, int64_t numel) {
    for_tile<VEC>(numel, [&](int64_t i, auto vec) {
        if constexpr (decltype(vec)::value) {
            const float4 r = HAS_NOISE ? ld4(noise + i) : uniform4(static_cast<uint64_t>(i) >> 2, seed, offset, radius);
            st4(out + i, add4(add4(ld4(x + i), ld4(delta + i)), r));     // (x + d) + noise, vmifgsm.py:50
        } else {
            float r;
            if (HAS_NOISE) r = noise[i];
            else {
                const float4 q = uniform4(static_cast<uint64_t>(i) >> 2, seed, offset, radius);
                r = (&q.x)[i & 3];
            }
            out[i] = (x[i] + delta[i]) + r;
        }
    });
}

template <bool VEC, bool FIRST>
__global__ __launch_bounds__(kBlock) void grad_accumulate_kernel(float* acc, const float* __restrict__ g,
                                                                 int64_t numel) {
    for_tile<VEC>(numel, [&](int64_t i, auto vec) {
        if constexpr (decltype(vec)::value) st4(acc + i, FIRST ? ld4(g + i) : add4(ld4(acc + i), ld4(g + i)));
        else acc[i] = FIRST ? g[i] : acc[i] + g[i];
    });
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void variance_finalize_kernel(const float* __restrict__ acc,
                                                                   const float* __restrict__ cur,
                                                                   float* var, float count, int64_t numel) {
    for_tile<VEC>(numel, [&](int64_t i, auto vec) {
        if constexpr (decltype(vec)::value) {
            const float4 a = ld4(acc + i), c = ld4(cur + i);
            st4(var + i, make_float4(a.x / count - c.x, a.y / count - c.y, a.z / count - c.z, a.w / count - c.w));
        } else {
            var[i] = acc[i] / count - cur[i];
        }
    });
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void axpy_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                      float coeff, float* __restrict__ out, int64_t numel) {
    for_tile<VEC>(numel, [&](int64_t i, auto vec) {
        if constexpr (decltype(vec)::value) st4(out + i, add4(ld4(x + i), scale4(ld4(m + i), coeff)));
        else out[i] = x[i] + m[i] * coeff;
    });
}

template <bool VEC, bool HAS_NOISE>
__global__ __launch_bounds__(kBlock) void init_delta_kernel(float* __restrict__ delta,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ noise, float eps,
                                                            uint64_t seed, uint64_t offset, int64_t numel) {
    for_tile<VEC>(numel, [&](int64_t i, auto vec) {
        if constexpr (decltype(vec)::value) {
            const float4 r = HAS_NOISE ? ld4(noise + i) : uniform4(static_cast<uint64_t>(i) >> 2, seed, offset, eps);
            const float4 xx = ld4(x + i);
            st4(delta + i, make_float4(fminf(fmaxf(r.x, 0.0f - xx.x), 1.0f - xx.x), fminf(fmaxf(r.y, 0.0f - xx.y), 1.0f - xx.y),
                                       fminf(fmaxf(r.z, 0.0f - xx.z), 1.0f - xx.z), fminf(fmaxf(r.w, 0.0f - xx.w), 1.0f - xx.w)));
        } else {
            float r;
            if (HAS_NOISE) r = noise[i];
            else {
                const float4 q = uniform4(static_cast<uint64_t>(i) >> 2, seed, offset, eps);
                r = (&q.x)[i & 3];
            }
            delta[i] = fminf(fmaxf(r, 0.0f - x[i]), 1.0f - x[i]);
        }
    });
}

// ---- quantiser ----------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t q8(float x, float d) {
    return static_cast<uint32_t>(static_cast<int>((x + d) * 255.0f)) & 0xFFu;   // C truncation, utils.py:64
}

// C == 3, H*W % 4 == 0: a lane converts 4 consecutive pixels = 3 planes x float4 -> 12 bytes
__global__ __launch_bounds__(kBlock) void quantize_rgb_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ delta,
                                                              uint8_t* __restrict__ out, int64_t hw) {
    const int64_t img = blockIdx.y;
    const int64_t p = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) * 4;
    if (p >= hw) return;
    const float* xi = x + img * 3 * hw + p;
    const float* di = delta + img * 3 * hw + p;
    const float4 xr = ld4(xi), xg = ld4(xi + hw), xb = ld4(xi + 2 * hw);
    const float4 dr = ld4(di), dg = ld4(di + hw), db = ld4(di + 2 * hw);
    const uint32_t r0 = q8(xr.x, dr.x), g0 = q8(xg.x, dg.x), b0 = q8(xb.x, db.x);
    const uint32_t r1 = q8(xr.y, dr.y), g1 = q8(xg.y, dg.y), b1 = q8(xb.y, db.y);
    const uint32_t r2 = q8(xr.z, dr.z), g2 = q8(xg.z, dg.z), b2 = q8(xb.z, db.z);
    const uint32_t r3 = q8(xr.w, dr.w), g3 = q8(xg.w, dg.w), b3 = q8(xb.w, db.w);
    uint32_t* o = reinterpret_cast<uint32_t*>(out + (img * hw + p) * 3);     // 12-byte aligned: p % 4 == 0
    o[0] = r0 | (g0 << 8) | (b0 << 16) | (r1 << 24);
    o[1] = g1 | (b1 << 8) | (r2 << 16) | (g2 << 24);
    o[2] = b2 | (r3 << 8) | (g3 << 16) | (b3 << 24);
}

__global__ __launch_bounds__(kBlock) void quantize_generic_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ delta,
                                                                  uint8_t* __restrict__ out, int c,
                                                                  int64_t hw, int64_t total) {
    const int64_t o = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;   // NHWC output index
    if (o >= total) return;
    const int64_t ch = o % c, pix = (o / c) % hw, img = o / (c * hw);
    const int64_t i = (img * c + ch) * hw + pix;
    out[o] = static_cast<uint8_t>(q8(x[i], delta[i]));
}

static bool all16(std::initializer_list<const void*> ptrs) {
    for (const void* p : ptrs)
        if (p != nullptr && !aligned16(p)) return false;
    return true;
}

}  // namespace ta

using namespace ta;

#define TA_FLAT_GRID(numel) dim3(static_cast<unsigned>(ceil_div((numel), kTile)))

extern "C" int ta_scale_copies_fwd(const float* x, float* y, int64_t n, int64_t e, int num_scale, void* stream) {
    TA_REQUIRE(x && y && n > 0 && e > 0 && num_scale > 0 && num_scale < 31, "bad arguments");
    const int64_t ne = n * e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ne % 4 == 0 && all16({x, y}))
        hipLaunchKernelGGL(scale_copies_fwd_kernel<true>, TA_FLAT_GRID(ne), dim3(kBlock), 0, st, x, y, ne, num_scale);
    else
        hipLaunchKernelGGL(scale_copies_fwd_kernel<false>, TA_FLAT_GRID(ne), dim3(kBlock), 0, st, x, y, ne, num_scale);
    return check_launch("scale_copies_fwd");
}

#define TA_IMAGE_GRID(n, e) dim3(static_cast<unsigned>(ceil_div((e), kTile)), static_cast<unsigned>(n))

extern "C" int ta_scale_copies_bwd(const float* gy, float* gx, float* ws, int64_t n, int64_t e, int num_scale,
                                   void* stream) {
    TA_REQUIRE(gy && gx && n > 0 && n <= 65535 && e > 0 && num_scale > 0 && num_scale < 31, "bad arguments");
    const int64_t ne = n * e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (e % 4 == 0 && all16({gy, gx}))
        hipLaunchKernelGGL(scale_copies_bwd_kernel<true>, TA_IMAGE_GRID(n, e), dim3(kBlock), 0, st, gy, gx, ws, ne, e, num_scale);
    else
        hipLaunchKernelGGL(scale_copies_bwd_kernel<false>, TA_IMAGE_GRID(n, e), dim3(kBlock), 0, st, gy, gx, ws, ne, e, num_scale);
    return check_launch("scale_copies_bwd");
}

extern "C" int ta_sum_copies_bwd(const float* gy, float* gx, float* ws, int64_t n, int64_t e, int copies, void* stream) {
    TA_REQUIRE(gy && gx && n > 0 && n <= 65535 && e > 0 && copies > 0, "bad arguments");
    const int64_t ne = n * e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (e % 4 == 0 && all16({gy, gx}))
        hipLaunchKernelGGL(sum_copies_bwd_kernel<true>, TA_IMAGE_GRID(n, e), dim3(kBlock), 0, st, gy, gx, ws, ne, e, copies);
    else
        hipLaunchKernelGGL(sum_copies_bwd_kernel<false>, TA_IMAGE_GRID(n, e), dim3(kBlock), 0, st, gy, gx, ws, ne, e, copies);
    return check_launch("sum_copies_bwd");
}

extern "C" int ta_sum_members(const float* const* gs, int m, float* gx, float* ws, int64_t n, int64_t e, void* stream) {
    TA_REQUIRE(gs && gx && m >= 1 && m <= kMaxMembers && n > 0 && n <= 65535 && e > 0, "bad arguments (1 <= m <= 8)");
    MemberPtrs ptrs;
    bool vec = e % 4 == 0 && aligned16(gx);
    for (int k = 0; k < kMaxMembers; ++k) {
        ptrs.p[k] = k < m ? gs[k] : nullptr;
        TA_REQUIRE(k >= m || gs[k] != nullptr, "null member gradient");
        vec = vec && (k >= m || aligned16(gs[k]));
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (vec)
        hipLaunchKernelGGL(sum_members_kernel<true>, TA_IMAGE_GRID(n, e), dim3(kBlock), 0, st, ptrs, m, gx, ws, e);
    else
        hipLaunchKernelGGL(sum_members_kernel<false>, TA_IMAGE_GRID(n, e), dim3(kBlock), 0, st, ptrs, m, gx, ws, e);
    return check_launch("sum_members");
}

extern "C" int ta_admix_fwd(const float* x, const int64_t* perm, float* y, int64_t n, int64_t e, int num_admix,
                            int num_scale, float strength, void* stream) {
    TA_REQUIRE(x && perm && y && n > 0 && n <= 65535 && e > 0 && num_admix > 0 && num_scale > 0 && num_scale < 31,
               "bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(static_cast<unsigned>(ceil_div(e, kTile)), static_cast<unsigned>(n));
    if (e % 4 == 0 && all16({x, y}))
        hipLaunchKernelGGL(admix_fwd_kernel<true>, grid, dim3(kBlock), 0, st, x, perm, y, n, e, num_admix, num_scale, strength);
    else
        hipLaunchKernelGGL(admix_fwd_kernel<false>, grid, dim3(kBlock), 0, st, x, perm, y, n, e, num_admix, num_scale, strength);
    return check_launch("admix_fwd");
}

extern "C" int ta_admix_bwd(const float* gy, float* gx, float* ws, int64_t n, int64_t e, int num_admix, int num_scale,
                            void* stream) {
    TA_REQUIRE(gy && gx && n > 0 && n <= 65535 && e > 0 && num_admix > 0 && num_scale > 0 && num_scale < 31,
               "bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(static_cast<unsigned>(ceil_div(e, kTile)), static_cast<unsigned>(n));
    if (e % 4 == 0 && all16({gy, gx}))
        hipLaunchKernelGGL(admix_bwd_kernel<true>, grid, dim3(kBlock), 0, st, gy, gx, ws, n, e, num_admix, num_scale);
    else
        hipLaunchKernelGGL(admix_bwd_kernel<false>, grid, dim3(kBlock), 0, st, gy, gx, ws, n, e, num_admix, num_scale);
    return check_launch("admix_bwd");
}

extern "C" int ta_vmi_neighbor(const float* x, const float* delta, const float* noise, float* out, float radius,
                               uint64_t seed, uint64_t offset, int64_t numel, void* stream) {
    TA_REQUIRE(x && delta && out && numel > 0, "bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = all16({x, delta, noise, out});
#define TA_VN(V, N) hipLaunchKernelGGL((vmi_neighbor_kernel<V, N>), TA_FLAT_GRID(numel), dim3(kBlock), 0, st, x, delta, noise, out, radius, seed, offset, numel)
    if (vec) { if (noise) { TA_VN(true, true); } else { TA_VN(true, false); } }
    else { if (noise) { TA_VN(false, true); } else { TA_VN(false, false); } }
#undef TA_VN
    return check_launch("vmi_neighbor");
}

extern "C" int ta_grad_accumulate(float* acc, const float* g, int first, int64_t numel, void* stream) {
    TA_REQUIRE(acc && g && numel > 0, "bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = all16({acc, g});
#define TA_GA(V, F) hipLaunchKernelGGL((grad_accumulate_kernel<V, F>), TA_FLAT_GRID(numel), dim3(kBlock), 0, st, acc, g, numel)
    if (vec) { if (first) { TA_GA(true, true); } else { TA_GA(true, false); } }
    else { if (first) { TA_GA(false, true); } else { TA_GA(false, false); } }
#undef TA_GA
    return check_launch("grad_accumulate");
}

extern "C" int ta_variance_finalize(const float* acc, const float* cur_grad, float* var, float count, int64_t numel,
                                    void* stream) {
    TA_REQUIRE(acc && cur_grad && var && numel > 0, "bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (all16({acc, cur_grad, var}))
        hipLaunchKernelGGL(variance_finalize_kernel<true>, TA_FLAT_GRID(numel), dim3(kBlock), 0, st, acc, cur_grad, var, count, numel);
    else
        hipLaunchKernelGGL(variance_finalize_kernel<false>, TA_FLAT_GRID(numel), dim3(kBlock), 0, st, acc, cur_grad, var, count, numel);
    return check_launch("variance_finalize");
}

extern "C" int ta_axpy(const float* x, const float* m, float coeff, float* out, int64_t numel, void* stream) {
    TA_REQUIRE(x && m && out && numel > 0, "bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (all16({x, m, out}))
        hipLaunchKernelGGL(axpy_kernel<true>, TA_FLAT_GRID(numel), dim3(kBlock), 0, st, x, m, coeff, out, numel);
    else
        hipLaunchKernelGGL(axpy_kernel<false>, TA_FLAT_GRID(numel), dim3(kBlock), 0, st, x, m, coeff, out, numel);
    return check_launch("axpy");
}

extern "C" int ta_init_delta_uniform(float* delta, const float* x, const float* noise, float eps, uint64_t seed,
                                     uint64_t offset, int64_t numel, void* stream) {
    TA_REQUIRE(delta && x && numel > 0, "bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = all16({delta, x, noise});
#define TA_ID(V, N) hipLaunchKernelGGL((init_delta_kernel<V, N>), TA_FLAT_GRID(numel), dim3(kBlock), 0, st, delta, x, noise, eps, seed, offset, numel)
    if (vec) { if (noise) { TA_ID(true, true); } else { TA_ID(true, false); } }
    else { if (noise) { TA_ID(false, true); } else { TA_ID(false, false); } }
#undef TA_ID
    return check_launch("init_delta_uniform");
}

extern "C" int ta_quantize_u8_nhwc(const float* x, const float* delta, uint8_t* out, int64_t n, int c, int h, int w,
                                   void* stream) {
    TA_REQUIRE(x && delta && out && n > 0 && c > 0 && h > 0 && w > 0, "bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t hw = static_cast<int64_t>(h) * w;
    if (c == 3 && hw % 4 == 0 && n <= 65535 && all16({x, delta}) && (reinterpret_cast<uintptr_t>(out) & 3u) == 0) {
        const dim3 grid(static_cast<unsigned>(ceil_div(hw, kBlock * 4)), static_cast<unsigned>(n));
        hipLaunchKernelGGL(quantize_rgb_kernel, grid, dim3(kBlock), 0, st, x, delta, out, hw);
    } else {
        const int64_t total = n * c * hw;
        hipLaunchKernelGGL(quantize_generic_kernel, dim3(static_cast<unsigned>(ceil_div(total, kBlock))), dim3(kBlock), 0,
                           st, x, delta, out, c, hw, total);
    }
    return check_launch("quantize_u8_nhwc");
}
