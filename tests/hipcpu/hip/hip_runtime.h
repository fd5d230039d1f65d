// TEST INFRASTRUCTURE -- a minimal host stand-in for <hip/hip_runtime.h>.
//
// Lets a kernel source file of transferattack_amd/csrc be compiled with g++ and executed on the CPU so that the
// kernel's LOGIC (indexing, staging through "LDS", barriers, rounding order) can be checked bit for bit against the
// oracle in a container that has no GPU.  It says nothing about speed and is never part of the product: only tests/
// builds it (tests/hipcpu/hipcpu_build.py), the package cannot import it.
//
// Model: every lane of a workgroup is a fiber (its own stack, switched in user space) on one OS thread; __syncthreads() and the wave shuffles park
// the fiber until its group has arrived.  Workgroups are independent, so a small pool of OS threads runs them side by
// side; `__shared__` arrays are thread_local statics (one workgroup per OS thread at a time).  Only what the csrc
// kernels use is provided.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#ifdef HIPCPU_SINGLE_WORKER               /* sanitizer build: plain statics (instrumented), one worker thread */
#define __shared__ static
#else
#define __shared__ static thread_local    /* `extern __shared__` is rewritten to `extern thread_local` by hipcpu_build.py */
#endif

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipErrorNotSupported = 801 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipErrorNotSupported ? "not supported by the host stand-in" : "host stand-in"; }
/* there is no device clock here: the launch-timing entry points (ta_timing_begin / ta_timing_end) report an error */
static inline hipError_t hipEventCreate(hipEvent_t*) { return hipErrorNotSupported; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipErrorNotSupported; }
static inline hipError_t hipEventElapsedTime(float*, hipEvent_t, hipEvent_t) { return hipErrorNotSupported; }

namespace hipcpu {
struct Idx { unsigned x, y, z; };
const Idx& thread_idx();
const Idx& block_idx();
const Idx& block_dim();
const Idx& grid_dim();
void barrier();
float shfl_xor(float v, int lane_mask, int width);
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
}  // namespace hipcpu

#define threadIdx (hipcpu::thread_idx())
#define blockIdx (hipcpu::block_idx())
#define blockDim (hipcpu::block_dim())
#define gridDim (hipcpu::grid_dim())

static inline void __syncthreads() { hipcpu::barrier(); }
static inline float __shfl_xor(float v, int lane_mask, int width = 64) { return hipcpu::shfl_xor(v, lane_mask, width); }

/* one workgroup = one OS thread: LDS atomics need no hardware atomicity here */
/* global-memory flag shared by workgroups on different OS threads */
static inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline hipError_t hipMemsetAsync(void* p, int value, size_t bytes, hipStream_t) { memset(p, value, bytes); return hipSuccess; }
static inline int atomicMin(int* p, int v) { const int old = *p; if (v < old) *p = v; return old; }
static inline int atomicMax(int* p, int v) { const int old = *p; if (v > old) *p = v; return old; }

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * (uint64_t)b) >> 32); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }      /* only used on wave-uniform values */
#define __builtin_assume(cond) ((void)0)
namespace hipcpu { int wave_any(int pred); int readlane(int v, int src); void mfma_f32_16x16x4(float a, float b, float (&c)[4]); }
static inline int __builtin_amdgcn_readlane(int v, int lane) { return hipcpu::readlane(v, lane); }   /* every live lane calls it */
struct f32x4 {                                   /* clang's ext_vector_type(4) float: .x/.y/.z/.w and [] */
    float x, y, z, w;
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
static inline f32x4 hipcpu_mfma_f32_16x16x4(float a, float b, f32x4 c) {
    float d[4] = {c.x, c.y, c.z, c.w};
    hipcpu::mfma_f32_16x16x4(a, b, d);
    return f32x4{d[0], d[1], d[2], d[3]};
}
static inline int __any(int pred) { return hipcpu::wave_any(pred); }
template <class V> static inline V __builtin_elementwise_fma(V a, V b, V c) {     /* clang's vector fma, per element */
    return V{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)};
}
/* streaming hints have no meaning on the host */
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int64_t min(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t max(int64_t a, int64_t b) { return a > b ? a : b; }

#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
    hipcpu::launch((grid), (block), (smem), [&]() { kernel(__VA_ARGS__); })
