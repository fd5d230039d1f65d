// Single-launch fused update for gfx950: get_momentum + update_delta (attack.py:124-128, 145-153) with g
// read from HBM exactly once (24 B/element, no second pass over g, no kernel boundary).
//
// Each workgroup owns one 3072-element tile of one image: it loads its slice of g into registers, reduces
// |g| (wave butterflies + LDS), and PUBLISHES the partial sum as an 8-byte {tag, value} granule with one
// agent-scope write-through store.  It then issues the loads of m, delta and x -- their HBM latency hides
// the exchange -- while wave 0 sweeps the image's granules with relaxed agent-scope loads until every tag
// equals this launch's generation (data-is-the-flag hand-off: no fences, placement independent, valid
// across the 8 XCDs' private L2s).  The partials are re-added in tile order (same order as the two-launch
// path -> bit-identical results), the update is applied from registers, and the last workgroup of an image
// to finish advances that image's generation so the next launch needs no memset.
//
// Progress: the workgroups of an image have consecutive linear ids; they are dispatched before any later
// image's, so the oldest incomplete image is always fully resident once its last tile is dispatched.  HIP
// does not promise dispatch order, so every spin is bounded: on timeout the error word of the sync buffer
// is set (checked by the host) instead of hanging the GPU.
#include "update_common.h"

namespace ta {

using gu32 = __attribute__((address_space(1))) unsigned int;
using gu64 = __attribute__((address_space(1))) unsigned long long;

struct FusedSync {          // layout of sync_ws for (n, tiles)
    unsigned int* gen;      // [n]      generation of the last completed launch, per image
    unsigned int* depart;   // [n]      workgroups of the image that finished reading the granules
    unsigned int* err;      // [1]      sticky: a spin timed out
    unsigned long long* gran;   // [n * tiles] {tag << 32 | float bits}
};

__host__ __device__ inline FusedSync carve(void* ws, int64_t n, int tiles) {
    FusedSync s;
    unsigned int* w = static_cast<unsigned int*>(ws);
    s.gen = w;
    s.depart = w + n;
    s.err = w + 2 * n;
    const int64_t words = 2 * n + 2;                       // keep the granules 8-byte aligned
    s.gran = reinterpret_cast<unsigned long long*>(w + ((words + 1) & ~int64_t(1)));
    return s;
}

constexpr unsigned kSpinLimit = 1u << 17;      // ~0.1 s of polling, then give up (sticky error word)

template <int VEC, bool HAS_V, bool HAS_MIN, bool HAS_MOUT, bool HAS_XADV>
__global__ __launch_bounds__(kBlock) void mi_update_fused_kernel(
    const float* __restrict__ g, const float* __restrict__ v, const float* m_in, float* m_out, float* delta,
    const float* __restrict__ x, float* __restrict__ x_adv, void* sync_ws, StepParams p, int64_t n, int64_t e,
    int tiles, int64_t img0) {
    __shared__ float lds[kBlock / kWave + 1];
    const FusedSync sync = carve(sync_ws, n, tiles);
    const int64_t img = img0 + blockIdx.y;
    const int tile = blockIdx.x;
    const int64_t base = img * e + static_cast<int64_t>(tile) * kTile;
    const int64_t left = e - static_cast<int64_t>(tile) * kTile;
    constexpr int S = Slots<VEC>::n;

    const unsigned tag = __hip_atomic_load((gu32*)(sync.gen + img), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT) + 1u;

    // ---- phase 1: this tile's slice of g stays in registers
    Pack<VEC> pg[S];
    bool full[S];
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int64_t off = (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
        full[u] = off + VEC <= left;
        if (full[u]) {
            pg[u].load(g + base + off);
            if (HAS_V) {
                Pack<VEC> pv;
                pv.load(v + base + off);
#pragma unroll
                for (int k = 0; k < VEC; ++k) pg[u][k] = pg[u][k] + pv[k];      // grad + variance, vmifgsm.py:89
            }
        }
    }
#pragma unroll
    for (int u = 0; u < S; ++u) {
        if (full[u]) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc += fabsf(pg[u][k]);
        } else if (VEC > 1) {
            const int64_t off = (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
            for (int64_t i = off; i < left && i < off + VEC; ++i)
                acc += fabsf(HAS_V ? g[base + i] + v[base + i] : g[base + i]);
        }
    }
    const float partial = block_sum(acc, lds);
    if (threadIdx.x == 0) {
        const unsigned long long granule = (static_cast<unsigned long long>(tag) << 32) | __float_as_uint(partial);
        __hip_atomic_store((gu64*)(sync.gran + img * tiles + tile), granule, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- phase 2: get the other operands moving before waiting on the exchange
    Pack<VEC> pm[S], pd[S], px[S];
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int64_t off = (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
        if (full[u]) {
            if (HAS_MIN) pm[u].load(m_in + base + off);
            pd[u].load(delta + base + off);
            px[u].load(x + base + off);
        }
    }

    // ---- phase 3: wave 0 sweeps the image's granules (one relaxed agent-scope load per lane and pass)
    if (threadIdx.x < kWave) {
        const int lane = threadIdx.x;
        float t = 0.0f;
        // once any exchange of this buffer has timed out, later workgroups do not wait again
        bool timed_out = __hip_atomic_load((gu32*)(sync.err), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        for (int i0 = 0; i0 < tiles; i0 += kWave) {         // lane-strided, tile order within a lane
            const int i = i0 + lane;
            float val = 0.0f;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                if (i < tiles) {
                    const unsigned long long gr = __hip_atomic_load(
                        (gu64*)(sync.gran + img * tiles + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = static_cast<unsigned>(gr >> 32) == tag;
                    val = __uint_as_float(static_cast<unsigned>(gr));
                }
                if (__all(ok) || timed_out) break;
                if (++spins > kSpinLimit) { timed_out = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (i < tiles) t += val;
        }
        t = wave_sum(t);
        if (lane == 0) {
            lds[kBlock / kWave] = t;
            if (timed_out)
                __hip_atomic_store((gu32*)(sync.err), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // departure: the last workgroup of the image re-arms it for the next launch
            const unsigned old = __hip_atomic_fetch_add((gu32*)(sync.depart + img), 1u, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
            if (old == static_cast<unsigned>(tiles) - 1u) {
                __hip_atomic_store((gu32*)(sync.depart + img), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store((gu32*)(sync.gen + img), tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
    const float mean = lds[kBlock / kWave] / static_cast<float>(e);

    // ---- phase 4: update from registers
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int64_t off = (static_cast<int64_t>(u) * kBlock + threadIdx.x) * VEC;
        if (full[u]) {
            Pack<VEC> om, od, oa;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float q = pg[u][k] / mean;
                const float mprev = HAS_MIN ? pm[u][k] : 0.0f;
                const float mn = mprev * p.decay + q;
                const float d = project(pd[u][k] + p.alpha * sign_of(mn), px[u][k], p.neg_eps, p.eps);
                om[k] = mn;
                od[k] = d;
                oa[k] = px[u][k] + d;
            }
            if (HAS_MOUT) om.store(m_out + base + off);
            od.store(delta + base + off);
            if (HAS_XADV) oa.store(x_adv + base + off);
        } else if (VEC > 1) {
            for (int64_t i = off; i < left && i < off + VEC; ++i) {
                const float gg = HAS_V ? g[base + i] + v[base + i] : g[base + i];
                const float q = gg / mean;
                const float mprev = HAS_MIN ? m_in[base + i] : 0.0f;
                const float mn = mprev * p.decay + q;
                const float xx = x[base + i];
                const float d = project(delta[base + i] + p.alpha * sign_of(mn), xx, p.neg_eps, p.eps);
                if (HAS_MOUT) m_out[base + i] = mn;
                delta[base + i] = d;
                if (HAS_XADV) x_adv[base + i] = xx + d;
            }
        }
    }
}

// how many whole images (tiles workgroups each) the device holds at once for this kernel
static int64_t resident_images(const void* kernel, int tiles) {
    static thread_local const void* last_kernel = nullptr;
    static thread_local int64_t last_blocks = 0;
    if (kernel != last_kernel) {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess) return 0;
        last_kernel = kernel;
        last_blocks = static_cast<int64_t>(cus) * per_cu;
    }
    return last_blocks / tiles;
}

}  // namespace ta

using namespace ta;

extern "C" int64_t ta_fused_sync_bytes(int64_t n, int64_t e) {
    if (n <= 0 || e <= 0) return 0;
    const int64_t tiles = ceil_div(e, kTile);
    const int64_t words = ((2 * n + 2 + 1) & ~int64_t(1));
    return words * 4 + n * tiles * 8;
}

// Reads and clears the sticky timeout word (host-synchronous; for tests and debugging, not the hot path).
extern "C" int ta_fused_sync_error(void* sync_ws, int64_t n, int64_t e, void* stream) {
    TA_REQUIRE(sync_ws && n > 0 && e > 0, "bad arguments");
    const FusedSync s = carve(sync_ws, n, static_cast<int>(ceil_div(e, kTile)));
    unsigned int host = 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t err = hipMemcpyAsync(&host, s.err, sizeof(host), hipMemcpyDeviceToHost, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (err != hipSuccess) {
        set_error("fused_sync_error: %s", hipGetErrorString(err));
        return static_cast<int>(err);
    }
    if (host != 0) {
        (void)hipMemsetAsync(s.err, 0, sizeof(host), st);
        set_error("mi_update_fused: inter-workgroup exchange timed out");
        return TA_EINVAL;
    }
    return 0;
}

extern "C" int ta_mi_update_fused(const float* g, const float* v, const float* m_in, float* m_out, float* delta,
                                  const float* x, float* x_adv, void* sync_ws, float decay, float alpha, float eps,
                                  int64_t n, int64_t e, void* stream) {
    TA_REQUIRE(n > 0 && e > 0 && n <= 65535, "bad batch (n=%lld, e=%lld)", (long long)n, (long long)e);
    TA_REQUIRE(g && delta && x && sync_ws, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tiles = static_cast<int>(ceil_div(e, kTile));
    const StepParams p{decay, alpha, -eps, eps};
    bool vec = e % kVec == 0;
    for (const void* ptr : {(const void*)g, (const void*)v, (const void*)m_in, (const void*)m_out, (const void*)delta,
                            (const void*)x, (const void*)x_adv})
        if (ptr && !aligned16(ptr)) vec = false;
    const int key = (v ? 8 : 0) | (m_in ? 4 : 0) | (m_out ? 2 : 0) | (x_adv ? 1 : 0);
    // The exchange needs every workgroup of an image resident at once.  A launch therefore never exceeds what the
    // device can hold (occupancy query x CU count, same stream => the device is otherwise drained): bigger
    // batches are cut into several launches of whole images.
    int64_t img0 = 0;
    int64_t chunk = 0;
#define TA_MF(VEC, HV, HMI, HMO, HXA)                                                                            \
    {                                                                                                            \
        auto kern = mi_update_fused_kernel<VEC, HV, HMI, HMO, HXA>;                                              \
        if (chunk == 0) chunk = resident_images(reinterpret_cast<const void*>(kern), tiles);                     \
        TA_REQUIRE(chunk > 0, "an image of %d tiles does not fit the device at once", tiles);                   \
        for (img0 = 0; img0 < n; img0 += chunk) {                                                                \
            const dim3 grid(tiles, static_cast<unsigned>(n - img0 < chunk ? n - img0 : chunk));                  \
            hipLaunchKernelGGL(kern, grid, dim3(kBlock), 0, st, g, v, m_in, m_out, delta, x, x_adv, sync_ws, p, n, e, \
                               tiles, img0);                                                                     \
        }                                                                                                        \
    }
#define TA_MF_CASES(VEC)                                       \
    switch (key) {                                             \
        case 0: TA_MF(VEC, false, false, false, false); break; \
        case 1: TA_MF(VEC, false, false, false, true); break;  \
        case 2: TA_MF(VEC, false, false, true, false); break;  \
        case 3: TA_MF(VEC, false, false, true, true); break;   \
        case 4: TA_MF(VEC, false, true, false, false); break;  \
        case 5: TA_MF(VEC, false, true, false, true); break;   \
        case 6: TA_MF(VEC, false, true, true, false); break;   \
        case 7: TA_MF(VEC, false, true, true, true); break;    \
        case 8: TA_MF(VEC, true, false, false, false); break;  \
        case 9: TA_MF(VEC, true, false, false, true); break;   \
        case 10: TA_MF(VEC, true, false, true, false); break;  \
        case 11: TA_MF(VEC, true, false, true, true); break;   \
        case 12: TA_MF(VEC, true, true, false, false); break;  \
        case 13: TA_MF(VEC, true, true, false, true); break;   \
        case 14: TA_MF(VEC, true, true, true, false); break;   \
        default: TA_MF(VEC, true, true, true, true); break;    \
    }
    if (vec) { TA_MF_CASES(4) } else { TA_MF_CASES(1) }
#undef TA_MF_CASES
#undef TA_MF
    return check_launch("mi_update_fused");
}
