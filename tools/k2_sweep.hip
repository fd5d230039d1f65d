// Stand-alone tuning probe for the fused update kernel (NOT part of libta_hip.so): times variants of the K2 inner
// structure on MI355X to choose the shipped configuration.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/k2_sweep.hip -o gpurun_out/k2_sweep && ./gpurun_out/k2_sweep
// Reports, per variant and batch size: mean kernel time (hipEvents over REPS launches, operands rotating over 4
// buffer sets so the 256 MiB Infinity Cache cannot hold them) and GB/s at 24 B/element.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int E = 3 * 224 * 224;

__device__ __forceinline__ float sign_of(float m) { return float(m > 0.f) - float(m < 0.f); }
__device__ __forceinline__ float project(float d, float x, float neg_eps, float eps) {
    d = fminf(fmaxf(d, neg_eps), eps);
    d = fmaxf(d, 0.0f - x);
    return fminf(d, 1.0f - x);
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

typedef float floatx4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 ld(const float* p) {
    if (NT) {
        const floatx4 v = __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const float4*>(p);
}
template <bool NT> __device__ __forceinline__ void st(float* p, float4 v) {
    if (NT) {
        floatx4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
        __builtin_nontemporal_store(w, reinterpret_cast<floatx4*>(p));
    } else *reinterpret_cast<float4*>(p) = v;
}

// one workgroup = UNROLL * BLOCK float4 of one image; ws holds `tiles` partial sums per image
template <int BLOCK, int UNROLL, bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(BLOCK) void k2(const float* __restrict__ g, float* m, float* delta,
                                            const float* __restrict__ x, const float* __restrict__ ws, int tiles_ws,
                                            float decay, float alpha, float eps) {
    constexpr int TILE = BLOCK * 4 * UNROLL;
    const long img = blockIdx.y;
    const long base = img * E + (long)blockIdx.x * TILE;
    const long left = E - (long)blockIdx.x * TILE;
    float4 pg[UNROLL], pm[UNROLL], pd[UNROLL], px[UNROLL];
    bool full[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const long off = ((long)u * BLOCK + threadIdx.x) * 4;
        full[u] = off + 4 <= left;
        if (full[u]) {
            pg[u] = ld<NT_LD>(g + base + off);
            pm[u] = ld<NT_LD>(m + base + off);
            pd[u] = ld<NT_LD>(delta + base + off);
            px[u] = ld<NT_LD>(x + base + off);
        }
    }
    const int lane = threadIdx.x & 63;
    float t = 0.f;
    for (int i = lane; i < tiles_ws; i += 64) t += ws[img * tiles_ws + i];
    const float mean = wave_sum(t) / (float)E;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const long off = ((long)u * BLOCK + threadIdx.x) * 4;
        if (!full[u]) continue;
        float4 om, od;
        float* gm = &pg[u].x; float* mm = &pm[u].x; float* dd = &pd[u].x; float* xx = &px[u].x;
        float* o1 = &om.x; float* o2 = &od.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float mn = mm[k] * decay + gm[k] / mean;
            o1[k] = mn;
            o2[k] = project(dd[k] + alpha * sign_of(mn), xx[k], -eps, eps);
        }
        st<NT_ST>(m + base + off, om);
        st<NT_ST>(delta + base + off, od);
    }
}

// grid-stride persistent form: `blocks` workgroups walk over all (image, tile) pairs
template <int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void k2_persistent(const float* __restrict__ g, float* m, float* delta,
                                                       const float* __restrict__ x, const float* __restrict__ ws,
                                                       int tiles_ws, int n, float decay, float alpha, float eps) {
    constexpr int TILE = BLOCK * 4;
    const int tiles = (E + TILE - 1) / TILE;
    const long total = (long)n * tiles;
    for (long w = blockIdx.x; w < total; w += gridDim.x) {
        const long img = w / tiles;
        const long off = (w % tiles) * TILE + (long)threadIdx.x * 4;
        if (off + 4 > E) continue;
        const long base = img * E + off;
        const float4 pg = ld<NT>(g + base), pm = ld<NT>(m + base), pd = ld<NT>(delta + base), px = ld<NT>(x + base);
        const int lane = threadIdx.x & 63;
        float t = 0.f;
        for (int i = lane; i < tiles_ws; i += 64) t += ws[img * tiles_ws + i];
        const float mean = wave_sum(t) / (float)E;
        float4 om, od;
        const float* gm = &pg.x; const float* mm = &pm.x; const float* dd = &pd.x; const float* xx = &px.x;
        float* o1 = &om.x; float* o2 = &od.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float mn = mm[k] * decay + gm[k] / mean;
            o1[k] = mn;
            o2[k] = project(dd[k] + alpha * sign_of(mn), xx[k], -eps, eps);
        }
        st<NT>(m + base, om);
        st<NT>(delta + base, od);
    }
}

// round 4: the byte source (x as uint8, rebuilt with the IEEE quotient's bits), with / without the x + delta write
__device__ __forceinline__ float u8_to_unit(unsigned k) {
    const float kf = (float)k, r255 = 1.0f / 255.0f, q = kf * r255;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, kf), r255, q);
}
template <int BLOCK, bool NT, bool XU8, bool XADV>
__global__ __launch_bounds__(BLOCK) void k2b(const float* __restrict__ g, float* m, float* delta, const float* __restrict__ x,
                                             const unsigned char* __restrict__ xb, float* __restrict__ xadv,
                                             const float* __restrict__ ws, int tiles_ws, float decay, float alpha, float eps) {
    constexpr int TILE = BLOCK * 4;
    const long img = blockIdx.y;
    const long off = (long)blockIdx.x * TILE + (long)threadIdx.x * 4;
    if (off + 4 > E) return;
    const long base = img * E + off;
    const float4 pg = ld<NT>(g + base), pm = ld<NT>(m + base), pd = ld<NT>(delta + base);
    float4 px;
    if (XU8) {
        const unsigned b = NT ? __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(xb + base))
                              : *reinterpret_cast<const unsigned*>(xb + base);
        px = make_float4(u8_to_unit(b & 255u), u8_to_unit((b >> 8) & 255u), u8_to_unit((b >> 16) & 255u), u8_to_unit(b >> 24));
    } else px = ld<NT>(x + base);
    const int lane = threadIdx.x & 63;
    float t = 0.f;
    for (int i = lane; i < tiles_ws; i += 64) t += ws[img * tiles_ws + i];
    const float mean = wave_sum(t) / (float)E;
    float4 om, od, oa;
    const float* gm = &pg.x; const float* mm = &pm.x; const float* dd = &pd.x; const float* xx = &px.x;
    float* o1 = &om.x; float* o2 = &od.x; float* o3 = &oa.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float mn = mm[k] * decay + gm[k] / mean;
        o1[k] = mn;
        o2[k] = project(dd[k] + alpha * sign_of(mn), xx[k], -eps, eps);
        o3[k] = xx[k] + o2[k];
    }
    st<NT>(m + base, om);
    st<NT>(delta + base, od);
    if (XADV) st<false>(xadv + base, oa);
}

// persistent + software-pipelined: a workgroup walks tiles w, w + G, ...; the loads of the next tile are in flight while
// the current one is computed and stored
template <int BLOCK, bool NT, bool XU8, bool XADV>
__global__ __launch_bounds__(BLOCK) void k2p(const float* __restrict__ g, float* m, float* delta, const float* __restrict__ x,
                                             const unsigned char* __restrict__ xb, float* __restrict__ xadv,
                                             const float* __restrict__ ws, int tiles_ws, int n, float decay, float alpha, float eps) {
    constexpr int TILE = BLOCK * 4;
    const int tiles = (E + TILE - 1) / TILE;
    const long total = (long)n * tiles;
    auto where = [&](long w, long& img, long& base, bool& ok) {
        img = w / tiles;
        const long off = (w % tiles) * TILE + (long)threadIdx.x * 4;
        ok = w < total && off + 4 <= E;
        base = img * E + off;
    };
    long img, base; bool ok;
    where(blockIdx.x, img, base, ok);
    float4 pg, pm, pd, px; unsigned pb = 0;
    auto fetch = [&](long b) {
        pg = ld<NT>(g + b); pm = ld<NT>(m + b); pd = ld<NT>(delta + b);
        if (XU8) pb = *reinterpret_cast<const unsigned*>(xb + b); else px = ld<NT>(x + b);
    };
    if (ok) fetch(base);
    for (long w = blockIdx.x; w < total; w += gridDim.x) {
        const float4 cg = pg, cm = pm, cd = pd; float4 cx = px; const unsigned cb = pb;
        const long cimg = img, cbase = base; const bool cok = ok;
        where(w + gridDim.x, img, base, ok);
        if (ok) fetch(base);
        if (!cok) continue;
        if (XU8) cx = make_float4(u8_to_unit(cb & 255u), u8_to_unit((cb >> 8) & 255u), u8_to_unit((cb >> 16) & 255u), u8_to_unit(cb >> 24));
        const int lane = threadIdx.x & 63;
        float t = 0.f;
        for (int i = lane; i < tiles_ws; i += 64) t += ws[cimg * tiles_ws + i];
        const float mean = wave_sum(t) / (float)E;
        float4 om, od, oa;
        const float* gm = &cg.x; const float* mm = &cm.x; const float* dd = &cd.x; const float* xx = &cx.x;
        float* o1 = &om.x; float* o2 = &od.x; float* o3 = &oa.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float mn = mm[k] * decay + gm[k] / mean;
            o1[k] = mn;
            o2[k] = project(dd[k] + alpha * sign_of(mn), xx[k], -eps, eps);
            o3[k] = xx[k] + o2[k];
        }
        st<NT>(m + cbase, om);
        st<NT>(delta + cbase, od);
        if (XADV) st<false>(xadv + cbase, oa);
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void copy4(const float* __restrict__ a, const float* __restrict__ b,
                                             const float* __restrict__ c, const float* __restrict__ d, float* o1,
                                             float* o2, long n4) {
    // same traffic shape as K2: 4 streams in, 2 streams out
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 va = ld<NT>(a + i * 4), vb = ld<NT>(b + i * 4), vc = ld<NT>(c + i * 4), vd = ld<NT>(d + i * 4);
        st<NT>(o1 + i * 4, make_float4(va.x + vb.x, va.y + vb.y, va.z + vb.z, va.w + vb.w));
        st<NT>(o2 + i * 4, make_float4(vc.x + vd.x, vc.y + vd.y, vc.z + vd.z, vc.w + vd.w));
    }
}

// ceilings of the memory system for other traffic shapes: 1-in/1-out float4 copy, and a pure 4-stream read
template <bool NT>
__global__ __launch_bounds__(256) void copy1(const float* __restrict__ a, float* o, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) st<NT>(o + i * 4, ld<NT>(a + i * 4));
}
template <bool NT>
__global__ __launch_bounds__(256) void read4(const float* __restrict__ a, const float* __restrict__ b,
                                             const float* __restrict__ c, const float* __restrict__ d, float* o, long n4) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 va = ld<NT>(a + i * 4), vb = ld<NT>(b + i * 4), vc = ld<NT>(c + i * 4), vd = ld<NT>(d + i * 4);
        acc += va.x + vb.y + vc.z + vd.w;
    }
    if (acc == 12345.678f) o[0] = acc;
}

struct Set { float *g, *m, *d, *x, *xa; unsigned char* xb; };

int main() {
    const int REPS = 40;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int n : {32, 125}) {
        const long numel = (long)n * E;
        std::vector<Set> sets(4);
        std::vector<float> host(numel);
        for (long i = 0; i < numel; ++i) host[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
        for (auto& s : sets) {
            for (float** p : {&s.g, &s.m, &s.d, &s.x, &s.xa}) {
                CHECK(hipMalloc(p, numel * 4));
                CHECK(hipMemcpy(*p, host.data(), numel * 4, hipMemcpyHostToDevice));
            }
            CHECK(hipMalloc(&s.xb, numel));
            CHECK(hipMemcpy(s.xb, host.data(), numel, hipMemcpyHostToDevice));
        }
        float* ws; CHECK(hipMalloc(&ws, 4 * n * 256)); CHECK(hipMemset(ws, 0, 4 * n * 256));
        std::vector<float> wsh(n * 256, 100.0f); CHECK(hipMemcpy(ws, wsh.data(), 4 * n * 256, hipMemcpyHostToDevice));
        auto run = [&](const char* name, auto launch) {
            for (int i = 0; i < 4; ++i) launch(sets[i % 4]);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < REPS; ++i) launch(sets[i % 4]);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / REPS;
            printf("n=%-4d %-34s %8.2f us  %7.1f GB/s\n", n, name, us, 24.0 * numel / us / 1e3);
        };
#define K2V(B, U, NL, NS)                                                                                      \
        run("k2 block" #B " unroll" #U " ntld" #NL " ntst" #NS, [&](Set& s) {                                  \
            const int tiles = (E + B * 4 * U - 1) / (B * 4 * U);                                               \
            hipLaunchKernelGGL((k2<B, U, NL, NS>), dim3(tiles, n), dim3(B), 0, 0, s.g, s.m, s.d, s.x, ws, 49,  \
                               1.0f, 0.00627f, 0.0627f);                                                       \
        })
        K2V(256, 3, false, false); K2V(256, 3, true, false); K2V(256, 3, false, true); K2V(256, 3, true, true);
        K2V(256, 1, false, false); K2V(256, 2, false, false); K2V(256, 4, false, false); K2V(256, 6, false, false);
        K2V(512, 1, false, false); K2V(512, 3, false, false); K2V(1024, 1, false, false); K2V(128, 3, false, false);
        K2V(256, 1, true, true); K2V(256, 2, true, true); K2V(512, 1, true, true);
        for (int blocks : {1024, 2048, 4096}) {
            char name[64]; snprintf(name, sizeof name, "k2 persistent %d blocks", blocks);
            run(name, [&](Set& s) { hipLaunchKernelGGL((k2_persistent<256, false>), dim3(blocks), dim3(256), 0, 0, s.g, s.m, s.d, s.x, ws, 49, n, 1.0f, 0.00627f, 0.0627f); });
            snprintf(name, sizeof name, "k2 persistent %d blocks nt", blocks);
            run(name, [&](Set& s) { hipLaunchKernelGGL((k2_persistent<256, true>), dim3(blocks), dim3(256), 0, 0, s.g, s.m, s.d, s.x, ws, 49, n, 1.0f, 0.00627f, 0.0627f); });
        }
        for (int blocks : {2048, 8192}) {
            char name[64]; snprintf(name, sizeof name, "copy 4in/2out %d blocks", blocks);
            run(name, [&](Set& s) { hipLaunchKernelGGL((copy4<false>), dim3(blocks), dim3(256), 0, 0, s.g, s.m, s.d, s.x, s.m, s.d, numel / 4); });
            snprintf(name, sizeof name, "copy 4in/2out %d blocks nt", blocks);
            run(name, [&](Set& s) { hipLaunchKernelGGL((copy4<true>), dim3(blocks), dim3(256), 0, 0, s.g, s.m, s.d, s.x, s.m, s.d, numel / 4); });
        }
        auto run_bytes = [&](const char* name, double bytes_per_elem, auto launch) {
            for (int i = 0; i < 4; ++i) launch(sets[i % 4]);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < REPS; ++i) launch(sets[i % 4]);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / REPS;
            printf("n=%-4d %-34s %8.2f us  %7.1f GB/s (at %.0f B/element)\n", n, name, us, bytes_per_elem * numel / us / 1e3, bytes_per_elem);
        };
        run_bytes("copy 1in/1out 8192 blocks", 8.0, [&](Set& s) { hipLaunchKernelGGL((copy1<false>), dim3(8192), dim3(256), 0, 0, s.g, s.m, numel / 4); });
        run_bytes("copy 1in/1out 8192 blocks nt", 8.0, [&](Set& s) { hipLaunchKernelGGL((copy1<true>), dim3(8192), dim3(256), 0, 0, s.g, s.m, numel / 4); });
        run_bytes("read 4 streams 8192 blocks", 16.0, [&](Set& s) { hipLaunchKernelGGL((read4<false>), dim3(8192), dim3(256), 0, 0, s.g, s.m, s.d, s.x, ws, numel / 4); });
        run_bytes("read 4 streams 8192 blocks nt", 16.0, [&](Set& s) { hipLaunchKernelGGL((read4<true>), dim3(8192), dim3(256), 0, 0, s.g, s.m, s.d, s.x, ws, numel / 4); });
        // ---- round 4 variants: bytes per element executed / algorithmic (24 contract, +4 with the x + delta write)
        auto run2 = [&](const char* name, double exec_b, double alg_b, auto launch) {
            for (int i = 0; i < 4; ++i) launch(sets[i % 4]);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < REPS; ++i) launch(sets[i % 4]);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / REPS;
            printf("n=%-4d %-44s %8.2f us  executed %4.0f B/el %7.1f GB/s   algorithmic %4.0f B/el %7.1f GB/s (%.3f of 8 TB/s)\n", n, name, us,
                   exec_b, exec_b * numel / us / 1e3, alg_b, alg_b * numel / us / 1e3, alg_b * numel / us / 1e3 / 8000.0);
        };
        const int t512 = (E + 2047) / 2048;
#define K2B(B, NTF, U8, XA)                                                                                               \
        run2("k2b block" #B " nt" #NTF " u8=" #U8 " xadv=" #XA, 20.0 + (U8 ? 1 : 4) + (XA ? 4 : 0), 24.0 + (XA ? 4 : 0), [&](Set& s) { \
            hipLaunchKernelGGL((k2b<B, NTF, U8, XA>), dim3((E + B * 4 - 1) / (B * 4), n), dim3(B), 0, 0, s.g, s.m, s.d, s.x, s.xb, s.xa, ws, 49, \
                               1.0f, 0.00627f, 0.0627f);                                                                  \
        })
        (void)t512;
        K2B(512, false, false, true); K2B(512, false, true, true); K2B(512, false, false, false); K2B(512, false, true, false);
        K2B(512, true, false, true); K2B(512, true, true, true); K2B(512, true, false, false); K2B(512, true, true, false);
        K2B(256, false, true, true); K2B(256, true, true, true); K2B(1024, false, true, true); K2B(1024, true, true, true);
        for (int blocks : {512, 1024, 1536, 2048}) {
            char name[96];
#define K2P(B, NTF, U8, XA)                                                                                               \
            snprintf(name, sizeof name, "k2p %d wgs block" #B " nt" #NTF " u8=" #U8 " xadv=" #XA, blocks);               \
            run2(name, 20.0 + (U8 ? 1 : 4) + (XA ? 4 : 0), 24.0 + (XA ? 4 : 0), [&](Set& s) {                              \
                hipLaunchKernelGGL((k2p<B, NTF, U8, XA>), dim3(blocks), dim3(B), 0, 0, s.g, s.m, s.d, s.x, s.xb, s.xa, ws, 49, n, \
                                   1.0f, 0.00627f, 0.0627f);                                                              \
            })
            K2P(512, false, true, true); K2P(512, true, true, true); K2P(256, false, true, true); K2P(512, false, true, false);
        }
        for (auto& s : sets) { for (float* p : {s.g, s.m, s.d, s.x, s.xa}) CHECK(hipFree(p)); CHECK(hipFree(s.xb)); }
        CHECK(hipFree(ws));
    }
    return 0;
}
