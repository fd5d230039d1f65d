// TEST INFRASTRUCTURE -- runtime half of tests/hipcpu/hip/hip_runtime.h (see there).
#include <cstring>
#include <stdarg.h>
#include <stdio.h>
#include <sys/mman.h>
#include <atomic>
#include <thread>
#include <vector>
#include "hip/hip_runtime.h"

#ifdef HIPCPU_SINGLE_WORKER
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
#endif

#if !defined(__x86_64__)
#error "tests/hipcpu switches fibers with a few lines of x86-64 assembly (the build container and the GPU boxes are x86-64)"
#endif
// void hipcpu_switch(void** save_sp, void* load_sp): park the caller (callee-saved registers + stack pointer) and resume
// the context whose stack pointer is load_sp.  No signal-mask system call as in swapcontext -- a workgroup switches a
// few hundred times per barrier.
extern "C" void hipcpu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl hipcpu_switch
    .type hipcpu_switch, @function
hipcpu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipcpu_switch, .-hipcpu_switch
)");

namespace hipcpu {
namespace {

constexpr size_t kStackBytes = 128 * 1024;

struct Fiber {
    void* sp;
    Idx tid;
    bool done;
};

// one per OS thread of the pool: the lanes of the workgroup it is currently running
struct Worker {
    void* scheduler = nullptr;
    std::vector<Fiber> lanes;
    char* stacks = nullptr;
    Idx bid, bdim, gdim;
    unsigned current = 0, alive = 0;
    unsigned arrived = 0, generation = 0;                 // __syncthreads
    std::vector<unsigned> wave_arrived, wave_generation, wave_alive;
    std::vector<float> slots;                             // [waves][64] shuffle exchange
    const std::function<void()>* body = nullptr;
};

thread_local Worker* worker = nullptr;

void yield() {
    Worker* w = worker;
    hipcpu_switch(&w->lanes[w->current].sp, w->scheduler);
}

void release_if_complete(Worker* w) {
    if (w->alive != 0 && w->arrived == w->alive) {
        w->arrived = 0;
        ++w->generation;
    }
}

void release_wave_if_complete(Worker* w, unsigned wave) {
    if (w->wave_alive[wave] != 0 && w->wave_arrived[wave] == w->wave_alive[wave]) {
        w->wave_arrived[wave] = 0;
        ++w->wave_generation[wave];
    }
}

void wave_barrier(Worker* w, unsigned wave) {
    const unsigned gen = w->wave_generation[wave];
    ++w->wave_arrived[wave];
    release_wave_if_complete(w, wave);
    while (gen == w->wave_generation[wave]) yield();
}

void lane_entry() {                                       // first frame of every fiber; never returns
    Worker* w = worker;
    (*w->body)();
    const unsigned lane = w->current;
    w->lanes[lane].done = true;
    --w->alive;                                           // a lane that has left no longer holds up the others
    --w->wave_alive[lane / 64];
    release_if_complete(w);
    release_wave_if_complete(w, lane / 64);
    void* parked;
    hipcpu_switch(&parked, w->scheduler);
    __builtin_unreachable();
}

void run_workgroup(Worker* w, unsigned lanes) {
    const unsigned waves = (lanes + 63) / 64;
    w->alive = lanes;
    w->arrived = 0;
    w->wave_arrived.assign(waves, 0);
    w->wave_generation.assign(waves, 0);
    w->wave_alive.assign(waves, 0);
    for (unsigned t = 0; t < lanes; ++t) {
        Fiber& f = w->lanes[t];
        f.done = false;
        f.tid = Idx{t % w->bdim.x, (t / w->bdim.x) % w->bdim.y, t / (w->bdim.x * w->bdim.y)};
        ++w->wave_alive[t / 64];
        // initial frame: six callee-saved registers, then lane_entry as the return address of hipcpu_switch, placed so
        // that lane_entry starts with the stack alignment of a normal call (rsp + 8 divisible by 16)
        char* top = w->stacks + static_cast<size_t>(t + 1) * kStackBytes;
        void** slot = reinterpret_cast<void**>(top - 16);
        slot[0] = reinterpret_cast<void*>(&lane_entry);
        slot[1] = nullptr;
        for (int k = 1; k <= 6; ++k) slot[-k] = nullptr;
        f.sp = slot - 6;
    }
    // HIPCPU_ORDER=reverse runs the lanes of a workgroup last-to-first between barriers: a kernel whose result depends on
    // the order in which lanes of DIFFERENT waves touch LDS / memory between two barriers (a missing barrier) gives a
    // different answer under the two orders
    static const bool reverse = []() {
        const char* e = getenv("HIPCPU_ORDER");
        return e != nullptr && e[0] == 'r';
    }();
    unsigned remaining = lanes;
    while (remaining != 0)
        for (unsigned i = 0; i < lanes; ++i) {
            const unsigned t = reverse ? lanes - 1 - i : i;
            if (w->lanes[t].done) continue;
            w->current = t;
            hipcpu_switch(&w->scheduler, w->lanes[t].sp);
            if (w->lanes[t].done) --remaining;
        }
}

}  // namespace

const Idx& thread_idx() { return worker->lanes[worker->current].tid; }
const Idx& block_idx() { return worker->bid; }
const Idx& block_dim() { return worker->bdim; }
const Idx& grid_dim() { return worker->gdim; }

void barrier() {
    Worker* w = worker;
    const unsigned gen = w->generation;
    ++w->arrived;
    release_if_complete(w);
    while (gen == w->generation) yield();
}

float shfl_xor(float v, int lane_mask, int width) {
    (void)width;                                          // the kernels only use full 64-lane butterflies
    Worker* w = worker;
    const unsigned t = w->current, wave = t / 64, lane = t & 63;
    w->slots[64 * wave + lane] = v;
    wave_barrier(w, wave);
    const float got = w->slots[64 * wave + (lane ^ static_cast<unsigned>(lane_mask))];
    wave_barrier(w, wave);
    return got;
}

// v_mfma_f32_16x16x4_f32 on one wave: D[16x16] = C + A[16x4] . B[4x16], a k-ordered fmaf chain per element (what the
// instruction computes, bit for bit); lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15], D[4 * (l >> 4) + reg][l & 15]
void mfma_f32_16x16x4(float a, float b, float (&c)[4]) {
    Worker* w = worker;
    const unsigned t = w->current, wave = t / 64, lane = t & 63;
    float arow[4][4], bcol[4];
    w->slots[64 * wave + lane] = a;
    wave_barrier(w, wave);
    for (unsigned reg = 0; reg < 4; ++reg)
        for (unsigned k = 0; k < 4; ++k) arow[reg][k] = w->slots[64 * wave + 16 * k + 4 * (lane >> 4) + reg];
    wave_barrier(w, wave);
    w->slots[64 * wave + lane] = b;
    wave_barrier(w, wave);
    for (unsigned k = 0; k < 4; ++k) bcol[k] = w->slots[64 * wave + 16 * k + (lane & 15)];
    wave_barrier(w, wave);
    for (unsigned reg = 0; reg < 4; ++reg)
        for (unsigned k = 0; k < 4; ++k) c[reg] = fmaf(arow[reg][k], bcol[k], c[reg]);
}

// v_readlane_b32: every (live) lane of the wave calls it with the same source lane and gets that lane's value
int readlane(int v, int src) {
    Worker* w = worker;
    const unsigned t = w->current, wave = t / 64, lane = t & 63;
    float bits;
    memcpy(&bits, &v, 4);
    w->slots[64 * wave + lane] = bits;
    wave_barrier(w, wave);
    const float got = w->slots[64 * wave + (static_cast<unsigned>(src) & 63u)];
    wave_barrier(w, wave);
    int out;
    memcpy(&out, &got, 4);
    return out;
}

int wave_any(int pred) {                                  // every lane of the wave must call it (as on the device)
    Worker* w = worker;
    const unsigned t = w->current, wave = t / 64, lane = t & 63;
    w->slots[64 * wave + lane] = pred ? 1.0f : 0.0f;
    wave_barrier(w, wave);
    int any = 0;
    for (unsigned l = 0; l < 64; ++l) {
        const unsigned other = 64 * wave + l;
        if (other < w->lanes.size() && !w->lanes[other].done && w->slots[other] != 0.0f) any = 1;
    }
    wave_barrier(w, wave);
    return any;
}

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    (void)smem_bytes;                                     // dynamic LDS is a fixed 160 KiB array (hipcpu_build.py)
    const unsigned lanes = block.x * block.y * block.z;
    const uint64_t groups = static_cast<uint64_t>(grid.x) * grid.y * grid.z;
    unsigned pool = std::thread::hardware_concurrency();
    if (pool == 0) pool = 1;
#ifdef HIPCPU_SINGLE_WORKER
    pool = 1;                                             // `__shared__` arrays are process-wide statics in this build
#endif
    if (pool > groups) pool = static_cast<unsigned>(groups);
    std::atomic<uint64_t> next{0};
    auto work = [&]() {
        Worker w;
        w.lanes.resize(lanes);
        w.slots.assign(64 * ((lanes + 63) / 64), 0.0f);
        w.stacks = static_cast<char*>(mmap(nullptr, kStackBytes * lanes, PROT_READ | PROT_WRITE,
                                           MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
        w.bdim = Idx{block.x, block.y, block.z};
        w.gdim = Idx{grid.x, grid.y, grid.z};
        w.body = &body;
        worker = &w;
        for (uint64_t g = next.fetch_add(1); g < groups; g = next.fetch_add(1)) {
            w.bid = Idx{static_cast<unsigned>(g % grid.x), static_cast<unsigned>((g / grid.x) % grid.y),
                        static_cast<unsigned>(g / (static_cast<uint64_t>(grid.x) * grid.y))};
            run_workgroup(&w, lanes);
        }
        worker = nullptr;
#ifdef HIPCPU_SINGLE_WORKER
        __asan_unpoison_memory_region(w.stacks, kStackBytes * lanes);    // fibers leave stack red zones in the shadow
#endif
        munmap(w.stacks, kStackBytes * lanes);
    };
    std::vector<std::thread> threads;
    for (unsigned i = 1; i < pool; ++i) threads.emplace_back(work);
    work();
    for (auto& th : threads) th.join();
}

}  // namespace hipcpu
