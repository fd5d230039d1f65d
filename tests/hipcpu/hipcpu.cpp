// TEST INFRASTRUCTURE -- runtime half of tests/hipcpu/hip/hip_runtime.h (see there).
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <vector>
#include "hip/hip_runtime.h"

namespace hipcpu {

thread_local Idx thread_idx, block_idx, block_dim, grid_dim;

namespace {
pthread_barrier_t group_barrier;
std::vector<pthread_barrier_t> wave_barriers;
std::vector<float> wave_slots;              // [waves][64]
thread_local int wave_of_thread;
}  // namespace

void barrier() { pthread_barrier_wait(&group_barrier); }

float shfl_xor(float v, int lane_mask, int width) {
    (void)width;                            // the kernels only use full 64-lane butterflies
    const int lane = thread_idx.x & 63;
    float* slots = wave_slots.data() + 64 * wave_of_thread;
    slots[lane] = v;
    pthread_barrier_wait(&wave_barriers[wave_of_thread]);
    const float got = slots[lane ^ lane_mask];
    pthread_barrier_wait(&wave_barriers[wave_of_thread]);
    return got;
}

struct Job {
    dim3 grid, block;
    const std::function<void()>* body;
    unsigned lane;
};

static void* lane_main(void* arg) {
    const Job* job = static_cast<const Job*>(arg);
    const unsigned t = job->lane;
    thread_idx = Idx{t % job->block.x, (t / job->block.x) % job->block.y, t / (job->block.x * job->block.y)};
    block_dim = Idx{job->block.x, job->block.y, job->block.z};
    grid_dim = Idx{job->grid.x, job->grid.y, job->grid.z};
    wave_of_thread = static_cast<int>(t / 64);
    for (unsigned bz = 0; bz < job->grid.z; ++bz)
        for (unsigned by = 0; by < job->grid.y; ++by)
            for (unsigned bx = 0; bx < job->grid.x; ++bx) {
                block_idx = Idx{bx, by, bz};
                (*job->body)();
                pthread_barrier_wait(&group_barrier);       // next workgroup reuses the "LDS"
            }
    return nullptr;
}

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    (void)smem_bytes;                                       // dynamic LDS is a fixed 160 KiB array (build.py)
    const unsigned lanes = block.x * block.y * block.z;
    const unsigned waves = (lanes + 63) / 64;
    pthread_barrier_init(&group_barrier, nullptr, lanes);
    wave_barriers.resize(waves);
    wave_slots.assign(64 * waves, 0.0f);
    for (unsigned w = 0; w < waves; ++w) {
        const unsigned in_wave = (w + 1) * 64 <= lanes ? 64 : lanes - w * 64;
        pthread_barrier_init(&wave_barriers[w], nullptr, in_wave);
    }
    std::vector<Job> jobs(lanes);
    std::vector<pthread_t> threads(lanes);
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 256 * 1024);
    for (unsigned t = 0; t < lanes; ++t) {
        jobs[t] = Job{grid, block, &body, t};
        pthread_create(&threads[t], &attr, lane_main, &jobs[t]);
    }
    for (unsigned t = 0; t < lanes; ++t) pthread_join(threads[t], nullptr);
    pthread_attr_destroy(&attr);
    for (unsigned w = 0; w < waves; ++w) pthread_barrier_destroy(&wave_barriers[w]);
    pthread_barrier_destroy(&group_barrier);
}

}  // namespace hipcpu

// entry points of fused_update.hip: the single-launch exchange needs concurrently resident workgroups, which this
// one-workgroup-at-a-time model cannot provide
#include <stdint.h>
namespace ta { void set_error(const char* fmt, ...); }
extern "C" int64_t ta_fused_sync_bytes(int64_t, int64_t) { return 8; }
extern "C" int ta_mi_update_fused(const float*, const float*, const float*, float*, float*, const float*, float*, void*,
                                  float, float, float, int64_t, int64_t, void*) {
    ta::set_error("ta_mi_update_fused is not available in the host stand-in");
    return -1;
}
extern "C" int ta_fused_sync_error(void*, int64_t, int64_t, void*) { return 0; }
