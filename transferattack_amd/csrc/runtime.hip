// Error reporting, ABI bookkeeping and launch timing for libta_hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <vector>
#include "ta_common.h"

namespace ta {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(err));
        return static_cast<int>(err);
    }
    return 0;
}

// ---- summation order of the |g| sums (ta_set_sum_order): a process-wide setting handed in through the ABI, no environment
static std::atomic<int> g_sum_order_lanes{0};
int sum_order_lanes() { return g_sum_order_lanes.load(std::memory_order_relaxed); }

// ---- launch timing: a pool of event pairs handed to the fused update while armed
static std::mutex g_timing_mutex;
static std::vector<LaunchEvents> g_timing_pool;
static size_t g_timing_claimed = 0;
static bool g_timing_armed = false;

LaunchEvents claim_launch_events() {
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    if (!g_timing_armed || g_timing_claimed >= g_timing_pool.size()) return LaunchEvents{};
    return g_timing_pool[g_timing_claimed++];
}

static void release_timing_pool() {
    for (LaunchEvents& ev : g_timing_pool) {
        if (ev.start) (void)hipEventDestroy(ev.start);
        if (ev.stop) (void)hipEventDestroy(ev.stop);
    }
    g_timing_pool.clear();
    g_timing_claimed = 0;
    g_timing_armed = false;
}

}  // namespace ta

extern "C" int ta_timing_begin(int capacity) {
    TA_REQUIRE(capacity > 0 && capacity <= (1 << 20), "capacity %d", capacity);
    std::lock_guard<std::mutex> lock(ta::g_timing_mutex);
    TA_REQUIRE(!ta::g_timing_armed, "launch timing is already armed");
    ta::g_timing_pool.assign(static_cast<size_t>(capacity), ta::LaunchEvents{});
    for (ta::LaunchEvents& ev : ta::g_timing_pool) {
        hipError_t err = hipEventCreate(&ev.start);
        if (err == hipSuccess) err = hipEventCreate(&ev.stop);
        if (err != hipSuccess) {
            ta::set_error("hipEventCreate: %s", hipGetErrorString(err));
            ta::release_timing_pool();
            return static_cast<int>(err);
        }
    }
    ta::g_timing_claimed = 0;
    ta::g_timing_armed = true;
    return 0;
}

extern "C" int ta_timing_end(float* ms, int capacity, int* count) {
    TA_REQUIRE(ms && count && capacity >= 0, "null pointer");
    std::lock_guard<std::mutex> lock(ta::g_timing_mutex);
    TA_REQUIRE(ta::g_timing_armed, "launch timing is not armed");
    const size_t n = ta::g_timing_claimed < static_cast<size_t>(capacity) ? ta::g_timing_claimed : static_cast<size_t>(capacity);
    hipError_t err = hipSuccess;
    for (size_t i = 0; i < n && err == hipSuccess; ++i) {
        err = hipEventSynchronize(ta::g_timing_pool[i].stop);
        if (err == hipSuccess) err = hipEventElapsedTime(&ms[i], ta::g_timing_pool[i].start, ta::g_timing_pool[i].stop);
    }
    *count = static_cast<int>(n);
    ta::release_timing_pool();
    if (err != hipSuccess) {
        ta::set_error("launch timing: %s", hipGetErrorString(err));
        return static_cast<int>(err);
    }
    return 0;
}

extern "C" int ta_set_sum_order(int lanes) {
    TA_REQUIRE(lanes == 0 || lanes == 8 || lanes == 16, "sum order: 0 (kernel order), 8 or 16 (ATen's cascade), got %d", lanes);
    ta::g_sum_order_lanes.store(lanes, std::memory_order_relaxed);
    return 0;
}

extern "C" int ta_get_sum_order(void) { return ta::sum_order_lanes(); }

extern "C" int ta_abi_version(void) { return TA_ABI_VERSION; }
extern "C" const char* ta_last_error(void) { return ta::g_error; }
