// Every kernel launch of the library goes through GIGA_LAUNCH, which also counts it: `giga_launch_count()` (C ABI,
// measurement hook) lets bench.py report launches per step without a profiler attached.  The counter is the library's only
// process-wide mutable word; it is never read by the compute path.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

namespace giga {
extern std::atomic<unsigned long long> g_launch_count;      // defined in giga_capi.hip
// measurement hook (giga_launch_probe): bracket the launch whose ordinal (value of g_launch_count after it is counted) equals
// g_probe_target with two HIP events on the launch's own stream, and remember the kernel expression as written at the launch site
extern std::atomic<unsigned long long> g_probe_target;      // 0 = no probe armed
extern void* g_probe_ev[2];
extern const char* volatile g_probe_name;
}
namespace giga {
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) ONCE per (kernel, device) instead of before every launch: the call takes a
// runtime lock and a code-object lookup, host time that a launch-bound caller (a single-scene plan: ~17 launches of a few
// microseconds each) pays on every kernel.  Lock-free table keyed by the kernel's host address and the current device.
constexpr unsigned DYN_LDS_SLOTS = 1024;                   // >> number of kernel instantiations x devices
inline std::atomic<uintptr_t> g_dyn_lds_keys[DYN_LDS_SLOTS];
inline std::atomic<int> g_dyn_lds_vals[DYN_LDS_SLOTS];
inline void dyn_lds_once(const void* kern, int bytes) {
    constexpr unsigned N = DYN_LDS_SLOTS;
    std::atomic<uintptr_t>* keys = g_dyn_lds_keys;
    std::atomic<int>* vals = g_dyn_lds_vals;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uintptr_t key = reinterpret_cast<uintptr_t>(kern) * 64u + (uintptr_t)(dev & 63) + 1u;
    unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 54) & (N - 1);
    for (unsigned probe = 0; probe < N; ++probe, h = (h + 1) & (N - 1)) {
        uintptr_t k = keys[h].load(std::memory_order_acquire);
        if (k == key) {
            if (vals[h].load(std::memory_order_relaxed) >= bytes) return;
            break;                                           // a larger request than recorded: set again below
        }
        if (k == 0) {
            uintptr_t expect = 0;
            if (keys[h].compare_exchange_strong(expect, key, std::memory_order_acq_rel) || expect == key) break;
        }
    }
    // (recorded only when the runtime accepted it: a failed set is retried by the next launch instead of being remembered as done.
    //  The record lives as long as the process: after hipDeviceReset the attribute is gone while the record stays -- a caller that
    //  resets devices calls giga_forget_device_state(), which runs dyn_lds_forget().)
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
        keys[h].load(std::memory_order_relaxed) == key)
        vals[h].store(bytes, std::memory_order_relaxed);
}
// forget every recorded limit (the keys stay: a slot is found again, its value 0 makes the next launch set the attribute anew)
inline void dyn_lds_forget() {
    for (unsigned i = 0; i < DYN_LDS_SLOTS; ++i) g_dyn_lds_vals[i].store(0, std::memory_order_relaxed);
}
}  // namespace giga
#define GIGA_LAUNCH_ARG1(a, ...) a
#define GIGA_LAUNCH_ARG5(a, b, c, d, e, ...) e
#define GIGA_LAUNCH(...)                                                                                                       \
    do {                                                                                                                       \
        const unsigned long long giga_n_ = ::giga::g_launch_count.fetch_add(1, std::memory_order_relaxed) + 1;                  \
        const bool giga_pr_ = giga_n_ == ::giga::g_probe_target.load(std::memory_order_acquire);   /* pairs with the release store */ \
        if (giga_pr_) {                                                                                                        \
            const char* giga_nm_ = hipKernelNameRefByPtr(reinterpret_cast<const void*>(GIGA_LAUNCH_ARG1(__VA_ARGS__)),           \
                                                         GIGA_LAUNCH_ARG5(__VA_ARGS__));                                      \
            ::giga::g_probe_name = giga_nm_ ? giga_nm_ : GIGA_LAUNCH_STR(GIGA_LAUNCH_ARG1(__VA_ARGS__));                        \
            (void)hipEventRecord(static_cast<hipEvent_t>(::giga::g_probe_ev[0]), GIGA_LAUNCH_ARG5(__VA_ARGS__));               \
        }                                                                                                                      \
        hipLaunchKernelGGL(__VA_ARGS__);                                                                                       \
        if (giga_pr_) (void)hipEventRecord(static_cast<hipEvent_t>(::giga::g_probe_ev[1]), GIGA_LAUNCH_ARG5(__VA_ARGS__));     \
    } while (0)
#define GIGA_LAUNCH_STR2(x) #x
#define GIGA_LAUNCH_STR(x) GIGA_LAUNCH_STR2(x)
