// Shared host-side plumbing for libk2hip: error transport across the C ABI,
// HIP call checking, a grow-only device arena.
#pragma once
#include <atomic>
#include <type_traits>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/k2hip.h"
#include "errors.h"

namespace k2hip {

#define K2_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            ::k2hip::failf(K2HIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                                  \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: a host that opens one model per GPU in ONE process (the C ABI
// allows it) needs it on every device, so the "done" flag is kept per (call site, device).
struct LdsAttrOnce {
    std::atomic<bool> done[64] = {};   // (handles on several host threads reach a call site together: setting the attribute twice is harmless)
    template <typename F>
    void ensure(F* func, int bytes) {
        int dev = 0;
        K2_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || done[dev].load(std::memory_order_acquire)) return;
        K2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(func), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        done[dev].store(true, std::memory_order_release);
    }
};

// compute units of the calling thread's current device (256 on MI355X), cached per device
inline int device_cu_count() {
    static std::atomic<int> n[64] = {};
    int dev = 0;
    K2_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return 256;
    int v = n[dev].load(std::memory_order_relaxed);
    if (!v) {
        K2_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        v = v > 0 ? v : 256;
        n[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// Blocking copy / fill on the current device's utility stream (non-blocking flag; one stream and one mutex per device), never on the
// legacy (null) stream: a legacy-stream operation synchronises with every blocking stream of the device -- the HOST APPLICATION's
// streams included -- and is refused outright while any stream of the process is being captured into a hipGraph by someone else.
// Same arguments as hipMemcpy / hipMemset; like those on a non-blocking engine stream, they order with nothing but the host.
// (tunables.cpp)
hipError_t copy_blocking(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t fill_blocking(void* dst, int value, size_t bytes);

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Grow-only bump arena in device memory.  One per model; reset at the start of
// every fused call, so steady-state calls allocate nothing (Guideline 9: no
// hipMalloc inside the hot path once warm).
class Arena {
  public:
    ~Arena() { release(); }
    void release() {
        if (base_) (void)hipFree(base_);
        base_ = nullptr;
        cap_ = 0;
        off_ = 0;
    }
    void reset() { off_ = 0; high_ = 0; }
    // dry mode: take() only counts (returns offsets from a null base); used to size the arena
    void set_dry(bool d) { dry_ = d; }
    // Ensure capacity BEFORE taking pointers (growing invalidates them).
    void reserve(int64_t bytes) {
        if (bytes <= cap_) return;
        if (off_ != 0) failf(K2HIP_ERR_INVALID, "arena grown while in use");
        release();
        K2_HIP(hipMalloc(&base_, (size_t)bytes));
        cap_ = bytes;
    }
    template <typename T>
    T* take(int64_t n) {
        int64_t bytes = align_up((int64_t)sizeof(T) * (n > 0 ? n : 1), 256);
        if (dry_) {
            off_ += bytes;
            if (off_ > high_) high_ = off_;
            return nullptr;
        }
        if (off_ + bytes > cap_) failf(K2HIP_ERR_CAPACITY, "device arena exhausted: need %lld more bytes (cap %lld)",
                                       (long long)(off_ + bytes - cap_), (long long)cap_);
        T* p = reinterpret_cast<T*>(static_cast<char*>(base_) + off_);
        off_ += bytes;
        if (off_ > high_) high_ = off_;
        return p;
    }
    int64_t mark() const { return off_; }
    void rewind(int64_t m) { off_ = m; }
    int64_t capacity() const { return cap_; }
    const void* base() const { return base_; }
    int64_t high_water() const { return high_; }

  private:
    void* base_ = nullptr;
    int64_t cap_ = 0, off_ = 0, high_ = 0;
    bool dry_ = false;
};


#if defined(__HIPCC__)
// The wave's maximum in every lane, on data-parallel-primitive moves within rows of 16 lanes + four v_readlane (no LDS crossbar: a
// ds_bpermute butterfly is a dependent chain of ~900 cycles).  max is exact in any order, so this returns the butterfly's bits;
// sums keep their butterfly (their rounding depends on the order).  Whole-wave call sites only.
__device__ __forceinline__ float wave_max_dpp(float v) {
    auto mv = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false));
    };
    v = fmaxf(v, mv(v, std::integral_constant<int, 0xB1>{}));    // quad_perm [1,0,3,2]
    v = fmaxf(v, mv(v, std::integral_constant<int, 0x4E>{}));    // quad_perm [2,3,0,1]
    v = fmaxf(v, mv(v, std::integral_constant<int, 0x141>{}));   // row_half_mirror
    v = fmaxf(v, mv(v, std::integral_constant<int, 0x140>{}));   // row_mirror
    float r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
#pragma unroll
    for (int row = 1; row < 4; row++) r = fmaxf(r, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16 * row)));
    return r;
}
#endif

}  // namespace k2hip
