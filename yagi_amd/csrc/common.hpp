// common.hpp -- shared plumbing of libyagi_hip.so: status/error reporting (the C-ABI rendering of
// yagi's error::Error, src/error.rs:4-14), HIP call checking, RAII device buffers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/yagi_hip.h"

namespace yagi {

using cf32 = yagi_cf32;

// thread-local message, like the String carried by every Error variant
void set_error(const char *fmt, ...);
int fail(int status, const char *fmt, ...);

#define YG_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ::yagi::fail(YAGI_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr,              \
                                hipGetErrorString(e_), __FILE__, __LINE__);                   \
    } while (0)

// Propagate a failed status.  It is also the exception barrier of the C ABI: the entry points reach every
// allocating member function (init, clone, prepare_*, ...) through YG_TRY, so a std::bad_alloc or any other
// C++ exception raised below is turned into YAGI_ERR_INTERNAL here instead of unwinding into a C / Rust caller.
int api_exception() noexcept;
#define YG_TRY(expr)                                                                          \
    do {                                                                                      \
        int s_;                                                                               \
        try {                                                                                 \
            s_ = (expr);                                                                      \
        } catch (...) {                                                                       \
            s_ = ::yagi::api_exception();                                                     \
        }                                                                                     \
        if (s_ != YAGI_OK) return s_;                                                         \
    } while (0)

#define YG_LAUNCH_CHECK() YG_HIP(hipGetLastError())

// Owning device allocation.  Non-copyable; grow() reallocates without preserving content.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        YG_HIP(hipMalloc(&p, n));
        bytes = n;
        return YAGI_OK;
    }
    int ensure(size_t n) { return (n <= bytes) ? YAGI_OK : alloc(n + n / 4); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

inline hipStream_t to_stream(yagi_stream_t s) { return static_cast<hipStream_t>(s); }

constexpr int kWave = 64;   // gfx950 wavefront width

}  // namespace yagi
