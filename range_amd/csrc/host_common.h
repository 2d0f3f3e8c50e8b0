// Host-side helpers shared by the translation units of librange_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/range_hip.h"

namespace range_host {

// text of the last failure on this thread, returned by range_last_error()
inline thread_local std::string g_err;

inline int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess)                                                             \
            return ::range_host::fail(RANGE_ERR_HIP, "%s failed: %s (%s:%d)", #expr,       \
                                      hipGetErrorString(e__), __FILE__, __LINE__);         \
    } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t ensure(size_t count) {
        if (count <= n) return hipSuccess;
        if (p) {
            // growth only happens between batches; make sure nothing still reads the old buffer
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) return e;
            (void)hipFree(p);
            p = nullptr;
            n = 0;
        }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t upload(const std::vector<T>& h) {
        hipError_t e = ensure(h.size());
        if (e != hipSuccess) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace range_host
