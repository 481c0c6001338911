// Shared host-side helpers for libkinetica_hip (HIP runtime error handling, device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace kin {

struct KinError : std::runtime_error {
  int code;
  KinError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// status codes, mirrored in include/kinetica_hip.h
enum : int { OK = 0, ERR_INVALID_ARG = 1, ERR_UNSUPPORTED = 2, ERR_DEVICE = 3, ERR_SOLVE_FAILED = 4,
             ERR_CAPACITY = 5, ERR_STATE = 6 };

#define KIN_HIP(expr)                                                                       \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                \
      char b__[512];                                                                        \
      snprintf(b__, sizeof b__, "HIP error %d (%s) at %s:%d: %s", (int)e__,                 \
               hipGetErrorString(e__), __FILE__, __LINE__, #expr);                          \
      throw ::kin::KinError(::kin::ERR_DEVICE, b__);                                        \
    }                                                                                       \
  } while (0)

// Device buffer with value semantics disabled; grows on demand, never shrinks.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    if (count <= n && p) return;
    release();
    if (count == 0) count = 1;
    KIN_HIP(hipMalloc((void**)&p, count * sizeof(T)));
    n = count;
  }
  void upload(const std::vector<T>& h, hipStream_t s = nullptr) {
    alloc(h.size());
    if (!h.empty()) KIN_HIP(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void upload(const T* h, size_t count, hipStream_t s = nullptr) {
    alloc(count);
    if (count) KIN_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void download(T* h, size_t count, hipStream_t s = nullptr) const {
    if (count) KIN_HIP(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
  }
  void zero(hipStream_t s = nullptr) {
    if (p) KIN_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace kin
