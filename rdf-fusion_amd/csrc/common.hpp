// common.hpp — shared host-side declarations of librdfgpu (error handling, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "../../include/rdfgpu.h"

namespace rdfgpu {

using u8 = uint8_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i64 = int64_t;

// Error carried across the engine as a C++ exception and turned into a status + thread-local
// message at the ABI (no exception ever crosses the C boundary).
struct Error : std::runtime_error {
  int status;
  Error(int st, const std::string& msg) : std::runtime_error(msg), status(st) {}
};

[[noreturn]] inline void fail(int status, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Error(status, buf);
}

#define RDFGPU_HIP(expr)                                                                       \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      int _st = (_e == hipErrorOutOfMemory) ? RDFGPU_ERR_OOM                                    \
                : (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? RDFGPU_ERR_NO_DEVICE \
                                                                          : RDFGPU_ERR_DEVICE;  \
      ::rdfgpu::fail(_st, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    }                                                                                          \
  } while (0)

void set_last_error(const std::string& msg);

// PERM[c][k] = G,S,P,O position stored at level k of permutation c
// (lib/storage/src/index/components.rs:63-85).
static constexpr int PERM[3][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}};

}  // namespace rdfgpu
