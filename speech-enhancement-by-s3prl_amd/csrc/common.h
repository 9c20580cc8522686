// common.h -- host-side helpers shared by the C-ABI translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/se_amd.h"

namespace se {

void set_error(const char* fmt, ...);   // stores a thread-local message (se_last_error)
int hip_fail(hipError_t e, const char* what, const char* file, int line);   // -> SE_ERR_HIP / SE_ERR_OOM

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// clears a device buffer with a kernel (graph-capture safe; see zero.hip)
int zero_async(void* ptr, size_t bytes, hipStream_t st);
int zero_async3(void* p0, size_t b0, void* p1, size_t b1, void* p2, size_t b2, hipStream_t st);

}  // namespace se

#define SE_HIP(call)                                                        \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) return se::hip_fail(e_, #call, __FILE__, __LINE__); \
  } while (0)

#define SE_LAUNCH_CHECK() SE_HIP(hipGetLastError())

#define SE_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      se::set_error(__VA_ARGS__);             \
      return SE_ERR_INVALID;                  \
    }                                         \
  } while (0)
