// clkprobe.h -- developer build only (-DSE_AMD_CLKPROBE, tools/clk_probe.py): the clock the chip HOLDS inside a kernel.  Lane 0 of every workgroup
// records s_memtime (shader clock cycles) and s_memrealtime (constant 100 MHz) at the kernel's first and last instruction; the in-kernel clock is
// delta(memtime) / delta(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).  The product build compiles none of this (the macros are empty);
// the values go to a buffer of their own that no kernel reads.
#pragma once
#include <hip/hip_runtime.h>
#ifdef SE_AMD_CLKPROBE
#define SE_CLKPROBE_SLOTS 8192
#define SE_CLKPROBE_DECL(name)                                                                                   \
  __device__ unsigned long long name[4 * SE_CLKPROBE_SLOTS];                                                     \
  extern "C" int se_dev_##name(unsigned long long* host_out) {                                                   \
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(name), sizeof(unsigned long long) * 4 * SE_CLKPROBE_SLOTS) != hipSuccess) return 1; \
    void* p_ = nullptr;                      /* and cleared, so that the next read holds one kernel's workgroups only */ \
    if (hipGetSymbolAddress(&p_, HIP_SYMBOL(name)) != hipSuccess) return 1;                                      \
    return hipMemset(p_, 0, sizeof(unsigned long long) * 4 * SE_CLKPROBE_SLOTS) == hipSuccess ? 0 : 1;          \
  }
#define SE_CLKPROBE_BEGIN() \
  const unsigned long long cp_t0_ = __builtin_amdgcn_s_memtime(), cp_r0_ = __builtin_amdgcn_s_memrealtime()
#define SE_CLKPROBE_END(name)                                                                                    \
  do {                                                                                                           \
    const unsigned long long cp_t1_ = __builtin_amdgcn_s_memtime(), cp_r1_ = __builtin_amdgcn_s_memrealtime();   \
    if (threadIdx.x == 0) {                                                                                      \
      const unsigned cp_id_ = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) % SE_CLKPROBE_SLOTS; \
      name[4 * cp_id_] = cp_t0_; name[4 * cp_id_ + 1] = cp_r0_; name[4 * cp_id_ + 2] = cp_t1_; name[4 * cp_id_ + 3] = cp_r1_; \
    }                                                                                                            \
  } while (0)
#else
#define SE_CLKPROBE_DECL(name)
#define SE_CLKPROBE_BEGIN() do { } while (0)
#define SE_CLKPROBE_END(name) do { } while (0)
#endif
