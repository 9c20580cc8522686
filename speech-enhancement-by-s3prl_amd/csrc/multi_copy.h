// multi_copy.h -- batches many small device-to-device fp32 copies (or fp32 -> zero-padded bf16 casts) into single launches; the
// tensor table travels in the kernel arguments (optim.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace se {

struct CopyBatch;

class MultiCopy {
 public:
  MultiCopy(bool cast, hipStream_t st);
  ~MultiCopy();
  MultiCopy(const MultiCopy&) = delete;
  MultiCopy& operator=(const MultiCopy&) = delete;
  // copy: n fp32 elements.  cast: n = rows * ld bf16 outputs of a (rows, cols) fp32 source, rows zero-padded to ld (ld % 4 == 0)
  int add(void* dst, const float* src, size_t n, uint32_t cols = 0, uint32_t ld = 0);
  int flush();       // launches what has been added (asynchronous)

 private:
  CopyBatch* b_;
  uint32_t chunks_;
  bool cast_;
  hipStream_t st_;
};

}  // namespace se
