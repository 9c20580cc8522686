// prof.h -- optional in-library kernel timing (HIP events on the launch stream) for bench.py's roofline leg.
#pragma once
#include <hip/hip_runtime.h>

namespace se {
enum ProfKind { kProfGemm = 0, kProfMhsa = 1, kProfStft = 2, kProfIstft = 3, kProfLayerNorm = 4, kProfHead = 5, kProfMhsaBwd = 6, kProfKinds = 7 };
bool prof_on();
// records a start event; returns a slot (or -1 when profiling is off / the pool is full)
int prof_begin(int kind, double work, hipStream_t st);
void prof_end(int slot, hipStream_t st);

struct ProfScope {
  int slot;
  hipStream_t st;
  ProfScope(int kind, double work, hipStream_t s) : slot(prof_on() ? prof_begin(kind, work, s) : -1), st(s) {}
  ~ProfScope() {
    if (slot >= 0) prof_end(slot, st);
  }
};
}  // namespace se
