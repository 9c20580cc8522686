// prof.hip -- se_prof_*: per-kernel-family HIP-event timing, enabled only by bench.py's roofline pass.
// When disabled (the default) launches are untouched.  Events are recorded on the SAME stream as the kernel.
#include <mutex>
#include <vector>
#include "common.h"
#include "prof.h"

namespace se {
namespace {
struct Slot {
  hipEvent_t a, b;
  int kind;
  double work;
  bool closed;
};
std::mutex g_mu;
bool g_on = false;
std::vector<Slot> g_slots;          // event pool, grown on demand
size_t g_used = 0;
double g_ms[kProfKinds], g_work[kProfKinds];
long long g_n[kProfKinds];
}  // namespace

bool prof_on() { return g_on; }

int prof_begin(int kind, double work, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_used == g_slots.size()) {
    if (g_slots.size() >= (1u << 16)) return -1;
    Slot s;
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return -1;
    g_slots.push_back(s);
  }
  Slot& s = g_slots[g_used];
  s.kind = kind;
  s.work = work;
  s.closed = false;
  if (hipEventRecord(s.a, st) != hipSuccess) return -1;
  return (int)g_used++;
}

void prof_end(int slot, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (slot < 0 || (size_t)slot >= g_used) return;
  if (hipEventRecord(g_slots[slot].b, st) == hipSuccess) g_slots[slot].closed = true;
}
}  // namespace se

extern "C" int se_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(se::g_mu);
  se::g_on = on != 0;
  return SE_OK;
}

// Drains the recorded events (synchronises on them) into the per-kind totals, then returns the totals of `kind`.
extern "C" int se_prof_read(int kind, double* total_ms, double* total_work, long long* launches) {
  SE_REQUIRE(kind >= 0 && kind < se::kProfKinds, "se_prof_read: bad kind");
  std::lock_guard<std::mutex> lk(se::g_mu);
  for (size_t i = 0; i < se::g_used; ++i) {
    se::Slot& s = se::g_slots[i];
    if (!s.closed) continue;
    SE_HIP(hipEventSynchronize(s.b));
    float ms = 0.f;
    SE_HIP(hipEventElapsedTime(&ms, s.a, s.b));
    se::g_ms[s.kind] += ms;
    se::g_work[s.kind] += s.work;
    se::g_n[s.kind] += 1;
  }
  se::g_used = 0;
  if (total_ms) *total_ms = se::g_ms[kind];
  if (total_work) *total_work = se::g_work[kind];
  if (launches) *launches = se::g_n[kind];
  return SE_OK;
}

extern "C" int se_prof_reset(void) {
  std::lock_guard<std::mutex> lk(se::g_mu);
  se::g_used = 0;
  for (int k = 0; k < se::kProfKinds; ++k) {
    se::g_ms[k] = 0;
    se::g_work[k] = 0;
    se::g_n[k] = 0;
  }
  return SE_OK;
}
