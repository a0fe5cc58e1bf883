// Optional in-library kernel timing with HIP events (used by bench.py for the
// roofline figure): when enabled, every vg_gemm / vg_attn_* launch is bracketed
// by an event pair recorded on the launch stream, tagged with its kind and its
// ALGORITHMIC work (FLOPs).  vg_prof_read() synchronises the events and returns
// per-kind totals.  Disabled (the default) it costs one relaxed load per call.
#include <atomic>
#include <mutex>
#include <vector>

#include "vg_common.h"
#include "../../include/vaegslm_hip.h"

namespace {
struct Rec { hipEvent_t a, b; int kind; double work, bytes; };
std::atomic<int> g_on{0};
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
constexpr size_t MAX_RECS = 1 << 17;

hipEvent_t take_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}
}  // namespace

namespace vg_host {
int prof_begin(int kind, double work, hipStream_t stream, double bytes) {
  if (!g_on.load(std::memory_order_relaxed)) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_recs.size() >= MAX_RECS) return -1;
  Rec r{take_event(), take_event(), kind, work, bytes};
  hipEventRecord(r.a, stream);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}
void prof_end(int token, hipStream_t stream) {
  if (token < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if ((size_t)token < g_recs.size()) hipEventRecord(g_recs[token].b, stream);
}
}  // namespace vg_host

extern "C" int vg_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
  g_recs.clear();
  g_on.store(on ? 1 : 0);
  return 0;
}

extern "C" int vg_prof_read(int kind, double* total_ms, double* total_work, int* launches) {
  std::lock_guard<std::mutex> lk(g_mu);
  double ms = 0.0, work = 0.0;
  int n = 0;
  for (auto& r : g_recs) {
    if (r.kind != kind) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    ms += t; work += r.work; ++n;
  }
  if (total_ms) *total_ms = ms;
  if (total_work) *total_work = work;
  if (launches) *launches = n;
  return 0;
}

// summed ALGORITHMIC bytes (operands and results once each) of the recorded launches of one kind
extern "C" int vg_prof_read_bytes(int kind, double* total_bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  double bytes = 0.0;
  for (auto& r : g_recs)
    if (r.kind == kind) bytes += r.bytes;
  if (total_bytes) *total_bytes = bytes;
  return 0;
}
