// Error reporting, version and the built-in HIP-event kernel profiler of libpyannote_amd.so
// (C ABI declared in include/pyannote_amd.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace pa {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Pool of self-resetting tile-counter blocks (common.h): 4096 blocks of 16 ints per device, handed out in
// rotation -- on one stream a block is always free again when its turn comes (launches are ordered), and
// across streams 4096 launches would have to be in flight at once for two to meet.
int* tile_counters() {
  constexpr int MAXDEV = 16, BLOCKS = 4096;
  static std::mutex mu;
  static int* pool[MAXDEV] = {nullptr};
  static unsigned next[MAXDEV] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= MAXDEV) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  if (pool[dev] == nullptr) {
    if (hipMalloc((void**)&pool[dev], sizeof(int) * 16 * BLOCKS) != hipSuccess) return nullptr;
    (void)hipMemset(pool[dev], 0, sizeof(int) * 16 * BLOCKS);
  }
  return pool[dev] + 16 * (next[dev]++ % BLOCKS);
}

// ---------------------------------------------------------------------------------------------
// Profiler: when enabled, every kernel launcher brackets its launch with two hipEvents recorded on
// the SAME stream the kernel runs on (torch.cuda.Event would only see torch's current stream).
// pa_prof_report() synchronises, sums elapsed time / algorithmic flops / algorithmic bytes per kernel
// name and writes one JSON object.  Disabled (the default) it costs one relaxed load per launch.
// ---------------------------------------------------------------------------------------------
struct ProfRec {
  const char* name;
  hipEvent_t e0, e1;
  double flops, bytes;
};
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

ProfScope::ProfScope(const char* name, void* stream, double flops, double bytes) : idx_(-1), stream_(stream) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r{name, get_event(), get_event(), flops, bytes};
  (void)hipEventRecord(r.e0, (hipStream_t)stream);
  idx_ = (long)g_prof.size();
  g_prof.push_back(r);
}
ProfScope::~ProfScope() {
  if (idx_ < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_prof[idx_].e1, (hipStream_t)stream_);
}
}  // namespace pa

extern "C" {


int pa_version(void) { return 101; }
const char* pa_last_error(void) { return pa::g_err; }

void pa_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(pa::g_prof_mu);
  pa::g_prof_on = on != 0;
}

// Writes {"kernel": {"launches": n, "ms": total, "flops": total, "bytes": total}, ...} into buf
// (NUL-terminated, truncated to cap) and clears the records.  Returns the number of bytes needed.
size_t pa_prof_report(char* buf, size_t cap) {
  std::lock_guard<std::mutex> lk(pa::g_prof_mu);
  struct Agg {
    long n = 0;
    double ms = 0, flops = 0, bytes = 0;
  };
  std::map<std::string, Agg> agg;
  for (auto& r : pa::g_prof) {
    (void)hipEventSynchronize(r.e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    Agg& a = agg[r.name];
    a.n += 1;
    a.ms += ms;
    a.flops += r.flops;
    a.bytes += r.bytes;
    pa::g_pool.push_back(r.e0);
    pa::g_pool.push_back(r.e1);
  }
  pa::g_prof.clear();
  std::string s = "{";
  bool first = true;
  char tmp[256];
  for (auto& kv : agg) {
    snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"launches\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
             first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops,
             kv.second.bytes);
    s += tmp;
    first = false;
  }
  s += "}";
  if (buf && cap) {
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return s.size() + 1;
}
}
