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

// ---------------------------------------------------------------------------------------------
// HOST helper: the weight image pa_conv3x3_wino expects, from the reference's conv weight.
// U = G g G^T (Winograd F(2x2,3x3), float64 arithmetic, BatchNorm scale folded per output channel), then
// packed as one contiguous 32-KB slab per (32-cout slice, 16-cin stage): [cout/32][cin/16][row = 32 xi + n]
// [slot][4] with xi = 4a + b, n = cout % 32, and channel quad q of the stage at physical slot
// (q + 2 ((row >> 2) & 1)) & 3 (the bank-conflict-free LDS image of emb_winograd.hip, so that the kernel's
// LDS-DMA is a linear stream).  Same result as weights.winograd_pack(winograd_weights(w * scale)).
// ---------------------------------------------------------------------------------------------
extern "C" int pa_winograd_pack_host(const float* conv_weight, const float* bn_scale, int cout, int cin,
                                     float* U_slabs) {
  if (cout <= 0 || cin <= 0 || cout % 32 != 0 || cin % 16 != 0 || !conv_weight || !U_slabs) {
    pa::set_error("pa_winograd_pack_host: cout %% 32 == 0 and cin %% 16 == 0 required (got %d, %d)", cout, cin);
    return 3;
  }
  static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  const int stages = cin / 16;
  for (int o = 0; o < cout; ++o) {
    for (int i = 0; i < cin; ++i) {
      double g[3][3];
      for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) {
          // the product kernel folds BN in float32 (weights.py: conv.weight * scale[:, None, None, None])
          const float folded = bn_scale ? conv_weight[((size_t)o * cin + i) * 9 + p * 3 + q] * bn_scale[o]
                                        : conv_weight[((size_t)o * cin + i) * 9 + p * 3 + q];
          g[p][q] = (double)folded;
        }
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
          double u = 0.0;
          for (int p = 0; p < 3; ++p)
            for (int q = 0; q < 3; ++q) u += G[a][p] * g[p][q] * G[b][q];
          const int xi = 4 * a + b, row = 32 * xi + (o % 32);
          const int quad = (i % 16) / 4, slot = (quad + 2 * ((row >> 2) & 1)) & 3;
          const size_t slab = ((size_t)(o / 32) * stages + i / 16) * 512 * 16;
          U_slabs[slab + (size_t)row * 16 + slot * 4 + (i % 4)] = (float)u;
        }
    }
  }
  return 0;
}


// ---------------------------------------------------------------------------------------------
// HOST helper: the weight image pa_conv3x3_wino4 expects.  U = G g G^T of Winograd F(4x4,3x3) in float64 arithmetic
// (BatchNorm scale folded per output channel in float32 first, as weights.py does), packed as one contiguous 36-KB
// slab per (32-cout slice, 8-cin stage): [cout/32][cin/8][row = 32 xi + n][8], xi = 6a + b, n = cout % 32.
// Same result as weights.winograd4_pack(winograd4_weights(w * scale)).
// ---------------------------------------------------------------------------------------------
extern "C" int pa_winograd4_pack_host(const float* conv_weight, const float* bn_scale, int cout, int cin,
                                      float* U_slabs) {
  if (cout <= 0 || cin <= 0 || cout % 32 != 0 || cin % 8 != 0 || !conv_weight || !U_slabs) {
    pa::set_error("pa_winograd4_pack_host: cout %% 32 == 0 and cin %% 8 == 0 required (got %d, %d)", cout, cin);
    return 3;
  }
  static const double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1.0}};
  const int stages = cin / 8;
  for (int o = 0; o < cout; ++o)
    for (int i = 0; i < cin; ++i) {
      double g[3][3];
      for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) {
          const float folded = bn_scale ? conv_weight[((size_t)o * cin + i) * 9 + p * 3 + q] * bn_scale[o]
                                        : conv_weight[((size_t)o * cin + i) * 9 + p * 3 + q];
          g[p][q] = (double)folded;
        }
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
          // the same contraction order as torch.einsum("ap,oipq,bq->aboi") is NOT needed bit for bit by the kernel,
          // but the Python pack is the test's reference: sum over (p, q) in row-major order of exact products
          double u = 0.0;
          for (int p = 0; p < 3; ++p)
            for (int q = 0; q < 3; ++q) u += G[a][p] * g[p][q] * G[b][q];
          const size_t slab = ((size_t)(o / 32) * stages + i / 8) * 36 * 32 * 8;
          // rows with bit 3 of (o % 32) set hold their two channel quads swapped (LDS bank swizzle of the kernel)
          const int pos = ((o % 32) >> 3) & 1 ? (i % 8) ^ 4 : i % 8;
          U_slabs[slab + (size_t)(32 * (6 * a + b) + (o % 32)) * 8 + pos] = (float)u;
        }
    }
  return 0;
}
