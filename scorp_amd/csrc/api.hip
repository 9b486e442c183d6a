// api.hip — error plumbing and version of libscorp_gs (see include/scorp_gs.h).
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "common.hpp"

namespace scorp {
namespace {
thread_local char g_error[512] = "";
}
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace scorp

// ---- kernel timing ----
namespace scorp {
bool g_prof_on = false;
uint64_t g_prof_mask = ~0ull;
namespace {
const char *kKernelNames[kKNumKernels] = {"preprocess", "count_tiles", "scan_tiles", "scatter_pairs", "sort_tiles", "blend_forward",
                                          "blend_backward", "preprocess_backward", "ssim_l1_forward", "ssim_l1_backward", "knn_dist2", "adam", "preprocess_2d", "blend_forward_2d",
                                          "blend_backward_2d", "preprocess_backward_2d", "surfel_maps_forward", "surfel_maps_backward"};
struct Pending { hipEvent_t start, stop; int id; };
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_pool;
std::vector<hipEvent_t> g_open(kKNumKernels, nullptr);
double g_ms[kKNumKernels];
uint64_t g_count[kKNumKernels];
std::mutex g_mu;
hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace
void prof_begin(int id, hipStream_t stream) {
  if (!((g_prof_mask >> id) & 1)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t e = get_event();
  (void)hipEventRecord(e, stream);
  g_open[id] = e;
}
void prof_end(int id, hipStream_t stream) {
  if (!((g_prof_mask >> id) & 1)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t e = get_event();
  (void)hipEventRecord(e, stream);
  g_pending.push_back({g_open[id], e, id});
  g_open[id] = nullptr;
}
}  // namespace scorp

extern "C" int scorp_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(scorp::g_mu);
  for (auto &p : scorp::g_pending) { scorp::g_pool.push_back(p.start); scorp::g_pool.push_back(p.stop); }
  scorp::g_pending.clear();
  for (int k = 0; k < scorp::kKNumKernels; k++) { scorp::g_ms[k] = 0; scorp::g_count[k] = 0; }
  scorp::g_prof_on = on != 0;
  return SCORP_OK;
}
extern "C" int scorp_prof_select(uint64_t kernel_mask) {
  std::lock_guard<std::mutex> lk(scorp::g_mu);
  scorp::g_prof_mask = kernel_mask;
  return SCORP_OK;
}
extern "C" int scorp_prof_num_kernels(void) { return scorp::kKNumKernels; }
extern "C" const char *scorp_prof_kernel_name(int k) {
  return (k >= 0 && k < scorp::kKNumKernels) ? scorp::kKernelNames[k] : "";
}
extern "C" int scorp_prof_collect(double *total_ms, uint64_t *launches) {
  std::lock_guard<std::mutex> lk(scorp::g_mu);
  for (auto &p : scorp::g_pending) {
    hipError_t e = hipEventSynchronize(p.stop);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, p.start, p.stop);
    if (e != hipSuccess) { scorp::set_error("event timing failed: %s", hipGetErrorString(e)); return SCORP_ERR_HIP; }
    scorp::g_ms[p.id] += ms;
    scorp::g_count[p.id] += 1;
    scorp::g_pool.push_back(p.start);
    scorp::g_pool.push_back(p.stop);
  }
  scorp::g_pending.clear();
  for (int k = 0; k < scorp::kKNumKernels; k++) {
    if (total_ms) total_ms[k] = scorp::g_ms[k];
    if (launches) launches[k] = scorp::g_count[k];
  }
  return SCORP_OK;
}

extern "C" int scorp_version(void) { return 100; /* 0.1.0 */ }

#ifndef SCORP_SOURCE_SHA
#define SCORP_SOURCE_SHA "unknown"
#endif
// sha256 (first 16 hex digits) of the kernel sources this library was built from (scorp_amd/build.py): lets bench.py
// tell whether the PMC-derived figures under profiles/ were collected on the code that is running
extern "C" const char *scorp_source_sha(void) { return SCORP_SOURCE_SHA; }
extern "C" const char *scorp_last_error(void) { return scorp::g_error; }
