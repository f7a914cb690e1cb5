// Error reporting and version of the hyperpri_amd C ABI (see include/hyperpri_hip.h).
#include "common.h"
#include <string.h>

static thread_local char g_err[256] = "";

int hpri_set_error(int code, const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
  return code;
}

extern "C" const char* hpri_last_error(void) { return g_err; }
extern "C" int hpri_version(void) { return 100; }  // 0.1.0

// ---- launch-plan options (process-wide; defaults can also come from the environment) -----------------------------
//   conv_nbx_min          output-channel blocks from which conv kernels use the XCD-aware 1-D grid      (HPRI_NBX_MIN, 9)
//   wgrad_xcd_min_tiles   (C, N) tiles from which the weight-gradient kernels use the XCD-aware grid;
//                         0 = never                                                          (HPRI_WGRAD_XCD_MIN, 128)
//   wgrad_xcd_min_strips  ... and only with at least this many 64-pixel strips                (HPRI_WGRAD_XCD_STRIPS, 2048)
//   bf16v3_tile_width     conv_bf16v3 tile shapes: 0 = by padding cost (8 x 32 tiles first, the columns left over in 16 x 16 tiles and at
//                         most one column of 32 x 8 tiles), 1 = without the 32 x 8 column (round 4), 16 = 16 x 16 tiles only    (HPRI_V3_TILE_WIDTH, 0)
//   bn_wide_cq            BatchNorm / reduction kernels: 1 = tensors wider than 1024 channels take whole 1024-channel runs of one
//                         pixel per workgroup, 0 = 256-channel columns of four pixels                         (HPRI_BN_WIDE_CQ, 1)
//   wgrad_cu_reserve      compute units the fp32 Winograd weight gradient leaves free: its workgroups hold a whole CU each (104 KB of
//                         LDS) and its pixel splits are planned so that the grid is an exact multiple of the CUs -- with ONE CU held
//                         by another kernel (a collective's channel) the last workgroup starts when the first retires: 2 x the
//                         launch (profiles/r05_hog_kernels_fp32.json).  n > 0 plans for 256 - n CUs (a different split = another,
//                         equally deterministic summation order); 0 = all of them                          (HPRI_WGRAD_CU_RESERVE, 0)
#include <stdlib.h>
#include <atomic>
#include <mutex>
// Re-entrancy (include/hyperpri_hip.h: launchers may be called from any thread -- forward on the main thread, backward on
// autograd's worker threads): the options are atomics initialised exactly once, from the environment, under std::call_once;
// hpri_set_option stores with release order and the launchers' reads are relaxed loads of an int (a plan option changes
// block order only, never results).
static std::atomic<int> g_opt[6];
static std::once_flag g_opt_once;
static const char* const g_opt_name[6] = {"conv_nbx_min", "wgrad_xcd_min_tiles", "wgrad_xcd_min_strips", "bf16v3_tile_width", "bn_wide_cq", "wgrad_cu_reserve"};
static const char* const g_opt_env[6] = {"HPRI_NBX_MIN", "HPRI_WGRAD_XCD_MIN", "HPRI_WGRAD_XCD_STRIPS", "HPRI_V3_TILE_WIDTH", "HPRI_BN_WIDE_CQ", "HPRI_WGRAD_CU_RESERVE"};
static const int g_opt_default[6] = {9, 128, 2048, 0, 1, 0};

static void opt_init() {
  for (int i = 0; i < 6; ++i) {
    const char* e = getenv(g_opt_env[i]);
    int v = e ? atoi(e) : g_opt_default[i];
    if (v < 0) v = g_opt_default[i];
    g_opt[i].store(v, std::memory_order_relaxed);
  }
}

int hpri_option(int idx) {
  std::call_once(g_opt_once, opt_init);
  return g_opt[idx].load(std::memory_order_relaxed);
}

extern "C" int hpri_set_option(const char* name, int value) {
  std::call_once(g_opt_once, opt_init);
  for (int i = 0; i < 6; ++i)
    if (name && strcmp(name, g_opt_name[i]) == 0) {
      if (value < 0) return hpri_set_error(HPRI_ERR_ARG, "set_option: value must be >= 0");
      g_opt[i].store(value, std::memory_order_release);
      return HPRI_OK;
    }
  return hpri_set_error(HPRI_ERR_ARG, "set_option: unknown option");
}

// Compute units of the device the calling thread has current (hipDeviceProp_t.multiProcessorCount), cached per device:
// launch plans and the two-workgroups-per-CU stagger of the Winograd kernel size themselves by it instead of assuming 256.
static std::atomic<int> g_cus[16];
int hpri_cu_count() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  int n = g_cus[dev].load(std::memory_order_relaxed);
  if (n > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  g_cus[dev].store(n, std::memory_order_relaxed);
  return n;
}

#ifdef HPRI_DIAG_KERNELS   // measured neutral (round 3): diagnostics build only
// A stream of the LOWEST priority the device offers, for work that should only fill what the caller's stream leaves free
// (the engine's weight-gradient stream: its 256-workgroup launches otherwise hold every CU while the 2-64-workgroup finalize
// kernels on the critical path wait for a slot).  *stream receives a hipStream_t the caller owns (hpri_stream_destroy).
extern "C" int hpri_stream_create_low_priority(void** stream, int* priority) {
  HPRI_REQUIRE(stream != nullptr, "stream_create_low_priority: null pointer");
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return hpri_set_error(HPRI_ERR_LAUNCH, "stream priority range query failed");
  hipStream_t s = nullptr;
  if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) != hipSuccess) return hpri_set_error(HPRI_ERR_LAUNCH, "stream creation failed");
  *stream = reinterpret_cast<void*>(s);
  if (priority != nullptr) *priority = least;
  return HPRI_OK;
}
extern "C" int hpri_stream_destroy(void* stream) {
  if (stream != nullptr && hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)) != hipSuccess)
    return hpri_set_error(HPRI_ERR_LAUNCH, "stream destruction failed");
  return HPRI_OK;
}

#endif   // HPRI_DIAG_KERNELS

// ---- item queues of the persistent kernels (common.h) ------------------------------------------------------------------------------
// stream -> caller-owned counter buffer + the parity of the next launch.  A small table under a mutex: launchers take their half
// right before the launch (a handful of entries; the engine registers one buffer per stream it launches on).
static std::mutex g_q_mutex;
static struct { hipStream_t stream; unsigned* q; unsigned seq; bool used; } g_q[32];

HpriQueueHalves hpri_item_queue_take(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_q_mutex);
  for (auto& e : g_q)
    if (e.used && e.stream == stream) {
      const unsigned h = e.seq++ & 1u;
      return HpriQueueHalves{e.q + h * HPRI_Q_HALF, e.q + (h ^ 1u) * HPRI_Q_HALF};
    }
  return HpriQueueHalves{nullptr, nullptr};
}

extern "C" int hpri_item_queue_bytes(void) { return HPRI_Q_WORDS * 4; }

extern "C" int hpri_set_item_queue(void* queue, size_t bytes, hipStream_t stream) {
  HPRI_REQUIRE(queue == nullptr || (bytes >= (size_t)HPRI_Q_WORDS * 4 && ((uintptr_t)queue & 255) == 0),
               "set_item_queue: the queue must hold hpri_item_queue_bytes() zeroed bytes, 256-byte aligned");
  std::lock_guard<std::mutex> lock(g_q_mutex);
  for (auto& e : g_q)
    if (e.used && e.stream == stream) {
      if (queue == nullptr) e.used = false; else { e.q = reinterpret_cast<unsigned*>(queue); e.seq = 0; }
      return HPRI_OK;
    }
  if (queue == nullptr) return HPRI_OK;
  for (auto& e : g_q)
    if (!e.used) { e.stream = stream; e.q = reinterpret_cast<unsigned*>(queue); e.seq = 0; e.used = true; return HPRI_OK; }
  return hpri_set_error(HPRI_ERR_ARG, "set_item_queue: more than 32 streams hold a queue");
}

// ---- loss scale of the half-precision library (common.h: h16_t) ------------------------------------------------------------------
// The gradient that the fused head forms from the logits (hpri_outconv_bwd_bce / _x16) is multiplied by this factor -- set by the
// calling thread right before the launch (thread-local: backward runs on autograd's worker threads).  1 unless set.
static thread_local float g_loss_scale = 1.f;
float hpri_loss_scale() { return g_loss_scale; }
extern "C" int hpri_set_loss_scale(float scale) {
  HPRI_REQUIRE(scale > 0.f, "set_loss_scale: the scale must be positive");
  g_loss_scale = scale;
  return HPRI_OK;
}

extern "C" int hpri_get_option(const char* name) {
  for (int i = 0; i < 6; ++i)
    if (name && strcmp(name, g_opt_name[i]) == 0) return hpri_option(i);
  return hpri_set_error(HPRI_ERR_ARG, "get_option: unknown option");
}
