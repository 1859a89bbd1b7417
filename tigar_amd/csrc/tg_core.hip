// Runtime, vectors, CSR host<->device transfer, device-wide scan / reductions.
#include "tg_common.h"
#include <stdarg.h>

tg_ctx_t g_tg;
static char g_err[1024] = "";

void tg_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *tg_last_error(void) { return g_err; }

extern "C" int tg_init(int device) {
  if (g_tg.ready && g_tg.device == device) return 0;
  int count = 0;
  TG_CHECK_HIP(hipGetDeviceCount(&count));
  TG_REQUIRE(count > 0, "no HIP device visible");
  TG_REQUIRE(device >= 0 && device < count, "device %d out of range (%d visible)", device, count);
  TG_CHECK_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  TG_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  g_tg.device = device;
  g_tg.num_cu = prop.multiProcessorCount;
  TG_CHECK_HIP(hipStreamCreateWithFlags(&g_tg.stream, hipStreamNonBlocking));
  for (int i = 0; i < 8; i++) {
    TG_CHECK_HIP(hipEventCreate(&g_tg.ev0[i]));
    TG_CHECK_HIP(hipEventCreate(&g_tg.ev1[i]));
  }
  TG_CHECK_HIP(hipEventCreate(&g_tg.pev0));
  TG_CHECK_HIP(hipEventCreate(&g_tg.pev1));
  TG_CHECK_HIP(hipMalloc((void **)&g_tg.scratch, TG_SCRATCH_DOUBLES * sizeof(double)));
  TG_CHECK_HIP(hipHostMalloc((void **)&g_tg.host_pinned, 64 * sizeof(double), hipHostMallocDefault));
  TG_CHECK_HIP(hipEventCreateWithFlags(&g_tg.xev, hipEventDisableTiming));
  g_tg.streams[0] = g_tg.stream;
  g_tg.scratches[0] = g_tg.scratch;
  g_tg.cur_stream = 0;
  g_tg.multi = false;
  g_tg.ready = true;
  return 0;
}

// ---- second stream
extern "C" int tg_stream_set(int i) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(i == 0 || i == 1, "tg_stream_set: stream %d (0 or 1)", i);
  if (i == 1 && !g_tg.streams[1]) {
    // lowest priority: what runs here is meant to fill the gaps the work on stream 0 leaves (with equal
    // priorities the producer's big grid starves the many small, latency-bound kernels of a PtAP stage:
    // k_box_reach 0.34 -> 2.2 ms, no net gain)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    TG_CHECK_HIP(hipStreamCreateWithPriority(&g_tg.streams[1], hipStreamNonBlocking, prio_lo));
    TG_CHECK_HIP(hipMalloc((void **)&g_tg.scratches[1], TG_SCRATCH_DOUBLES * sizeof(double)));
  }
  if (i == 1 && !g_tg.multi) {
    // blocks freed so far carry no events: make them safe for both streams once
    TG_CHECK_HIP(hipStreamSynchronize(g_tg.streams[0]));
    g_tg.multi = true;
  }
  g_tg.cur_stream = i;
  g_tg.stream = g_tg.streams[i];
  g_tg.scratch = g_tg.scratches[i];
  return 0;
}

// stream `waiter` waits for everything enqueued so far on stream `waited` (no host synchronisation)
extern "C" int tg_stream_wait(int waiter, int waited) {
  TG_REQUIRE_INIT();
  TG_REQUIRE((waiter == 0 || waiter == 1) && (waited == 0 || waited == 1), "tg_stream_wait: streams are 0 and 1");
  if (waiter == waited || !g_tg.streams[waiter] || !g_tg.streams[waited]) return 0;
  TG_CHECK_HIP(hipEventRecord(g_tg.xev, g_tg.streams[waited]));
  TG_CHECK_HIP(hipStreamWaitEvent(g_tg.streams[waiter], g_tg.xev, 0));
  return 0;
}

// ---- small host tables to the device in stream order, without waiting: the bytes are parked in a ring of pinned slots,
// so the caller's array may go out of scope at once and the copy runs when the stream gets there.  A slot is reused only
// after the copy that read it has completed (its event).  Larger tables take the waiting path.
#define TG_STAGE_SLOTS 64
#define TG_STAGE_BYTES (64 * 1024)
static char *g_stage_buf = nullptr;
static hipEvent_t g_stage_ev[TG_STAGE_SLOTS];
static bool g_stage_used[TG_STAGE_SLOTS];
static int g_stage_next = 0;

int tg_h2d_staged(void *dst, const void *src, size_t bytes) {
  if (bytes == 0) return 0;
  if (bytes > TG_STAGE_BYTES) {
    TG_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g_tg.stream));
    TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
    return 0;
  }
  if (!g_stage_buf) {
    TG_CHECK_HIP(hipHostMalloc((void **)&g_stage_buf, (size_t)TG_STAGE_SLOTS * TG_STAGE_BYTES, hipHostMallocDefault));
    for (int i = 0; i < TG_STAGE_SLOTS; i++) {
      TG_CHECK_HIP(hipEventCreateWithFlags(&g_stage_ev[i], hipEventDisableTiming));
      g_stage_used[i] = false;
    }
  }
  const int slot = g_stage_next;
  g_stage_next = (g_stage_next + 1) % TG_STAGE_SLOTS;
  if (g_stage_used[slot]) TG_CHECK_HIP(hipEventSynchronize(g_stage_ev[slot]));
  char *stage = g_stage_buf + (size_t)slot * TG_STAGE_BYTES;
  memcpy(stage, src, bytes);
  TG_CHECK_HIP(hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, g_tg.stream));
  TG_CHECK_HIP(hipEventRecord(g_stage_ev[slot], g_tg.stream));
  g_stage_used[slot] = true;
  return 0;
}

static void tg_stage_release(void) {
  if (!g_stage_buf) return;
  for (int i = 0; i < TG_STAGE_SLOTS; i++) hipEventDestroy(g_stage_ev[i]);
  hipHostFree(g_stage_buf);
  g_stage_buf = nullptr;
  g_stage_next = 0;
}

extern "C" int tg_shutdown(void) {
  if (!g_tg.ready) return 0;
  for (int i = 0; i < 2; i++)
    if (g_tg.streams[i]) hipStreamSynchronize(g_tg.streams[i]);
  tg_sell_cache_clear();
  tg_kron_cache_clear();
  tg_asm_cache_clear();
  tg_stage_release();
  tg_pool_trim();
  hipFree(g_tg.scratches[0]);
  if (g_tg.scratches[1]) hipFree(g_tg.scratches[1]);
  if (g_tg.streams[1]) hipStreamDestroy(g_tg.streams[1]);
  if (g_tg.xev) hipEventDestroy(g_tg.xev);
  g_tg.stream = g_tg.streams[0];
  hipHostFree(g_tg.host_pinned);
  for (int i = 0; i < 8; i++) {
    hipEventDestroy(g_tg.ev0[i]);
    hipEventDestroy(g_tg.ev1[i]);
  }
  hipStreamDestroy(g_tg.stream);
  g_tg = tg_ctx_t();
  return 0;
}

extern "C" int tg_sync(void) {
  TG_REQUIRE_INIT();
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  return 0;
}

extern "C" int tg_device_info(char *name, int name_len, int *num_cu, int64_t *hbm_bytes) {
  TG_REQUIRE_INIT();
  hipDeviceProp_t prop;
  TG_CHECK_HIP(hipGetDeviceProperties(&prop, g_tg.device));
  if (name && name_len > 0) {
    snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (num_cu) *num_cu = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return 0;
}

extern "C" int tg_mem_info(int64_t *free_bytes, int64_t *total_bytes) {
  TG_REQUIRE_INIT();
  size_t f = 0, t = 0;
  TG_CHECK_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return 0;
}

extern "C" int tg_prof_reset(void) {
  for (int i = 0; i < TG_PROF_NSLOTS; i++) {
    g_tg.prof_ms[i] = 0.0;
    g_tg.prof_n[i] = 0;
  }
  return 0;
}

extern "C" int tg_prof_get(int slot, double *total_ms, int64_t *count) {
  TG_REQUIRE(slot >= 0 && slot < TG_PROF_NSLOTS, "profile slot out of range");
  if (total_ms) *total_ms = g_tg.prof_ms[slot];
  if (count) *count = g_tg.prof_n[slot];
  return 0;
}

extern "C" int tg_timer_start(int slot) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(slot >= 0 && slot < 8, "timer slot out of range");
  TG_CHECK_HIP(hipEventRecord(g_tg.ev0[slot], g_tg.stream));
  return 0;
}

extern "C" int tg_timer_stop(int slot, double *ms) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(slot >= 0 && slot < 8, "timer slot out of range");
  TG_CHECK_HIP(hipEventRecord(g_tg.ev1[slot], g_tg.stream));
  TG_CHECK_HIP(hipEventSynchronize(g_tg.ev1[slot]));
  float f = 0.f;
  TG_CHECK_HIP(hipEventElapsedTime(&f, g_tg.ev0[slot], g_tg.ev1[slot]));
  if (ms) *ms = (double)f;
  return 0;
}

// ------------------------------------------------------------------------------ allocator
// Caching device allocator.  hipMalloc / hipFree of multi-GB buffers cost up to ~0.1 ms per MB on
// some hosts and hipFree synchronises the device; the hot path allocates its outputs (M, M^T,
// K, temporaries) afresh on every call, so freed blocks are kept in a size-keyed pool and
// handed out again.  All work runs on one stream, so reuse is stream-ordered and needs no sync.
#include <algorithm>
#include <map>
#include <chrono>
#include <unordered_map>
static std::multimap<size_t, void *> g_pool_free;
static std::unordered_map<void *, size_t> g_pool_size;
static size_t g_pool_bytes = 0;
// Two-stream mode.  A block belongs to the stream that was current when it was allocated; objects
// cross streams only by the caller's hand-over (tg_stream_wait, then use AND release on the consuming
// stream).  A freed block records an event on the stream it is freed on and, if that is not the stream
// it was allocated on, on that one as well; whoever takes it out of the pool on stream c waits for the
// recorded event of the other stream (its own stream is ordered anyway) -- and prefers blocks that need
// no such wait, so that the temporaries of one stream do not tie it to the kernels of the other.
struct tg_block_events {
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool rec[2] = {false, false};
};
static std::unordered_map<void *, tg_block_events> g_pool_ev;
static std::unordered_map<void *, int> g_live_sid;     // stream of allocation (two-stream mode only)

static void tg_pool_sync_all() {
  if (!g_tg.ready) return;
  for (int i = 0; i < 2; i++)
    if (g_tg.streams[i]) hipStreamSynchronize(g_tg.streams[i]);
}

static void tg_pool_drop_events(void *p) {
  g_live_sid.erase(p);
  auto it = g_pool_ev.find(p);
  if (it == g_pool_ev.end()) return;
  for (int i = 0; i < 2; i++)
    if (it->second.ev[i]) hipEventDestroy(it->second.ev[i]);
  g_pool_ev.erase(it);
}

static size_t tg_pool_limit() {
  static size_t lim = 0;
  if (!lim) {
    // default: three quarters of the device memory.  hipMalloc costs 30 ms/GB and up on this stack, so
    // every block that is needed again next step should stay cached: at cfg3 the blocks cached at the
    // end of a step (111 GB) plus K (75 GB, freed when the next step starts) need 186 GB -- with the
    // former 160 GB limit K's arrays or ~25 GB of smaller blocks were given back and re-allocated
    // EVERY step (0.7-2 s).  A failing hipMalloc trims the pool and retries, so the limit is not a
    // safety margin.  Round 6: nine tenths -- the streamed assembly on a mapped patch cycles through 237 GB (K, its
    // half-storage copy, the row blocks of A, the control functions); with three quarters 21 GB of the smallest blocks
    // were given back at the start of every step and one of the hipMallocs that replaced them took 0.6-0.7 s on the
    // nearly full device.
    const char *s = getenv("TIGAR_POOL_GB");
    double gb = 160.0;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot > 0) gb = 0.9 * (double)tot / (double)((size_t)1 << 30);
    lim = (size_t)((s ? atof(s) : gb) * (double)((size_t)1 << 30));
  }
  return lim;
}

extern "C" int tg_pool_stats(int64_t *pooled_bytes, int64_t *pooled_blocks, int64_t *live_blocks) {
  if (getenv("TIGAR_TRACE") && atoi(getenv("TIGAR_TRACE")) >= 3) {       // (what sits idle, largest first)
    fprintf(stderr, "[trace] idle blocks (MB):");
    int shown = 0;
    for (auto it = g_pool_free.rbegin(); it != g_pool_free.rend() && shown < 48; ++it, ++shown) fprintf(stderr, " %.0f", it->first / 1048576.0);
    fprintf(stderr, "\n");
  }
  if (pooled_bytes) *pooled_bytes = (int64_t)g_pool_bytes;
  if (pooled_blocks) *pooled_blocks = (int64_t)g_pool_free.size();
  if (live_blocks) *live_blocks = (int64_t)g_pool_size.size() - (int64_t)g_pool_free.size();
  return 0;
}

extern "C" int tg_pool_trim(void) {
  tg_pool_sync_all();
  for (auto &kv : g_pool_free) {
    g_pool_size.erase(kv.second);
    tg_pool_drop_events(kv.second);
    hipFree(kv.second);
  }
  g_pool_free.clear();
  g_pool_bytes = 0;
  return 0;
}

// TIGAR_POOL_POISON=1 (debugging): every block handed out -- fresh or recycled -- is filled with 0x7f bytes first (column
// indices of 2.1e9, doubles of 1.4e306), so that a kernel that reads what it has not written faults or shows NaN / Inf
// instead of getting away with the zeros of fresh memory.
static int tg_dmalloc_bytes_raw(void **p, size_t bytes);
int tg_dmalloc_bytes(void **p, size_t bytes) {
  static const int poison = getenv("TIGAR_POOL_POISON") ? atoi(getenv("TIGAR_POOL_POISON")) : 0;
  const int rc = tg_dmalloc_bytes_raw(p, bytes);
  if (!rc && poison && *p && g_tg.ready) {
    auto it = g_pool_size.find(*p);
    const size_t n = it != g_pool_size.end() ? it->second : bytes;
    if (hipMemsetAsync(*p, 0x7f, n, g_tg.stream) != hipSuccess) (void)hipGetLastError();
  }
  return rc;
}
static int tg_dmalloc_bytes_raw(void **p, size_t bytes) {
  *p = nullptr;
  if (bytes == 0) bytes = 1;
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes >= ((size_t)1 << 20)) {
    // size classes, eight per octave (<= 12.5 % padding): the sub-slabs of a streamed assembly ask for
    // slightly different sizes (boundary vs interior slabs, the last shorter slab); rounded to a class,
    // a later, slightly LARGER request reuses the block of an earlier one instead of going to
    // hipMalloc (26 slow allocations per run at 14 planes per sub-slab before, none after)
    int lg = 63 - __builtin_clzll((unsigned long long)bytes);
    // (32 classes per octave from 8 GiB on: 12.5 % of a 50 GB array is too much to give away)
    const size_t g = (size_t)1 << (lg - (bytes >= ((size_t)8 << 30) ? 5 : 3));
    bytes = (bytes + g - 1) & ~(g - 1);
  }
  auto it = g_pool_free.lower_bound(bytes);
  // accept a cached block that is at most 12.5 % (or 32 MiB) larger than requested
  const size_t fit = bytes + std::max<size_t>(bytes / 8, (size_t)32 << 20);
  if (it != g_pool_free.end() && it->first <= fit) {
    if (g_tg.multi) {
      // prefer a block that carries no event of the other stream
      const int other = 1 - g_tg.cur_stream;
      auto pick = it;
      int scanned = 0;
      for (auto c = it; c != g_pool_free.end() && c->first <= fit && scanned < 32; ++c, ++scanned) {
        auto ev = g_pool_ev.find(c->second);
        if (ev == g_pool_ev.end() || !ev->second.rec[other]) {
          pick = c;
          break;
        }
      }
      it = pick;
      auto ev = g_pool_ev.find(it->second);
      const bool tied = ev != g_pool_ev.end() && ev->second.rec[other] && ev->second.ev[other];
      if (tied && it->first < ((size_t)256 << 20)) {
        // a small block that would tie this stream to the kernels in flight on the other one (the
        // producer releases its tables right after launching its fill kernel): a fresh block is cheaper
        it = g_pool_free.end();
      } else {
        if (tied) TG_CHECK_HIP(hipStreamWaitEvent(g_tg.stream, ev->second.ev[other], 0));
        g_live_sid[it->second] = g_tg.cur_stream;
      }
    }
    if (it != g_pool_free.end()) {
      *p = it->second;
      g_pool_bytes -= it->first;
      g_pool_free.erase(it);
      return 0;
    }
  }
  const bool trace = getenv("TIGAR_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  // (a request the device cannot serve is not sent to the driver at all: a hipMalloc that FAILS takes its time -- measured below)
  hipError_t e = hipErrorOutOfMemory;
  {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr >= bytes + ((size_t)64 << 20)) e = hipMalloc(p, bytes);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (trace) {
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      fprintf(stderr, "[trace] hipMalloc(%.1f MB) not served with %.1f GB pooled in %zu blocks (%.1f ms) -> trim\n", bytes / 1048576.0,
              g_pool_bytes / 1073741824.0, g_pool_free.size(), ms);
    }
    // Give back just enough of the idle blocks (smallest first; hipFree and the hipMalloc that replaces a block later both cost
    // by the byte: 9 and 30 ms per GB here -- emptying a pool of 165 GB took 1.5-2 s of every step of a streamed assembly whose
    // first allocation of a step, 3.5 GB of control functions, found its size class evicted); everything only if that fails.
    {
      tg_pool_sync_all();
      size_t freed = 0;
      const size_t want = bytes + bytes / 4 + ((size_t)256 << 20);
      while (freed < want && !g_pool_free.empty()) {
        auto sm = g_pool_free.begin();
        freed += sm->first;
        g_pool_bytes -= sm->first;
        g_pool_size.erase(sm->second);
        tg_pool_drop_events(sm->second);
        hipFree(sm->second);
        g_pool_free.erase(sm);
      }
      if (trace) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "[trace]   gave back %.1f MB, %zu blocks left (%.1f ms)\n", freed / 1048576.0, g_pool_free.size(), ms);
      }
      e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      tg_pool_trim();
      e = hipMalloc(p, bytes);
    }
  }
  if (trace) {
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms > 20.0) fprintf(stderr, "[trace] slow hipMalloc(%.1f MB): %.1f ms\n", bytes / 1048576.0, ms);
    if (atoi(getenv("TIGAR_TRACE")) >= 2 && bytes >= ((size_t)256 << 20))
      fprintf(stderr, "[trace] hipMalloc %.0f MB (pool %.1f GB in %zu blocks)\n", bytes / 1048576.0,
              g_pool_bytes / 1073741824.0, g_pool_free.size());
  }
  if (e != hipSuccess) {
    *p = nullptr;
    tg_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    return 1;
  }
  g_pool_size[*p] = bytes;
  if (g_tg.multi) g_live_sid[*p] = g_tg.cur_stream;
  return 0;
}

// Takes the LARGEST cached block of at least min_bytes out of the pool (nothing is allocated): for
// consumers that can live in several pieces of any size (the sliced copy of tg_sell.hip) and would
// otherwise ask the driver for tens of GB while as much sits idle in the pool.  Returns 1 if there is none.
int tg_pool_take_largest(size_t min_bytes, void **p, size_t *bytes) {
  *p = nullptr;
  *bytes = 0;
  if (g_pool_free.empty()) return 1;
  auto it = std::prev(g_pool_free.end());
  if (it->first < min_bytes) return 1;
  if (g_tg.multi) {
    auto ev = g_pool_ev.find(it->second);
    const int other = 1 - g_tg.cur_stream;
    if (ev != g_pool_ev.end() && ev->second.rec[other] && ev->second.ev[other])
      TG_CHECK_HIP(hipStreamWaitEvent(g_tg.stream, ev->second.ev[other], 0));
    g_live_sid[it->second] = g_tg.cur_stream;
  }
  *p = it->second;
  *bytes = it->first;
  g_pool_bytes -= it->first;
  g_pool_free.erase(it);
  return 0;
}

void tg_dfree(void *p) {
  if (!p) return;
  auto it = g_pool_size.find(p);
  if (it == g_pool_size.end()) {  // not ours (should not happen)
    hipFree(p);
    return;
  }
  const size_t bytes = it->second;
  if (g_pool_bytes + bytes > tg_pool_limit()) {
    // pool full: hipMalloc costs ~30 ms/GB on this stack (1.5 s for the 50 GB value array of K), so
    // the LARGE blocks are the ones worth keeping -- evict cached blocks smaller than the incoming one,
    // smallest first, before giving the incoming block back to the driver
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    while (g_pool_bytes + bytes > tg_pool_limit() && !g_pool_free.empty() && g_pool_free.begin()->first < bytes) {
      auto sm = g_pool_free.begin();
      if (!synced && g_tg.ready) {
        tg_pool_sync_all();
        synced = true;
      }
      g_pool_bytes -= sm->first;
      g_pool_size.erase(sm->second);
      tg_pool_drop_events(sm->second);
      hipFree(sm->second);
      g_pool_free.erase(sm);
    }
    if (g_pool_bytes + bytes > tg_pool_limit()) {
      if (!synced && g_tg.ready) tg_pool_sync_all();
      g_pool_size.erase(it);
      tg_pool_drop_events(p);
      hipFree(p);
      if (getenv("TIGAR_TRACE")) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 20.0) fprintf(stderr, "[trace] slow hipFree(%.1f MB) (pool full): %.1f ms\n", bytes / 1048576.0, ms);
      }
      return;
    }
  }
  if (g_tg.multi && g_tg.ready) {
    tg_block_events &be = g_pool_ev[p];
    int asid = g_tg.cur_stream;
    auto ls = g_live_sid.find(p);
    if (ls != g_live_sid.end()) {
      asid = ls->second;
      g_live_sid.erase(ls);
    } else {
      asid = -1;   // allocated before the second stream existed: stream 0, already synchronised
    }
    for (int i = 0; i < 2; i++) {
      be.rec[i] = false;
      if (!g_tg.streams[i] || (i != g_tg.cur_stream && i != asid)) continue;
      if (!be.ev[i] && hipEventCreateWithFlags(&be.ev[i], hipEventDisableTiming) != hipSuccess) be.ev[i] = nullptr;
      if (be.ev[i] && hipEventRecord(be.ev[i], g_tg.streams[i]) == hipSuccess) {
        be.rec[i] = true;
      } else {
        // cannot order the reuse with an event: a host synchronisation of that stream does it
        hipStreamSynchronize(g_tg.streams[i]);
      }
    }
  }
  g_pool_free.emplace(bytes, p);
  g_pool_bytes += bytes;
}

// ------------------------------------------------------------------------------ vectors
extern "C" int tg_vec_create(int64_t n, tg_vec_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(n >= 0 && out, "bad arguments");
  tg_vec_s *v = new tg_vec_s();
  v->n = n;
  if (tg_dmalloc(&v->d, n)) {
    delete v;
    return 1;
  }
  TG_CHECK_HIP(hipMemsetAsync(v->d, 0, (size_t)(n > 0 ? n : 1) * sizeof(double), g_tg.stream));
  *out = v;
  return 0;
}

extern "C" int tg_vec_create_uninit(int64_t n, tg_vec_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(n >= 0 && out, "bad arguments");
  tg_vec_s *v = new tg_vec_s();
  v->n = n;
  if (tg_dmalloc(&v->d, n)) {
    delete v;
    return 1;
  }
  *out = v;
  return 0;
}

extern "C" int tg_vec_destroy(tg_vec_t v) {
  if (!v) return 0;
  // (the block goes back to the pool and its re-use is ordered behind the work in flight; with two
  // streams a host synchronisation here would serialise them -- a temporary released on the producer
  // stream would wait for the producer's own kernel)
  if (g_tg.ready && !g_tg.multi) hipStreamSynchronize(g_tg.stream);
  tg_dfree(v->d);
  delete v;
  return 0;
}

extern "C" int tg_vec_size(tg_vec_t v, int64_t *n) {
  TG_REQUIRE(v && n, "bad arguments");
  *n = v->n;
  return 0;
}

extern "C" int tg_vec_upload(tg_vec_t v, const double *host, int64_t n) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(v && host && n == v->n, "size mismatch in tg_vec_upload (%lld vs %lld)", (long long)n,
             (long long)(v ? v->n : -1));
  TG_CHECK_HIP(hipMemcpyAsync(v->d, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  return 0;
}

extern "C" int tg_vec_download(tg_vec_t v, double *host, int64_t n) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(v && host && n == v->n, "size mismatch in tg_vec_download");
  TG_CHECK_HIP(hipMemcpyAsync(host, v->d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  return 0;
}

__global__ void k_fill(double *x, int64_t n, double a) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) x[i] = a;
}


extern "C" int tg_vec_fill(tg_vec_t v, double a) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(v, "null vector");
  hipLaunchKernelGGL(k_fill, dim3(tg_grid_1d(v->n, 256)), dim3(256), 0, g_tg.stream, v->d, v->n, a);
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_vec_copy(tg_vec_t dst, tg_vec_t src) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(dst && src && dst->n == src->n, "size mismatch in tg_vec_copy");
  TG_CHECK_HIP(hipMemcpyAsync(dst->d, src->d, (size_t)src->n * sizeof(double), hipMemcpyDeviceToDevice,
                              g_tg.stream));
  return 0;
}

extern "C" int tg_vec_copy_range(tg_vec_t dst, int64_t dst_off, tg_vec_t src, int64_t src_off, int64_t n) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(dst && src && n >= 0 && dst_off >= 0 && src_off >= 0 && dst_off + n <= dst->n && src_off + n <= src->n,
             "range error in tg_vec_copy_range");
  if (n)
    TG_CHECK_HIP(hipMemcpyAsync(dst->d + dst_off, src->d + src_off, (size_t)n * sizeof(double),
                                hipMemcpyDeviceToDevice, g_tg.stream));
  return 0;
}

__global__ void k_axpy(double *y, double a, const double *x, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] += a * x[i];
}

extern "C" int tg_vec_axpy(tg_vec_t y, double a, tg_vec_t x) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(x && y && x->n == y->n, "size mismatch in tg_vec_axpy");
  hipLaunchKernelGGL(k_axpy, dim3(tg_grid_1d(y->n, 256)), dim3(256), 0, g_tg.stream, y->d, a, x->d, y->n);
  TG_LAUNCH_CHECK();
  return 0;
}

__global__ void k_pointwise_mult(double *w, const double *x, const double *y, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) w[i] = x[i] * y[i];
}

// w = x .* y (PETSc VecPointwiseMult; w may alias x or y)
extern "C" int tg_vec_pointwise_mult(tg_vec_t w, tg_vec_t x, tg_vec_t y) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(w && x && y && x->n == y->n && w->n == x->n, "size mismatch in tg_vec_pointwise_mult");
  hipLaunchKernelGGL(k_pointwise_mult, dim3(tg_grid_1d(w->n, 256)), dim3(256), 0, g_tg.stream, w->d, x->d, y->d, w->n);
  TG_LAUNCH_CHECK();
  return 0;
}

// ---- deterministic dot: fixed grid of partial sums, then one block folds them ---------


#define TG_DOT_BLOCKS 1024
__global__ void __launch_bounds__(256) k_dot_partial(const double *x, const double *y, int64_t n,
                                                     double *partial) {
  __shared__ double lds4[4];
  double s = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) s += x[i] * y[i];
  s = tg_block_sum256(s, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// folds `nstreams` interleaved partial streams: partial[b*nstreams + k]
__global__ void __launch_bounds__(256) k_fold_partials(const double *partial, int nb, int nstreams,
                                                       double *out) {
  __shared__ double lds4[4];
  for (int k = 0; k < nstreams; k++) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) s += partial[(int64_t)b * nstreams + k];
    s = tg_block_sum256(s, lds4);
    if (threadIdx.x == 0) out[k] = s;
  }
}

extern "C" int tg_vec_dot(tg_vec_t x, tg_vec_t y, double *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(x && y && x->n == y->n && out, "size mismatch in tg_vec_dot");
  double *partial = g_tg.scratch;
  hipLaunchKernelGGL(k_dot_partial, dim3(TG_DOT_BLOCKS), dim3(256), 0, g_tg.stream, x->d, y->d, x->n, partial);
  hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(256), 0, g_tg.stream, partial, TG_DOT_BLOCKS, 1,
                     partial + TG_DOT_BLOCKS);
  TG_LAUNCH_CHECK();
  TG_CHECK_HIP(hipMemcpyAsync(g_tg.host_pinned, partial + TG_DOT_BLOCKS, sizeof(double), hipMemcpyDeviceToHost,
                              g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  *out = g_tg.host_pinned[0];
  return 0;
}

// l1 / linf norms (dolfin GenericVector::norm("l1"|"linf") [ext]): fixed grid of partials, one folding block
__global__ void __launch_bounds__(256) k_absnorm_partial(const double *x, int64_t n, int use_max, double *partial) {
  __shared__ double lds[256];
  double s = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double a = fabs(x[i]);
    s = use_max ? ((a > s || a != a) ? a : s) : s + a;   // NaN propagates in both norms
  }
  lds[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const double b = lds[threadIdx.x + o], a = lds[threadIdx.x];
      lds[threadIdx.x] = use_max ? ((b > a || b != b) ? b : a) : a + b;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = lds[0];
}

extern "C" int tg_vec_norm(tg_vec_t x, int kind, double *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(x && out && kind >= 0 && kind <= 2, "bad arguments to tg_vec_norm (kind 0 = l1, 1 = l2, 2 = linf)");
  if (kind == 1) {
    TG_TRY(tg_vec_dot(x, x, out));
    *out = sqrt(*out);
    return 0;
  }
  double *partial = g_tg.scratch;
  hipLaunchKernelGGL(k_absnorm_partial, dim3(TG_DOT_BLOCKS), dim3(256), 0, g_tg.stream, x->d, x->n, kind == 2 ? 1 : 0,
                     partial);
  hipLaunchKernelGGL(k_absnorm_partial, dim3(1), dim3(256), 0, g_tg.stream, partial, (int64_t)TG_DOT_BLOCKS,
                     kind == 2 ? 1 : 0, partial + TG_DOT_BLOCKS);
  TG_LAUNCH_CHECK();
  TG_CHECK_HIP(hipMemcpyAsync(g_tg.host_pinned, partial + TG_DOT_BLOCKS, sizeof(double), hipMemcpyDeviceToHost,
                              g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  *out = g_tg.host_pinned[0];
  return 0;
}

__global__ void k_zero_entries(double *y, int64_t n, const int32_t *dofs, int64_t nd, int64_t g0) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nd) {
    const int64_t d = (int64_t)dofs[i] - g0;
    if (d >= 0 && d < n) y[d] = 0.0;
  }
}

extern "C" int tg_vec_zero_entries(tg_vec_t y, const int32_t *dofs, int64_t n) {
  return tg_vec_zero_entries_offset(y, dofs, n, 0);
}

extern "C" int tg_vec_zero_entries_offset(tg_vec_t y, const int32_t *dofs, int64_t n, int64_t g0) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(y, "null vector");
  if (n <= 0) return 0;
  int32_t *d = nullptr;
  TG_TRY(tg_dmalloc(&d, n));
  TG_CHECK_HIP(hipMemcpyAsync(d, dofs, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream));
  hipLaunchKernelGGL(k_zero_entries, dim3((unsigned)tg_cdiv(n, 256)), dim3(256), 0, g_tg.stream, y->d, y->n, d, n, g0);
  TG_LAUNCH_CHECK();
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  tg_dfree(d);
  return 0;
}

__global__ void k_tensor3(double *out, int d, const double *b0, const double *b1, const double *b2, int64_t n0,
                          int64_t n1, double scale, int64_t row0, int64_t nloc) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // (64-bit divisions cost ~100 instructions each and made this store stream ALU-bound: 32-bit ones where the index allows)
  const bool small = row0 + nloc < 0xffffffffll;
  for (; i < nloc; i += stride) {
    int64_t r = row0 + i;
    int64_t a, b = 0, c = 0;
    if (small) {
      const uint32_t r32 = (uint32_t)r, t32 = r32 / (uint32_t)n0;
      a = r32 - t32 * (uint32_t)n0;
      if (d > 1) {
        c = t32 / (uint32_t)n1;
        b = t32 - (uint32_t)c * (uint32_t)n1;
      }
    } else {
      a = r % n0;
      const int64_t t = r / n0;
      if (d > 1) {
        b = t % n1;
        c = t / n1;
      }
    }
    double v = b0[a];
    if (d > 1) {
      v *= b1[b];
      if (d > 2) v *= b2[c];
    }
    out[i] = scale * v;
  }
}

extern "C" int tg_vec_tensor3(tg_vec_t out, int d, const double *const *b1d, const int64_t *n, double scale,
                              int64_t row0, int64_t row1) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(out && d >= 1 && d <= 3 && row1 >= row0 && out->n == row1 - row0, "bad arguments to tg_vec_tensor3");
  double *db[3] = {nullptr, nullptr, nullptr};
  for (int k = 0; k < d; k++) {
    TG_TRY(tg_dmalloc(&db[k], n[k]));
    TG_TRY(tg_h2d_staged(db[k], b1d[k], (size_t)n[k] * sizeof(double)));
  }
  hipLaunchKernelGGL(k_tensor3, dim3(tg_grid_1d(out->n, 256)), dim3(256), 0, g_tg.stream, out->d, d, db[0], db[1],
                     db[2], n[0], d > 1 ? n[1] : 1, scale, row0, out->n);
  TG_LAUNCH_CHECK();
  for (int k = 0; k < d; k++) tg_dfree(db[k]);      // (stream order: the pool hands them out to later work only)
  return 0;
}

// ------------------------------------------------------------------------------ CSR objects
int tg_csr_alloc(int64_t nrows, int64_t ncols, int64_t nnz, tg_csr_s **out) {
  tg_csr_s *m = new tg_csr_s();
  m->nrows = nrows;
  m->ncols = ncols;
  m->nnz = nnz;
  if (tg_dmalloc(&m->rowptr, nrows + 1) || tg_dmalloc(&m->col, nnz + TG_CSR_PAD) ||
      tg_dmalloc(&m->val, nnz + TG_CSR_PAD)) {
    tg_dfree(m->rowptr);
    tg_dfree(m->col);
    tg_dfree(m->val);
    delete m;
    return 1;
  }
  *out = m;
  return 0;
}

extern "C" int tg_csr_from_host(int64_t nrows, int64_t ncols, const int64_t *rowptr, const int32_t *col,
                                const double *val, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(nrows >= 0 && ncols >= 0 && rowptr && out, "bad arguments to tg_csr_from_host");
  int64_t nnz = rowptr[nrows] - rowptr[0];
  TG_REQUIRE(rowptr[0] == 0, "rowptr[0] must be 0");
  tg_csr_s *m = nullptr;
  TG_TRY(tg_csr_alloc(nrows, ncols, nnz, &m));
  TG_CHECK_HIP(hipMemcpyAsync(m->rowptr, rowptr, (size_t)(nrows + 1) * sizeof(int64_t), hipMemcpyHostToDevice,
                              g_tg.stream));
  if (nnz > 0) {
    TG_CHECK_HIP(hipMemcpyAsync(m->col, col, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream));
    TG_CHECK_HIP(hipMemcpyAsync(m->val, val, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, g_tg.stream));
  }
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  *out = m;
  return 0;
}

extern "C" int tg_csr_dims(tg_csr_t m, int64_t *nrows, int64_t *ncols, int64_t *nnz) {
  TG_REQUIRE(m, "null matrix");
  if (nrows) *nrows = m->nrows;
  if (ncols) *ncols = m->ncols;
  if (nnz) *nnz = m->nnz;
  return 0;
}

// rowptr[r] of a device matrix (capacity estimates from row blocks already computed)
extern "C" int tg_csr_rowptr_at(tg_csr_t m, int64_t r, int64_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(m && out && r >= 0 && r <= m->nrows, "tg_csr_rowptr_at: row out of range");
  TG_REQUIRE_CANONICAL(m);
  TG_CHECK_HIP(hipMemcpyAsync(out, m->rowptr + r, sizeof(int64_t), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  return 0;
}

// rows [r0, r1) of a device matrix to the host: rowptr_out[r1-r0+1] (relative to the first entry handed out),
// then at most `cap` entries.  Two-call protocol: with col/val == nullptr only the row pointers are filled, so
// the caller can size its buffers.  Parity tests sample rows of matrices that are too large to download whole.
extern "C" int tg_csr_download_rows(tg_csr_t m, int64_t r0, int64_t r1, int64_t *rowptr_out, int32_t *col, double *val,
                                    int64_t cap) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(m && rowptr_out && r0 >= 0 && r1 >= r0 && r1 <= m->nrows, "tg_csr_download_rows: bad row range");
  TG_REQUIRE_CANONICAL(m);
  const int64_t n = r1 - r0;
  TG_CHECK_HIP(hipMemcpyAsync(rowptr_out, m->rowptr + r0, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost,
                              g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  const int64_t e0 = rowptr_out[0], cnt = rowptr_out[n] - e0;
  for (int64_t i = 0; i <= n; i++) rowptr_out[i] -= e0;
  if (!col && !val) return 0;
  TG_REQUIRE(col && val && cnt <= cap, "tg_csr_download_rows: %lld entries do not fit the buffers (%lld)",
             (long long)cnt, (long long)cap);
  if (cnt > 0) {
    TG_CHECK_HIP(hipMemcpyAsync(col, m->col + e0, (size_t)cnt * sizeof(int32_t), hipMemcpyDeviceToHost, g_tg.stream));
    TG_CHECK_HIP(hipMemcpyAsync(val, m->val + e0, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, g_tg.stream));
    TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  }
  return 0;
}

extern "C" int tg_csr_is_loose(tg_csr_t m, int *loose) {
  TG_REQUIRE(m && loose, "null argument to tg_csr_is_loose");
  *loose = m->rowcnt ? 1 : 0;
  return 0;
}

extern "C" int tg_csr_download(tg_csr_t m, int64_t *rowptr, int32_t *col, double *val) {
  if (m && m->rowcnt) {   // loose rows: hand out the canonical form
    tg_csr_s *c = nullptr;
    TG_TRY(tg_csr_compact_impl(m, &c));
    const int rc = tg_csr_download(c, rowptr, col, val);
    tg_csr_destroy(c);
    return rc;
  }
  TG_REQUIRE_INIT();
  TG_REQUIRE(m, "null matrix");
  if (rowptr)
    TG_CHECK_HIP(hipMemcpyAsync(rowptr, m->rowptr, (size_t)(m->nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost,
                                g_tg.stream));
  if (col && m->nnz)
    TG_CHECK_HIP(hipMemcpyAsync(col, m->col, (size_t)m->nnz * sizeof(int32_t), hipMemcpyDeviceToHost, g_tg.stream));
  if (val && m->nnz)
    TG_CHECK_HIP(hipMemcpyAsync(val, m->val, (size_t)m->nnz * sizeof(double), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  return 0;
}

extern "C" int tg_csr_destroy(tg_csr_t m) {
  if (!m) return 0;
  if (g_tg.ready && !g_tg.multi) hipStreamSynchronize(g_tg.stream);
  tg_dfree(m->rowptr);
  tg_dfree(m->rowptr_val);
  if (!m->view) {
    tg_dfree(m->col);
    tg_dfree(m->val);
  }
  tg_dfree(m->rowblocks);
  tg_dfree(m->rowcnt);
  tg_dfree(m->diag_cache);
  tg_sell_drop(m);
  delete m;
  return 0;
}

// ------------------------------------------------------------------------------ scan
// Exclusive scan of int64, tile = 256 threads x 8 items.
#define TG_SCAN_ITEMS 8
#define TG_SCAN_TILE (256 * TG_SCAN_ITEMS)



__global__ void __launch_bounds__(256) k_scan_tile_sums(const int64_t *d, int64_t n, int64_t *sums) {
  __shared__ int64_t lds5[4];
  const int64_t base = (int64_t)blockIdx.x * TG_SCAN_TILE + (int64_t)threadIdx.x * TG_SCAN_ITEMS;
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < TG_SCAN_ITEMS; k++)
    if (base + k < n) s += d[base + k];
  int64_t total;
  tg_block_excl_scan_i64(s, lds5, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) k_scan_tiles(int64_t *d, int64_t n, const int64_t *tile_offsets) {
  __shared__ int64_t lds5[4];
  const int64_t base = (int64_t)blockIdx.x * TG_SCAN_TILE + (int64_t)threadIdx.x * TG_SCAN_ITEMS;
  int64_t v[TG_SCAN_ITEMS];
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < TG_SCAN_ITEMS; k++) {
    v[k] = (base + k < n) ? d[base + k] : 0;
    s += v[k];
  }
  int64_t ex = tg_block_excl_scan_i64(s, lds5, nullptr);
  ex += tile_offsets ? tile_offsets[blockIdx.x] : 0;
#pragma unroll
  for (int k = 0; k < TG_SCAN_ITEMS; k++) {
    if (base + k < n) d[base + k] = ex;
    ex += v[k];
  }
}

// in-place exclusive scan; recursion depth <= 3 for n up to 8.6e9
static int tg_scan_rec(int64_t *d, int64_t n) {
  if (n <= 0) return 0;
  const int64_t ntiles = tg_cdiv(n, TG_SCAN_TILE);
  if (ntiles == 1) {
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(256), 0, g_tg.stream, d, n, (const int64_t *)nullptr);
    TG_LAUNCH_CHECK();
    return 0;
  }
  int64_t *sums = nullptr;
  TG_TRY(tg_dmalloc(&sums, ntiles));
  hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)ntiles), dim3(256), 0, g_tg.stream, d, n, sums);
  TG_LAUNCH_CHECK();
  int rc = tg_scan_rec(sums, ntiles);
  if (!rc) {
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)ntiles), dim3(256), 0, g_tg.stream, d, n, (const int64_t *)sums);
    if (hipGetLastError() != hipSuccess) rc = 1;
  }
  hipStreamSynchronize(g_tg.stream);
  tg_dfree(sums);
  return rc;
}

// d has n+1 slots: d[0..n) hold counts, d[n] is ignored on input; on output d[i] =
// exclusive prefix and d[n] = total.
int tg_exclusive_scan_i64(int64_t *d, int64_t n, int64_t *host_total) {
  TG_CHECK_HIP(hipMemsetAsync(d + n, 0, sizeof(int64_t), g_tg.stream));
  TG_TRY(tg_scan_rec(d, n + 1));
  if (host_total) {
    TG_CHECK_HIP(hipMemcpyAsync(host_total, d + n, sizeof(int64_t), hipMemcpyDeviceToHost, g_tg.stream));
    TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  }
  return 0;
}
