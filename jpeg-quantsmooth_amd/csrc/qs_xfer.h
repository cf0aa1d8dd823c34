// qs_xfer.h -- pooled device / pinned buffers, streams, the helper-thread pool and the
// staged host<->device transfers of the job layer (qs_job.cpp, qs_shard.cpp).
//
// Every pooled device buffer and stream set remembers the HIP device it was created on
// and is only handed to a caller whose current device is that one: a host process may
// drive several GPUs (hipSetDevice per thread, or the sharded job route which walks over
// all of them from one thread).  Pinned staging memory is allocated portable, i.e. usable
// for DMA by every device.  The pools are C++17 inline variables: one instance per
// library, however many translation units include this header.
#pragma once
#include <new>
#include <string>
#include <mutex>
#include <vector>
#include <deque>
#include <atomic>
#include <thread>
#include <condition_variable>
#include <functional>
#include <memory>
#include <algorithm>

#include "qs_common.h"

extern "C" void qs_hip_release_cache(void);

namespace qsx {

struct CacheEntry { void* p; size_t n; int dev; };
inline std::mutex g_cache_mu;
inline std::vector<CacheEntry>& g_cache = *new std::vector<CacheEntry>;   // free device blocks (all devices); leaked: see PinnedBuf::pool()
inline const size_t kCacheMaxBytes = (size_t)6 << 30;   // per device

inline int current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; } return d; }
// scope guard: make `dev` current, put the caller's device back on exit
struct DeviceScope {
  int prev = -1;
  explicit DeviceScope(int dev) { prev = current_device(); if (dev != prev) (void)hipSetDevice(dev); else prev = -1; }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

// size classes of the pooled buffers: 64 KiB, then eight steps per octave (a request is rounded up by at most
// 12.5 %; whole powers of two turned a 34 MiB band into a 64 MiB block -- twice the pinning time in a cold process)
inline size_t round_size(size_t n) {
  size_t c = (size_t)64 << 10;
  if (n <= c) return c;
  while ((c << 1) < n) c <<= 1;                    // c < n <= 2c
  const size_t step = c >> 3;
  return c + (n - c + step - 1) / step * step;
}

// Test hooks (tests/test_gpu_faults.py), live only when the process was started with QS_HIP_TEST_HOOKS=1 (read
// once: a production process never calls getenv on these paths, which would race with a host application
// that calls setenv from another thread).  QS_HIP_TEST_FAIL_ALLOC=N makes the N-th device allocation request
// after the variable was set or changed fail once with hipErrorOutOfMemory, so that the error paths of the
// job layer (drain guards, pool returns, fall-backs, compute slots) can be exercised.
// QS_HIP_TEST_FAIL_PINNED=N does the same for pinned host buffers (the transfers then take their pageable
// fall-backs, or the call reports QS_HIP_ENOMEM where a pinned block is indispensable);
// QS_HIP_TEST_FAIL_FINISH=N makes the N-th Download::finish report a HIP error AFTER it has written its
// pieces to caller memory (a late failure while results are being scattered: the caller's arrays must be
// restored).
inline bool test_hooks_on() { static const bool on = getenv("QS_HIP_TEST_HOOKS") != nullptr; return on; }
struct TestFault {
  const char* name;
  std::atomic<bool> armed{false};
  std::mutex mu;
  std::string seen;
  long countdown = 0;
  explicit TestFault(const char* n) : name(n) {}
  bool fire() {
    if (!test_hooks_on()) return false;
    const char* v = getenv(name);
    if (!v) {
      if (armed.load(std::memory_order_relaxed)) { std::lock_guard<std::mutex> lk(mu); seen.clear(); armed = false; }
      return false;
    }
    std::lock_guard<std::mutex> lk(mu);
    if (seen != v) { seen = v; countdown = atol(v); armed = true; }
    return countdown > 0 && --countdown == 0;
  }
};
inline bool test_fail_alloc() { static TestFault f("QS_HIP_TEST_FAIL_ALLOC"); return f.fire(); }
inline bool test_fail_pinned() { static TestFault f("QS_HIP_TEST_FAIL_PINNED"); return f.fire(); }
inline bool test_fail_finish() { static TestFault f("QS_HIP_TEST_FAIL_FINISH"); return f.fire(); }

struct DevBuf {
  void* p = nullptr;
  size_t n = 0;
  int dev = 0;                                     // device the block lives on
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  hipError_t alloc(size_t bytes) {
    release();
    if (test_fail_alloc()) return hipErrorOutOfMemory;
    const size_t want = round_size(bytes);
    dev = current_device();                        // allocations belong to the caller's current device
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      size_t best = g_cache.size();                // the smallest free block of [want, 1.25 want] on this device
      for (size_t i = 0; i < g_cache.size(); ++i)
        if (g_cache[i].dev == dev && g_cache[i].n >= want && g_cache[i].n <= want + want / 4 &&
            (best == g_cache.size() || g_cache[i].n < g_cache[best].n)) best = i;
      if (best < g_cache.size()) {
        p = g_cache[best].p; n = g_cache[best].n; g_cache.erase(g_cache.begin() + best); return hipSuccess;
      }
    }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {                         // make room and retry once
      (void)hipGetLastError();
      qs_hip_release_cache();
      e = hipMalloc(&p, want);
    }
    if (e == hipSuccess) n = want; else p = nullptr;
    return e;
  }
  void release() {
    if (!p) return;
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      size_t held = 0;
      for (auto& c : g_cache) if (c.dev == dev) held += c.n;
      if (held + n <= kCacheMaxBytes) { g_cache.push_back({p, n, dev}); p = nullptr; n = 0; return; }
    }
    { DeviceScope on(dev); (void)hipFree(p); }
    p = nullptr; n = 0;
  }
  void take(DevBuf& o) { release(); p = o.p; n = o.n; dev = o.dev; o.p = nullptr; o.n = 0; }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// ---- host -> device upload of large pageable buffers -------------------------
// Measured on the MI355X box (tools/ubench_pcie.hip, 128 MiB): pageable
// hipMemcpy H2D 17-19 GB/s, pinned 57 GB/s, hipHostRegister 6 ms + 57 GB/s,
// memcpy into pinned memory 30 GB/s with one thread and 60-90 GB/s with 4-8;
// pageable D2H already runs at 55 GB/s.  So uploads above a few MiB go through a
// pooled pinned staging buffer: four threads copy 8 MiB chunks into it and each
// chunk's DMA is queued as soon as it is complete, overlapping the next copy.
// (Round 3, tools/cold_phases: the BLOCKING hipMemcpy of pageable memory reaches 18-44 GB/s, so a large upload
// is staged only when a pinned block of its size is already pooled -- see PinnedBuf::alloc(optional).)
struct PinnedBuf {
  void* p = nullptr;
  size_t n = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { release(); }
  // (inline function: one per library.  Leaked on purpose, like pending() and pinned_handouts() below: function-local
  //  statics are destroyed BEFORE the namespace-scope object that joins the background threads at exit (qs_job.cpp:
  //  BackgroundJoin) when they were constructed after it, and a prewarm / kick_background thread still at work would
  //  then touch a dead container.)
  static std::vector<CacheEntry>& pool() { static auto* v = new std::vector<CacheEntry>; return *v; }
  // optional: the caller has a fall-back that costs less than pinning a large block NOW (an upload can go
  // straight from pageable memory at ~44 GB/s on this platform, pinning costs ~0.18 ms per MiB): a block above
  // kOptionalMax is then only taken from the pool, and one is pinned in the background for the next call
  static constexpr size_t kOptionalMax = (size_t)16 << 20;
  bool alloc(size_t bytes, bool optional = false) {
    release();                                     // (a reused object must not leak the block it holds)
    if (test_fail_pinned()) return false;
    const size_t want = round_size(bytes);
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      auto& v = pool();
      size_t best = v.size();                      // the smallest free block of [want, 1.25 want]
      for (size_t i = 0; i < v.size(); ++i)
        if (v[i].n >= want && v[i].n <= want + want / 4 && (best == v.size() || v[i].n < v[best].n)) best = i;
      if (best < v.size()) { p = v[best].p; n = v[best].n; v.erase(v.begin() + best); return true; }
    }
    if (optional && want > kOptionalMax && upload_stage_mode() != 1) {
      if (upload_stage_mode() == 2) fill_later(want);
      return false;
    }
    if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
    n = want;
    return true;
  }
  // QS_HIP_UPLOAD_STAGE: 1 = always pin upload staging (round-2 behaviour), 0 = never above 16 MiB, unset = from
  // the pool only, filled in the background
  static int upload_stage_mode() {
    static const int m = [] { const char* v = getenv("QS_HIP_UPLOAD_STAGE"); return !v ? 2 : atoi(v) ? 1 : 0; }();
    return m;
  }
  // Blocks wanted for next time.  They are pinned by ONE background thread that starts when the job that missed
  // them has finished (kick_fills, called by the entry points on their way out): pinning takes the process's
  // mmap lock, and done during the job it slowed the job's own pageable copies down to a third.
  static std::vector<std::pair<size_t, int>>& pending() { static auto* v = new std::vector<std::pair<size_t, int>>; return *v; }
  static void fill_later(size_t want) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (pending().size() < 16) pending().push_back({want, current_device()});
  }
  static bool fills_pending() { std::lock_guard<std::mutex> lk(g_cache_mu); return !pending().empty(); }
  // pin everything on the list (on the calling thread: the background thread of kick_background)
  static void run_pending_fills() {
    for (;;) {
      std::pair<size_t, int> job;
      {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        if (pending().empty()) return;
        job = pending().back(); pending().pop_back();
      }
      (void)hipSetDevice(job.second);
      void* q = nullptr;
      if (hipHostMalloc(&q, job.first, hipHostMallocPortable) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        size_t held = 0;
        for (auto& c : pool()) held += c.n;
        if (held + job.first <= ((size_t)2 << 30)) { pool().push_back({q, job.first, -1}); q = nullptr; }
      } else (void)hipGetLastError();
      if (q) (void)hipHostFree(q);
    }
  }
  void release() {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    size_t held = 0;
    for (auto& c : pool()) held += c.n;
    if (held + n <= ((size_t)2 << 30)) pool().push_back({p, n, -1}); else (void)hipHostFree(p);
    p = nullptr; n = 0;
  }
};


// Pinned blocks handed to the caller as result arrays (qs_hip_job::coef_up): the download staging
// buffer itself changes owner instead of being copied into freshly malloc'ed (page-faulting)
// memory; qs_hip_free() recognises such a block and puts it back into the pinned pool.
inline std::vector<CacheEntry>& pinned_handouts() { static auto* v = new std::vector<CacheEntry>; return *v; }
inline void* pinned_handout(PinnedBuf& b) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  pinned_handouts().push_back({b.p, b.n, -1});
  void* p = b.p;
  b.p = nullptr; b.n = 0;
  return p;
}
inline bool pinned_return(void* p) {
  PinnedBuf b;
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    auto& v = pinned_handouts();
    size_t i = 0;
    while (i < v.size() && v[i].p != p) ++i;
    if (i == v.size()) return false;
    b.p = v[i].p; b.n = v[i].n;
    v.erase(v.begin() + i);
  }
  return true;                                     // (b's destructor returns the block to the pool)
}

inline const size_t kStageMin = (size_t)1 << 20, kStageChunk = (size_t)8 << 20;
inline const int kStageThreads = 4;   // parts per chunk
inline const int kPoolThreads = 8;    // helper threads (several transfers can be in flight)

// Persistent helper threads for the host halves of the transfers (copying between
// caller memory and pinned staging).  Leaked on purpose: the threads sleep on the
// condition variable until the process ends.
class HostPool {
 public:
  // `ready` (optional) counts finished items per group of `per_group` consecutive items,
  // for callers that consume the work group by group (upload_pieces: one DMA per chunk)
  struct Task {
    std::function<void(int)> fn;
    int n = 0;
    std::atomic<int> next{0};
    int done = 0;                      // guarded by mu
    std::vector<int> group_done;       // guarded by mu
    int per_group = 1;
    std::mutex mu;
    std::condition_variable cv;        // signalled when a group or the whole task completes
  };
  typedef std::shared_ptr<Task> Handle;
  static HostPool& get() { static HostPool* p = new HostPool(kPoolThreads); return *p; }
  // fn(i) for every i in [0, n) on the helper threads, in index order; returns at once
  Handle submit(int n, std::function<void(int)> fn, int per_group = 0) {
    auto t = std::make_shared<Task>();
    t->fn = std::move(fn); t->n = n;
    if (per_group > 0) { t->per_group = per_group; t->group_done.assign((size_t)(n + per_group - 1) / per_group, 0); }
    { std::lock_guard<std::mutex> lk(mu_); q_.push_back(t); }
    cv_.notify_all();
    return t;
  }
  // the waiters sleep on the task's condition variable (no spinning: a host core per
  // in-flight call would otherwise be burned for the length of the transfer)
  static void wait(const Handle& t) {
    std::unique_lock<std::mutex> lk(t->mu);
    t->cv.wait(lk, [&] { return t->done >= t->n; });
  }
  static void wait_group(const Handle& t, int g) {
    std::unique_lock<std::mutex> lk(t->mu);
    const int want = std::min(t->per_group, t->n - g * t->per_group);
    t->cv.wait(lk, [&] { return t->group_done[(size_t)g] >= want; });
  }

 private:
  explicit HostPool(int helpers) {
    for (int i = 0; i < helpers; ++i) std::thread([this] { loop(); }).detach();
  }
  void loop() {
    for (;;) {
      Handle t;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !q_.empty(); });
        t = q_.front();
        if (t->next.load(std::memory_order_relaxed) >= t->n) { q_.pop_front(); continue; }
      }
      for (;;) {
        const int i = t->next.fetch_add(1, std::memory_order_relaxed);
        if (i >= t->n) break;
        t->fn(i);
        bool wake;
        {
          std::lock_guard<std::mutex> lk(t->mu);
          wake = ++t->done >= t->n;
          if (!t->group_done.empty()) {
            const int g = i / t->per_group;
            wake |= ++t->group_done[(size_t)g] >= std::min(t->per_group, t->n - g * t->per_group);
          }
        }
        if (wake) t->cv.notify_all();
      }
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Handle> q_;
};

// One transfer = several pageable pieces that sit back to back (at the given
// offsets) in one device arena.  `stage` must outlive the stream work.
struct Piece { void* host; size_t off, len; };

// bytes [lo, hi) of the arena image <-> the pieces that overlap them
inline void copy_range(char* stage, const std::vector<Piece>& pieces, size_t lo, size_t hi, bool to_stage) {
  for (const Piece& pc : pieces) {
    const size_t a = std::max(lo, pc.off), e = std::min(hi, pc.off + pc.len);
    if (e <= a) continue;
    if (to_stage) memcpy(stage + a, static_cast<const char*>(pc.host) + (a - pc.off), e - a);
    else memcpy(static_cast<char*>(pc.host) + (a - pc.off), stage + a, e - a);
  }
}
// item i of a transfer = part (i % kStageThreads) of chunk (i / kStageThreads)
inline void copy_item(char* stage, const std::vector<Piece>& pieces, size_t bytes, int i, bool to_stage) {
  const size_t c0 = (size_t)(i / kStageThreads) * kStageChunk, clen = std::min(kStageChunk, bytes - c0);
  const size_t part = (clen / kStageThreads + 63) & ~(size_t)63;
  const size_t o = std::min(clen, (size_t)(i % kStageThreads) * part), e = std::min(clen, o + part);
  if (e > o) copy_range(stage, pieces, c0 + o, c0 + e, to_stage);
}

// host -> device: the helpers gather 8 MiB chunks into pinned memory; this thread
// queues a chunk's DMA as soon as its parts are in, so DMA and gathering overlap
inline hipError_t upload_pieces(void* dst, const std::vector<Piece>& pieces, size_t bytes, hipStream_t s, PinnedBuf& stage) {
  // small transfers go straight from the caller's memory -- unless they come in many pieces
  // (a row table whose rows are not adjacent): then one staged DMA beats a copy call per row
  // ... and large ones too while no pinned block of their size is at hand (a cold process): pageable memory goes
  // up at ~44 GB/s here, pinning 128 MiB costs 23 ms (tools/cold_phases) -- unless they come in many pieces
  if ((bytes < kStageMin && pieces.size() <= 4) || !stage.alloc(bytes, /*optional=*/pieces.size() <= 16)) {
    for (const Piece& pc : pieces) {
      // A large pageable piece goes through the BLOCKING copy: measured on the MI355X box (tools/cold_phases,
      // profiles/r03c) hipMemcpy moves 128 MiB of pageable memory in 3 ms (44 GB/s: the runtime pins the user pages on
      // the fly), hipMemcpyAsync on a stream takes the bounce-buffer path at ~6 GB/s.  The copy is complete when
      // the call returns, so whatever is queued on `s` afterwards sees the data.
      hipError_t e = pc.len >= kStageMin
          ? hipMemcpy(static_cast<char*>(dst) + pc.off, pc.host, pc.len, hipMemcpyHostToDevice)
          : hipMemcpyAsync(static_cast<char*>(dst) + pc.off, pc.host, pc.len, hipMemcpyHostToDevice, s);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  const int nchunks = (int)((bytes + kStageChunk - 1) / kStageChunk);
  char* stg = static_cast<char*>(stage.p);
  const std::vector<Piece>* pcs = &pieces;
  HostPool::Handle h = HostPool::get().submit(nchunks * kStageThreads,
      [=](int i) { copy_item(stg, *pcs, bytes, i, true); }, kStageThreads);
  hipError_t err = hipSuccess;
  for (int c = 0; c < nchunks; ++c) {
    HostPool::wait_group(h, c);
    const size_t c0 = (size_t)c * kStageChunk, clen = std::min(kStageChunk, bytes - c0);
    if (err == hipSuccess)
      err = hipMemcpyAsync(static_cast<char*>(dst) + c0, stg + c0, clen, hipMemcpyHostToDevice, s);
  }
  HostPool::wait(h);
  return err;
}

// device -> host.  issue(): the copy into pinned memory is queued on the stream right behind the
// kernels that produce the data (8 MiB chunks, one event each), no host wait.  Then either
//   finish():          once the caller knows which pieces it wants, the helpers scatter each chunk to the
//                      caller's arrays as its event fires (DMA of chunk c+1 overlaps the scatter of chunk c).
//                      A HIP error may surface after some chunks have been written: only for callers that
//                      can put the original data back (they hold the pinned upload staging copy);
//   land() + scatter(): first EVERYTHING arrives in library-owned memory (the pinned stage, or a malloc'ed
//                      buffer when no pinned block was to be had) -- the only step that can fail --, then
//                      plain host copies into the caller's arrays.  For callers without a restore copy:
//                      a reported failure must leave the caller's arrays untouched.
// (Pageable D2H of a few MiB per call runs at 12-17 GB/s here, the staged path at the DMA rate; and results
// reach caller memory only after the range-check flags have been seen.)
struct Download {
  PinnedBuf stage;
  std::vector<hipEvent_t> ev;
  size_t bytes = 0;
  bool staged = false, landed = false;
  void* tmp = nullptr;                    // landing buffer of an unstaged download
  Download() = default;
  Download(const Download&) = delete;
  Download& operator=(const Download&) = delete;
  ~Download() { reset(); }
  void reset() {                          // (the copies must have completed)
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    ev.clear(); stage.release(); bytes = 0; staged = false; landed = false;
    free(tmp); tmp = nullptr;
  }

  // always_stage: the destination will be many separate pieces (see upload_pieces)
  hipError_t issue(const void* src, size_t nbytes, hipStream_t s, bool always_stage = false) {
    bytes = nbytes; landed = false;
    staged = nbytes > 0 && (nbytes >= kStageMin || always_stage) && stage.alloc(nbytes);
    if (!staged) return hipSuccess;
    for (size_t c0 = 0; c0 < bytes; c0 += kStageChunk) {
      const size_t clen = std::min(kStageChunk, bytes - c0);
      hipError_t e = hipMemcpyAsync(static_cast<char*>(stage.p) + c0, static_cast<const char*>(src) + c0, clen,
                                    hipMemcpyDeviceToHost, s);
      hipEvent_t evt = nullptr;
      if (e == hipSuccess) e = hipEventCreateWithFlags(&evt, hipEventDisableTiming);
      if (e == hipSuccess) { ev.push_back(evt); e = hipEventRecord(evt, s); }
      if (e != hipSuccess) {                       // nothing usable was staged: wait_first / finish must not index ev[]
        (void)hipStreamSynchronize(s);             // (a chunk copy may be in flight into `stage`)
        reset();
        bytes = nbytes;
        return e;
      }
    }
    return hipSuccess;
  }
  // everything queued before issue() on the stream has completed when this returns
  hipError_t wait_first(hipStream_t s) const { return staged ? hipEventSynchronize(ev[0]) : hipStreamSynchronize(s); }

  // all bytes are in library-owned host memory when this returns hipSuccess; nothing of the caller's is touched
  hipError_t land(const void* src, hipStream_t s) {
    if (landed) return hipSuccess;
    hipError_t e = hipSuccess;
    if (staged) {
      for (size_t c = 0; c < ev.size() && e == hipSuccess; ++c) e = hipEventSynchronize(ev[c]);
    } else if (bytes) {
      free(tmp);                                    // (a retry after a failed land must not leak the earlier buffer)
      tmp = malloc(bytes);
      if (!tmp) return hipErrorOutOfMemory;
      e = hipStreamSynchronize(s);                  // then the blocking copy: the fast pageable path (see upload_pieces)
      if (e == hipSuccess) e = hipMemcpy(tmp, src, bytes, hipMemcpyDeviceToHost);
    } else {
      e = hipStreamSynchronize(s);
    }
    if (e == hipSuccess && test_fail_finish()) e = hipErrorUnknown;        // (test hook: a late transfer failure)
    landed = e == hipSuccess;
    return e;
  }
  // the malloc'ed landing buffer of an unstaged, landed download changes owner (free() releases it)
  void* take_landed() { void* q = landed && !staged ? tmp : nullptr; if (q) tmp = nullptr; return q; }
  // landed bytes -> the caller's pieces: host copies only, cannot fail
  void scatter(const std::vector<Piece>& pieces) {
    if (!landed || pieces.empty() || !bytes) return;
    char* from = static_cast<char*>(staged ? stage.p : tmp);
    if (bytes < kStageMin) { copy_range(from, pieces, 0, bytes, false); return; }
    const int nchunks = (int)((bytes + kStageChunk - 1) / kStageChunk);
    const std::vector<Piece>* pcs = &pieces;
    const size_t nbytes = bytes;
    HostPool::wait(HostPool::get().submit(nchunks * kStageThreads, [=](int i) { copy_item(from, *pcs, nbytes, i, false); }));
  }

  // overlapped form (see above): land-and-scatter chunk by chunk
  // have_restore: the caller can put the pieces back should this fail half-way (it holds the pinned upload staging):
  // an UNSTAGED download (small, or pinned memory exhausted) then goes straight into the pieces, one blocking copy
  // each, instead of through a full-size temporary and a second host copy
  // (only for a handful of pieces -- kDirectPieces: with the row-pointer entry point and rows that are not adjacent in
  //  memory an image is thousands of pieces, i.e. thousands of small synchronous copies; those land in the temporary
  //  and are scattered by the helper threads as before)
  static constexpr size_t kDirectPieces = 16;
  hipError_t finish(const void* src, const std::vector<Piece>& pieces, hipStream_t s, bool have_restore = false) {
    if (!staged && !landed && have_restore && bytes && !pieces.empty() && pieces.size() <= kDirectPieces) {
      hipError_t e = hipStreamSynchronize(s);
      for (size_t i = 0; i < pieces.size() && e == hipSuccess; ++i)
        e = hipMemcpy(pieces[i].host, static_cast<const char*>(src) + pieces[i].off, pieces[i].len, hipMemcpyDeviceToHost);
      if (e == hipSuccess && test_fail_finish()) e = hipErrorUnknown;      // (test hook: fails AFTER the pieces were written)
      landed = e == hipSuccess;
      return e;
    }
    if (!staged || landed) {
      hipError_t e = land(src, s);
      if (e == hipSuccess) scatter(pieces);
      return e;
    }
    // a chunk is handed to the helpers only once it has arrived: a helper never waits
    // for the GPU, so transfers of other host threads are not held up behind this one
    const int nchunks = (int)ev.size();
    char* stg = static_cast<char*>(stage.p);
    const std::vector<Piece>* pcs = &pieces;
    const size_t nbytes = bytes;
    std::vector<HostPool::Handle> hs;
    hipError_t e = hipSuccess;
    for (int c = 0; c < nchunks && e == hipSuccess; ++c) {
      e = hipEventSynchronize(ev[c]);
      if (e == hipSuccess && !pieces.empty())
        hs.push_back(HostPool::get().submit(kStageThreads, [=](int t) { copy_item(stg, *pcs, nbytes, c * kStageThreads + t, false); }));
    }
    for (auto& h : hs) HostPool::wait(h);
    if (e == hipSuccess && test_fail_finish()) e = hipErrorUnknown;        // (test hook: fails AFTER the pieces were written)
    landed = e == hipSuccess;
    return e;
  }
};

// copy `bytes` from pageable `src` to device `dst` on `s`
inline hipError_t upload(void* dst, const void* src, size_t bytes, hipStream_t s, PinnedBuf& stage) {
  return upload_pieces(dst, std::vector<Piece>{{const_cast<void*>(src), 0, bytes}}, bytes, s, stage);
}

struct Streams {
  hipStream_t s[3] = {nullptr, nullptr, nullptr};   // s[0] always exists; s[1], s[2] are created on first use (get)
  hipEvent_t luma_done = nullptr;
  int dev = 0;                         // the device the streams belong to
  // A stream is a hardware queue: creating one costs milliseconds in a fresh process (tools/cold_phases), and a
  // one-shot job (the CLI) mostly needs one.  Falls back to s[0] if the queue cannot be created.
  hipStream_t get(int i) {
    if (!s[i]) {
      DeviceScope on(dev);
      if (hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); s[i] = nullptr; return s[0]; }
    }
    return s[i];
  }
  // s[i] if it exists already, else s[0]: a job that would merely like to overlap its groups does not stop for 9 ms
  // to create a queue (and 9 more for the first copy on it); the missing ones are created after the call (warm_all
  // from the background thread of PinnedBuf::kick_fills, or by qs_hip_prewarm)
  hipStream_t get_ready(int i) {
    if (s[i]) return s[i];
    want_more.store(true, std::memory_order_relaxed);
    return s[0];
  }
  static std::atomic<bool> want_more;
  // create the missing streams and run a first (tiny) copy on each; the object must not be in use by a job
  void warm_all() {
    DeviceScope on(dev);
    void* d = nullptr; void* h = nullptr;
    if (hipMalloc(&d, 4096) != hipSuccess || hipHostMalloc(&h, 4096, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); }
    for (int i = 0; i < 3; ++i) {
      hipStream_t x = get(i);
      if (x && d && h) { (void)hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, x); (void)hipMemcpyAsync(h, d, 4096, hipMemcpyDeviceToHost, x); (void)hipStreamSynchronize(x); }
    }
    if (d) (void)hipFree(d);
    if (h) (void)hipHostFree(h);
    (void)hipGetLastError();
  }
  int count() const { return (s[0] != nullptr) + (s[1] != nullptr) + (s[2] != nullptr); }
  void sync_all() { for (auto& x : s) if (x) (void)hipStreamSynchronize(x); }
  ~Streams() {
    DeviceScope on(dev);
    for (auto& x : s) if (x) (void)hipStreamDestroy(x);
    if (luma_done) (void)hipEventDestroy(luma_done);
  }
};

inline std::atomic<bool> Streams::want_more{false};
inline std::vector<Streams*>& g_stream_pool = *new std::vector<Streams*>;   // (leaked on purpose, see PinnedBuf::pool())

struct StreamLease {     // borrow a ready-made set of streams of the CURRENT device, give it back on scope exit
  Streams* p = nullptr;
  StreamLease() {
    const int dev = current_device();
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      size_t best = g_stream_pool.size();          // of this device's sets, the one with the most queues
      for (size_t i = 0; i < g_stream_pool.size(); ++i)
        if (g_stream_pool[i]->dev == dev && (best == g_stream_pool.size() || g_stream_pool[i]->count() > g_stream_pool[best]->count())) best = i;
      if (best < g_stream_pool.size()) { p = g_stream_pool[best]; g_stream_pool.erase(g_stream_pool.begin() + best); return; }
    }
    Streams* n = new (std::nothrow) Streams;
    if (!n) return;
    n->dev = dev;
    bool ok = hipStreamCreateWithFlags(&n->s[0], hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&n->luma_done, hipEventDisableTiming) == hipSuccess;
    if (!ok) { delete n; return; }
    p = n;
  }
  StreamLease(const StreamLease&) = delete;
  StreamLease& operator=(const StreamLease&) = delete;
  StreamLease(StreamLease&& o) noexcept : p(o.p) { o.p = nullptr; }
  ~StreamLease() {
    if (!p) return;
    p->sync_all();                                        // nothing of this job may outlive it
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_stream_pool.push_back(p);
  }
};

// error paths: nothing may be freed while the streams still run.  Declare it AFTER the
// buffers it protects (locals are destroyed in reverse order).
struct DrainGuard {
  Streams* st;
  ~DrainGuard() { if (st) st->sync_all(); }
};

// What a call missed -- pinned staging blocks (PinnedBuf::fill_later), extra streams (Streams::get_ready) -- is set
// up by ONE background thread that starts when the call is over (the entry points call this on their way out):
// pinning takes the process's mmap lock and creating a queue takes 9 ms; done during the job both slowed it down.
inline std::atomic<bool> g_bg_busy{false};         // the background thread of kick_background is at work
inline void kick_background() {
  if (!PinnedBuf::fills_pending() && !Streams::want_more.load(std::memory_order_relaxed)) return;
  if (g_bg_busy.exchange(true)) return;
  const int dev = current_device();
  try {
    std::thread([dev] {
      (void)hipSetDevice(dev);
      PinnedBuf::run_pending_fills();
      if (Streams::want_more.exchange(false)) {
        // every pooled set of this device that lacks queues (taken out of the pool meanwhile: no job uses it)
        for (;;) {
          Streams* st = nullptr;
          {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            for (size_t i = 0; i < g_stream_pool.size(); ++i)
              if (g_stream_pool[i]->dev == dev && g_stream_pool[i]->count() < 3) { st = g_stream_pool[i]; g_stream_pool.erase(g_stream_pool.begin() + i); break; }
          }
          if (!st) break;
          st->warm_all();
          const bool done = st->count() == 3;
          { std::lock_guard<std::mutex> lk(g_cache_mu); g_stream_pool.push_back(st); }
          if (!done) break;                        // (queue creation failed: do not spin)
        }
      }
      g_bg_busy = false;
    }).detach();
  } catch (...) { g_bg_busy = false; }
}

}  // namespace qsx
