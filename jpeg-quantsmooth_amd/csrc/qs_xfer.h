// qs_xfer.h -- pooled device / pinned buffers, streams, the helper-thread pool and the
// staged host<->device transfers of the job layer.  Included by qs_job.cpp only.
#pragma once
#include <new>
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

namespace {


struct CacheEntry { void* p; size_t n; };
static std::mutex g_cache_mu;
static std::vector<CacheEntry> g_cache;            // free device blocks
static const size_t kCacheMaxBytes = (size_t)6 << 30;

static size_t round_size(size_t n) {               // size classes: powers of two from 64 KiB
  size_t c = (size_t)64 << 10;
  while (c < n) c <<= 1;
  return c;
}

struct DevBuf {
  void* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  hipError_t alloc(size_t bytes) {
    release();
    const size_t want = round_size(bytes);
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      for (size_t i = 0; i < g_cache.size(); ++i)
        if (g_cache[i].n == want) { p = g_cache[i].p; n = want; g_cache.erase(g_cache.begin() + i); return hipSuccess; }
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
    std::lock_guard<std::mutex> lk(g_cache_mu);
    size_t held = 0;
    for (auto& c : g_cache) held += c.n;
    if (held + n <= kCacheMaxBytes) g_cache.push_back({p, n}); else (void)hipFree(p);
    p = nullptr; n = 0;
  }
  void take(DevBuf& o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// ---- host -> device upload of large pageable buffers -------------------------
// Measured on the MI355X box (tools/ubench_pcie.hip, 128 MiB): pageable
// hipMemcpy H2D 17-19 GB/s, pinned 57 GB/s, hipHostRegister 6 ms + 57 GB/s,
// memcpy into pinned memory 30 GB/s with one thread and 60-90 GB/s with 4-8;
// pageable D2H already runs at 55 GB/s.  So uploads above a few MiB go through a
// pooled pinned staging buffer: four threads copy 8 MiB chunks into it and each
// chunk's DMA is queued as soon as it is complete, overlapping the next copy.
struct PinnedBuf {
  void* p = nullptr;
  size_t n = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { release(); }
  static std::vector<CacheEntry>& pool() { static std::vector<CacheEntry> v; return v; }
  bool alloc(size_t bytes) {
    const size_t want = round_size(bytes);
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      auto& v = pool();
      for (size_t i = 0; i < v.size(); ++i)
        if (v[i].n == want) { p = v[i].p; n = want; v.erase(v.begin() + i); return true; }
    }
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
    n = want;
    return true;
  }
  void release() {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    size_t held = 0;
    for (auto& c : pool()) held += c.n;
    if (held + n <= ((size_t)2 << 30)) pool().push_back({p, n}); else (void)hipHostFree(p);
    p = nullptr; n = 0;
  }
};


static const size_t kStageMin = (size_t)1 << 20, kStageChunk = (size_t)8 << 20;
static const int kStageThreads = 4;   // parts per chunk
static const int kPoolThreads = 8;    // helper threads (several transfers can be in flight)

// Persistent helper threads for the host halves of the transfers (copying between
// caller memory and pinned staging).  Leaked on purpose: the threads sleep on the
// condition variable until the process ends.
class HostPool {
 public:
  struct Task { std::function<void(int)> fn; int n = 0; std::atomic<int> next{0}, done{0}; };
  typedef std::shared_ptr<Task> Handle;
  static HostPool& get() { static HostPool* p = new HostPool(kPoolThreads); return *p; }
  // fn(i) for every i in [0, n) on the helper threads, in index order; returns at once
  Handle submit(int n, std::function<void(int)> fn) {
    auto t = std::make_shared<Task>();
    t->fn = std::move(fn); t->n = n;
    { std::lock_guard<std::mutex> lk(mu_); q_.push_back(t); }
    cv_.notify_all();
    return t;
  }
  static void wait(const Handle& t) {
    while (t->done.load(std::memory_order_acquire) < t->n) std::this_thread::yield();
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
        t->done.fetch_add(1, std::memory_order_release);
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
static void copy_range(char* stage, const std::vector<Piece>& pieces, size_t lo, size_t hi, bool to_stage) {
  for (const Piece& pc : pieces) {
    const size_t a = std::max(lo, pc.off), e = std::min(hi, pc.off + pc.len);
    if (e <= a) continue;
    if (to_stage) memcpy(stage + a, static_cast<const char*>(pc.host) + (a - pc.off), e - a);
    else memcpy(static_cast<char*>(pc.host) + (a - pc.off), stage + a, e - a);
  }
}
// item i of a transfer = part (i % kStageThreads) of chunk (i / kStageThreads)
static void copy_item(char* stage, const std::vector<Piece>& pieces, size_t bytes, int i, bool to_stage) {
  const size_t c0 = (size_t)(i / kStageThreads) * kStageChunk, clen = std::min(kStageChunk, bytes - c0);
  const size_t part = (clen / kStageThreads + 63) & ~(size_t)63;
  const size_t o = std::min(clen, (size_t)(i % kStageThreads) * part), e = std::min(clen, o + part);
  if (e > o) copy_range(stage, pieces, c0 + o, c0 + e, to_stage);
}

// host -> device: the helpers gather 8 MiB chunks into pinned memory; this thread
// queues a chunk's DMA as soon as its parts are in, so DMA and gathering overlap
static hipError_t upload_pieces(void* dst, const std::vector<Piece>& pieces, size_t bytes, hipStream_t s, PinnedBuf& stage) {
  if (bytes < kStageMin || !stage.alloc(bytes)) {
    for (const Piece& pc : pieces) {
      hipError_t e = hipMemcpyAsync(static_cast<char*>(dst) + pc.off, pc.host, pc.len, hipMemcpyHostToDevice, s);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  const int nchunks = (int)((bytes + kStageChunk - 1) / kStageChunk);
  auto done = std::make_shared<std::vector<std::atomic<int>>>(nchunks);
  for (auto& d : *done) d.store(0);
  char* stg = static_cast<char*>(stage.p);
  const std::vector<Piece>* pcs = &pieces;
  HostPool::Handle h = HostPool::get().submit(nchunks * kStageThreads, [=](int i) {
    copy_item(stg, *pcs, bytes, i, true);
    (*done)[i / kStageThreads].fetch_add(1, std::memory_order_release);
  });
  hipError_t err = hipSuccess;
  for (int c = 0; c < nchunks; ++c) {
    while ((*done)[c].load(std::memory_order_acquire) < kStageThreads) std::this_thread::yield();
    const size_t c0 = (size_t)c * kStageChunk, clen = std::min(kStageChunk, bytes - c0);
    if (err == hipSuccess)
      err = hipMemcpyAsync(static_cast<char*>(dst) + c0, stg + c0, clen, hipMemcpyHostToDevice, s);
  }
  HostPool::wait(h);
  return err;
}

// device -> host in two steps.  issue(): the copy into pinned memory is queued on the
// stream right behind the kernels that produce the data (8 MiB chunks, one event
// each), no host wait.  finish(): once the caller knows which pieces it wants, the
// helpers scatter each chunk to the caller's arrays as its event fires.  (Pageable
// D2H of a few MiB per call runs at 12-17 GB/s here, this path at the DMA rate; and
// results reach caller memory only after the range-check flags have been seen.)
struct Download {
  PinnedBuf stage;
  std::vector<hipEvent_t> ev;
  size_t bytes = 0;
  bool staged = false;
  Download() = default;
  Download(const Download&) = delete;
  Download& operator=(const Download&) = delete;
  ~Download() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }

  hipError_t issue(const void* src, size_t nbytes, hipStream_t s) {
    bytes = nbytes;
    staged = nbytes >= kStageMin && stage.alloc(nbytes);
    if (!staged) return hipSuccess;
    for (size_t c0 = 0; c0 < bytes; c0 += kStageChunk) {
      const size_t clen = std::min(kStageChunk, bytes - c0);
      hipError_t e = hipMemcpyAsync(static_cast<char*>(stage.p) + c0, static_cast<const char*>(src) + c0, clen,
                                    hipMemcpyDeviceToHost, s);
      hipEvent_t evt = nullptr;
      if (e == hipSuccess) e = hipEventCreateWithFlags(&evt, hipEventDisableTiming);
      if (e != hipSuccess) return e;
      ev.push_back(evt);
      if ((e = hipEventRecord(evt, s)) != hipSuccess) return e;
    }
    return hipSuccess;
  }
  // everything queued before issue() on the stream has completed when this returns
  hipError_t wait_first(hipStream_t s) const { return staged ? hipEventSynchronize(ev[0]) : hipStreamSynchronize(s); }

  hipError_t finish(const void* src, const std::vector<Piece>& pieces, hipStream_t s) {
    if (!staged) {
      for (const Piece& pc : pieces) {
        hipError_t e = hipMemcpyAsync(pc.host, static_cast<const char*>(src) + pc.off, pc.len, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return e;
      }
      return hipStreamSynchronize(s);
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
    return e;
  }
};

// copy `bytes` from pageable `src` to device `dst` on `s`
static hipError_t upload(void* dst, const void* src, size_t bytes, hipStream_t s, PinnedBuf& stage) {
  return upload_pieces(dst, std::vector<Piece>{{const_cast<void*>(src), 0, bytes}}, bytes, s, stage);
}

struct Streams {
  hipStream_t s[3] = {nullptr, nullptr, nullptr};
  hipEvent_t luma_done = nullptr;
  ~Streams() {
    for (auto& x : s) if (x) (void)hipStreamDestroy(x);
    if (luma_done) (void)hipEventDestroy(luma_done);
  }
};

static std::vector<Streams*> g_stream_pool;

struct StreamLease {     // borrow a ready-made set of streams, give it back on scope exit
  Streams* p = nullptr;
  StreamLease() {
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      if (!g_stream_pool.empty()) { p = g_stream_pool.back(); g_stream_pool.pop_back(); return; }
    }
    Streams* n = new (std::nothrow) Streams;
    if (!n) return;
    bool ok = true;
    for (int i = 0; i < 3 && ok; ++i) ok = hipStreamCreateWithFlags(&n->s[i], hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&n->luma_done, hipEventDisableTiming) == hipSuccess;
    if (!ok) { delete n; return; }
    p = n;
  }
  ~StreamLease() {
    if (!p) return;
    for (auto& x : p->s) (void)hipStreamSynchronize(x);   // nothing of this job may outlive it
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_stream_pool.push_back(p);
  }
};

}  // namespace
