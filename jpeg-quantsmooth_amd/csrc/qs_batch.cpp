// qs_batch.cpp -- many jobs in one call (qs_hip_do_quantsmooth_batch; an addition to the reference API for
// callers that serve many images, FROZEN since round 2): whole jobs spread over the configured devices,
// independent-component jobs as plane-set groups (qs_fused.cpp), coupled YCbCr jobs advancing in groups
// (run_coupled below), everything else job by job.  (Split out of qs_job.cpp.)
#include <functional>
#include <list>
#include <system_error>

#include "qs_jobint.h"

using namespace qsx;
using namespace qsj;

namespace {

// ---------------------------------------------------------------------------
// coupled execution: several YCbCr jobs whose chroma depends on luma (JOINT_YUV / UPSAMPLE_UV, CLI
// --quality 5/6) advance together -- the order of run_job (= the reference's component order,
// quantsmooth.h:2488-2752), stage by stage for the whole group: all luma planes as one plane set,
// then the low-res luma planes, then all chroma planes as one plane set, then the upsampling.  One
// image's planes alone leave most of the chip idle (1080p: 510 + 2 x 128 wave groups on 1024 SIMDs);
// a group of eight fills it.  Arithmetic and per-job semantics are run_job's eager mode: flags are
// read at the end, a job with a set range-check flag is re-run in the careful order from its
// untouched host input.
static bool job_couplable(const qs_hip_job* job, int flags, int niter) {
  static const bool off = getenv("QS_HIP_NO_COUPLE") != nullptr;
  if (off || niter < 1 || (flags & QS_LOW_QUALITY) || !job_needs_lowres(job, flags)) return false;
  for (int ci = 0; ci < 3; ++ci) {
    if (!job->has_quant[ci]) return false;
    int acc = 0;
    for (int i = 0; i < 64; ++i) acc |= job->quant[ci][i];
    if (acc <= 1 || acc >= 0x800) return false;            // iterations skipped / stop: the general path knows how
  }
  return true;
}

// Groups arrive from several worker threads at once.  Left alone they move in lockstep -- all upload,
// then all compute, then all download -- and nothing overlaps.  At most kSlots groups per device may
// have kernels queued at a time: the others upload meanwhile and start computing when an earlier
// group's kernels have finished and its results are on their way back.
struct ComputeSlots {
  std::mutex mu;
  std::condition_variable cv;
  int busy[64] = {0};
  static ComputeSlots& get() { static ComputeSlots c; return c; }
};
struct ComputeSlot {
  int dev; bool held = false;
  explicit ComputeSlot(int d) : dev(d & 63) {
    const int kSlots = (int)env_size("QS_HIP_COUPLE_SLOTS", 2);
    ComputeSlots& c = ComputeSlots::get();
    std::unique_lock<std::mutex> lk(c.mu);
    c.cv.wait(lk, [&] { return c.busy[dev] < kSlots; });
    ++c.busy[dev]; held = true;
  }
  void release() {
    if (!held) return;
    ComputeSlots& c = ComputeSlots::get();
    { std::lock_guard<std::mutex> lk(c.mu); --c.busy[dev]; }
    c.cv.notify_all(); held = false;
  }
  ~ComputeSlot() { release(); }
  ComputeSlot(const ComputeSlot&) = delete;
  ComputeSlot& operator=(const ComputeSlot&) = delete;
};

static size_t plane_stride(int wb, int hb) { return (qs_hip_plane_bytes(wb, hb) + 255) & ~(size_t)255; }

static int run_coupled(qs_hip_job* const* jobs, const std::vector<int>& which, int flags, int niter, int* results) {
  StreamLease lease;
  if (!lease.p) return qs_fail(QS_HIP_ENODEV, "could not create HIP streams: %s", hipGetErrorString(hipGetLastError()));
  hipStream_t s = lease.p->s[0];
  const double t_start = wall_ms();
  const int G = (int)which.size();
  if (G < 1 || 2 * G > QS_MAX_PLANES) return qs_fail(QS_HIP_EINVAL, "run_coupled: bad group size %d", G);
  const bool joint = (flags & QS_JOINT_YUV) != 0;
  const int diag = (flags & QS_DIAGONALS) != 0;

  struct CJob {
    bool sub, upsample;
    size_t coef_off[3], px_off[3], l_off, upx_off[2], upc_off[2], ubytes;
    int cst[3];
  };
  std::vector<CJob> cj((size_t)G);
  std::vector<QsConsts> hc;
  DevBuf coef, px, cst, status, upx, upc;
  PinnedBuf stage, hstatus;
  Download down;
  std::list<Download> down_up;                              // one per replacement array: each becomes the caller's
  std::vector<Download*> down_up_of((size_t)G * 2, nullptr);
  DrainGuard drain{lease.p};                                // (after every buffer: the stream is drained first)

  // ---- layout.  Coefficients: [all luma][all chroma], so each class is clamped by one launch.
  size_t coef_bytes = 0, px_bytes = 0, upx_bytes = 0, upc_bytes = 0;
  std::vector<const uint16_t*> qtabs;
  for (int pass = 0; pass < 2; ++pass)
    for (int g = 0; g < G; ++g) {
      const qs_hip_job* job = jobs[which[g]];
      CJob& J = cj[g];
      for (int ci = pass ? 1 : 0; ci < (pass ? 3 : 1); ++ci) {
        const size_t nb = (size_t)job->wblk[ci] * job->hblk[ci];
        J.coef_off[ci] = coef_bytes; coef_bytes += nb * 128;
        J.px_off[ci] = px_bytes; px_bytes += 2 * plane_stride(job->wblk[ci], job->hblk[ci]);   // two planes: fused pass A ping-pongs
        const uint16_t* q = job->quant[ci];
        J.cst[ci] = -1;
        for (size_t k = 0; k < qtabs.size() && J.cst[ci] < 0; ++k)
          if (!memcmp(qtabs[k], q, 64 * sizeof(uint16_t))) J.cst[ci] = (int)k;
        if (J.cst[ci] < 0) { J.cst[ci] = (int)qtabs.size(); qtabs.push_back(q); }
      }
      if (pass) continue;
      const int ws = job->hsamp[0], hs = job->vsamp[0];
      J.sub = !(ws == 1 && hs == 1);
      J.upsample = (flags & QS_UPSAMPLE_UV) && J.sub;        // reference :2805: image1 only when subsampled
      J.l_off = 0; J.ubytes = 0;
      if (J.sub) { J.l_off = px_bytes; px_bytes += (qs_hip_plane_bytes(job->wblk[1], job->hblk[1]) + 255) & ~(size_t)255; }
      if (J.upsample) {
        J.ubytes = (size_t)job->wblk[0] * job->hblk[0] * 128;
        for (int k = 0; k < 2; ++k) {
          J.upx_off[k] = upx_bytes; upx_bytes += (qs_hip_upsample_bytes(job->image_width, job->image_height, ws, hs) + 255) & ~(size_t)255;
          J.upc_off[k] = upc_bytes; upc_bytes += J.ubytes;
        }
      }
    }
  HIP_TRY(coef.alloc(coef_bytes));
  HIP_TRY(px.alloc(px_bytes));
  HIP_TRY(cst.alloc(qtabs.size() * sizeof(QsConsts)));
  HIP_TRY(status.alloc((size_t)G * 3 * sizeof(int32_t)));
  if (upx_bytes) { HIP_TRY(upx.alloc(upx_bytes)); HIP_TRY(upc.alloc(upc_bytes)); }
  hc.resize(qtabs.size());
  for (size_t k = 0; k < qtabs.size(); ++k)
    if (int r = qs_hip_consts_build(&hc[k], qtabs[k], flags)) return r;
  HIP_TRY(hipMemcpyAsync(cst.p, hc.data(), qtabs.size() * sizeof(QsConsts), hipMemcpyHostToDevice, s));
  {
    std::vector<Piece> pieces;
    for (int g = 0; g < G; ++g)
      for (int ci = 0; ci < 3; ++ci)
        host_pieces(jobs[which[g]], ci, 0, jobs[which[g]]->hblk[ci], cj[g].coef_off[ci], pieces);
    HIP_TRY(upload_pieces(coef.p, pieces, coef_bytes, s, stage));
  }
  HIP_TRY(hipMemsetAsync(status.p, 0, (size_t)G * 3 * sizeof(int32_t), s));
  ComputeSlot slot(current_device());                        // (released when this group's kernels have finished)

  auto coef_of = [&](int g, int ci) { return reinterpret_cast<int16_t*>(coef.as<char>() + cj[g].coef_off[ci]); };
  // Every component has two pixel planes: a pass B that is followed by another pass A (the next iteration's, or the
  // refresh pass the later stages read) writes that plane itself -- fused pass A -- and the two swap roles.
  std::vector<unsigned char> cur((size_t)G * 3, 0);          // which of its two planes is component (g, ci)'s current one
  auto plane_at = [&](int g, int ci, int which_) {
    return px.as<uint8_t>() + cj[g].px_off[ci] + (which_ ? plane_stride(jobs[which[g]]->wblk[ci], jobs[which[g]]->hblk[ci]) : 0);
  };
  auto plane_of = [&](int g, int ci) { return plane_at(g, ci, cur[(size_t)g * 3 + ci]); };
  auto lowres_of = [&](int g) { return cj[g].sub ? px.as<uint8_t>() + cj[g].l_off : plane_of(g, 0); };
  // next(g, ci): does the coming pass B write this plane's successor?
  auto make_set = [&](QsPlaneSet& set, int ci0, int ci1, const std::function<bool(int, int)>& next) {
    memset(&set, 0, sizeof set);
    int w = 0, n = 0;
    for (int g = 0; g < G; ++g)
      for (int ci = ci0; ci < ci1; ++ci, ++n) {
        const qs_hip_job* job = jobs[which[g]];
        set.wave0[n] = w;
        w += (job->wblk[ci] * job->hblk[ci] + 63) / 64;
        QsPlaneRef& R = set.ref[n];
        R.cst = cst.as<QsConsts>() + cj[g].cst[ci];
        R.coef = coef_of(g, ci);
        R.plane = plane_of(g, ci);
        R.plane_next = next(g, ci) ? plane_at(g, ci, !cur[(size_t)g * 3 + ci]) : nullptr;
        R.status = status.as<int32_t>() + g * 3 + ci;
        R.wblk = job->wblk[ci]; R.hblk = job->hblk[ci]; R.pitch = qs_plane_pitch(job->wblk[ci]);
        R.mode = QS_PLANE_REP_TOP | QS_PLANE_REP_BOT | (comp_rebalance(job, ci, flags) ? QS_PLANE_REBALANCE : 0);
      }
    set.n = n;
    for (int i = n; i < QS_MAX_PLANES + 2; ++i) set.wave0[i] = w;
  };
  auto flip = [&](int ci0, int ci1, const std::function<bool(int, int)>& next) {
    for (int g = 0; g < G; ++g) for (int ci = ci0; ci < ci1; ++ci) if (next(g, ci)) cur[(size_t)g * 3 + ci] ^= 1;
  };
  const std::function<bool(int, int)> none = [](int, int) { return false; }, all = [](int, int) { return true; };
  QsPlaneSet set;

  // ---- luma: pass A once, niter iterations; the last one also writes the refresh the chroma stages read (reference
  // :2495, :2622) and carries the +-1023 clamp: the fused IDCT reads the unclamped coefficients the kernel holds, so
  // the refresh is that of the unclamped luma, as in the reference, whose clamp sits behind its loop (:2668-2689).
  make_set(set, 0, 1, none);
  qs_launch_idct_set(set, 1, s);
  for (int it = 0; it < niter; ++it) {
    make_set(set, 0, 1, all);
    qs_launch_smooth_set(set, diag, it == niter - 1, s);
    flip(0, 1, all);
  }
  for (int g = 0; g < G; ++g) {                              // image2 (reference :2753-2815)
    const qs_hip_job* job = jobs[which[g]];
    if (cj[g].sub)
      qs_launch_downsample(plane_of(g, 0), job->wblk[0], job->hblk[0], lowres_of(g), job->wblk[1], job->hblk[1],
                           job->hsamp[0], job->vsamp[0], s);
  }

  // ---- chroma.  A job that is upsampled afterwards needs one more refresh (the last pass B writes it); jobs of
  // both kinds may share a group.  The clamp rides on the last iteration for all of them.
  QsPlaneAux lowres;
  memset(&lowres, 0, sizeof lowres);
  for (int g = 0; g < G; ++g) lowres.p[2 * g] = lowres.p[2 * g + 1] = lowres_of(g);
  bool any_up = false;
  for (int g = 0; g < G; ++g) any_up |= cj[g].upsample;
  const std::function<bool(int, int)> ups = [&](int g, int) { return cj[g].upsample; };
  make_set(set, 1, 3, none);
  qs_launch_idct_set(set, 1, s);
  for (int it = 0; it < niter; ++it) {
    const std::function<bool(int, int)>& next = it < niter - 1 ? all : ups;
    make_set(set, 1, 3, next);
    if (joint)                                               // JOINT_YUV acts through the low-res luma plane (reference :2636)
      qs_launch_joint_set(set, lowres, 0, 0, s);
    qs_launch_smooth_set(set, diag, it == niter - 1, s);
    flip(1, 3, next);
  }
  if (any_up) {
    for (int g = 0; g < G; ++g) {                            // UPSAMPLE_UV (reference :2691-2752)
      if (!cj[g].upsample) continue;
      const qs_hip_job* job = jobs[which[g]];
      const int ws = job->hsamp[0], hs = job->vsamp[0];
      const int w1 = (job->image_width + ws - 1) / ws, h1 = (job->image_height + hs - 1) / hs;
      const size_t pitch = qs_hip_upsample_pitch(job->image_width, ws);
      for (int k = 0; k < 2; ++k) {
        uint8_t* opx = upx.as<uint8_t>() + cj[g].upx_off[k];
        int16_t* oc = reinterpret_cast<int16_t*>(upc.as<char>() + cj[g].upc_off[k]);
        qs_launch_upsample(plane_of(g, 1 + k), lowres_of(g), job->wblk[1 + k], plane_of(g, 0), job->wblk[0], opx, (int)pitch,
                           job->wblk[0] * 8, job->hblk[0] * 8, w1, h1, h1 < 8 ? h1 : 8, ws, hs, s);
        qs_launch_fdct_plane(opx, (int)pitch, oc, job->wblk[0], job->hblk[0], s);
      }
    }
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(lease.p->luma_done, s));            // "this group's kernels are done"

  // ---- flags and results into pinned memory behind the kernels
  if (!hstatus.alloc((size_t)G * 3 * sizeof(int32_t))) return qs_fail(QS_HIP_ENOMEM, "out of pinned host memory");
  HIP_TRY(hipMemcpyAsync(hstatus.p, status.p, (size_t)G * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(down.issue(coef.p, coef_bytes, s, true));
  for (int g = 0; g < G; ++g)
    for (int k = 0; k < 2 && cj[g].upsample; ++k) {
      down_up.emplace_back();
      down_up_of[(size_t)g * 2 + k] = &down_up.back();
      HIP_TRY(down_up.back().issue(upc.as<char>() + cj[g].upc_off[k], cj[g].ubytes, s));
    }
  const double t_enq = wall_ms();
  HIP_TRY(hipEventSynchronize(lease.p->luma_done));
  slot.release();
  HIP_TRY(down.wait_first(s));

  // ---- scatter (the only place host memory is written); jobs whose range check tripped stay untouched
  const int32_t* hst = static_cast<const int32_t*>(hstatus.p);
  std::vector<int> rerun;
  std::vector<Piece> back;
  for (int g = 0; g < G; ++g) {
    if (hst[g * 3] | hst[g * 3 + 1] | hst[g * 3 + 2]) { rerun.push_back(which[g]); continue; }
    for (int ci = 0; ci < 3; ++ci)
      host_pieces(jobs[which[g]], ci, 0, jobs[which[g]]->hblk[ci], cj[g].coef_off[ci], back);
  }
  // A failure from here on must leave every job as it came in.  With a restore copy (the upload staging) the caller's
  // blocks are put back on any error exit below; without one, everything -- the coefficients and the replacement
  // arrays -- lands in library-owned host memory BEFORE the first byte of the caller's is written, and nothing can
  // fail after that.
  auto land_up = [&]() -> hipError_t {                       // replacement arrays into pinned staging / malloc'ed landing buffers
    for (int g = 0; g < G; ++g) {
      if (!cj[g].upsample) continue;
      const bool bad = (hst[g * 3] | hst[g * 3 + 1] | hst[g * 3 + 2]) != 0;
      for (int k = 0; k < 2; ++k) {
        Download& D = *down_up_of[(size_t)g * 2 + k];
        if (D.staged || !bad)
          if (hipError_t e = D.land(upc.as<char>() + cj[g].upc_off[k], s)) return e;
      }
    }
    return hipSuccess;
  };
  if (!stage.p) { HIP_TRY(down.land(coef.p, s)); HIP_TRY(land_up()); }
  hipError_t e = down.finish(coef.p, back, s, stage.p != nullptr);               // scatter, chunk by chunk as the chunks arrive
  if (e == hipSuccess) e = land_up();                        // (with a restore copy: overlapped with the scatter above)
  if (e != hipSuccess) {                                     // a late failure: put the original blocks back
    (void)hipStreamSynchronize(s);
    if (stage.p) for (const Piece& pc : back) memcpy(pc.host, static_cast<const char*>(stage.p) + pc.off, pc.len);
    return qs_fail(e == hipErrorOutOfMemory ? QS_HIP_ENOMEM : QS_HIP_ENODEV, "download failed: %s", hipGetErrorString(e));
  }
  for (int g = 0; g < G; ++g) {
    qs_hip_job* job = jobs[which[g]];
    if (hst[g * 3] | hst[g * 3 + 1] | hst[g * 3 + 2]) continue;
    if (cj[g].upsample) {                                    // reference :2836-2849
      for (int k = 0; k < 2; ++k) {
        Download& D = *down_up_of[(size_t)g * 2 + k];
        // a staged array IS the pinned download buffer: it changes owner (qs_hip_free gives it back to the pool)
        // ... an unstaged one is its malloc'ed landing buffer
        job->coef_up[k] = static_cast<int16_t*>(D.staged ? pinned_handout(D.stage) : D.take_landed());
      }
      job->up_wblk = job->wblk[0]; job->up_hblk = job->hblk[0];
      job->out_hsamp0 = job->out_vsamp0 = 1;
    }
    for (int ci = 0; ci < 3; ++ci)                           // reference :2851-2859
      for (int i = 0; i < 64; ++i) job->quant[ci][i] = 1;
    results[which[g]] = 0;
  }
  if (trace_on())
    fprintf(stderr, "qs_hip trace: coupled  %d job(s)  enqueue %.2f ms  drain+scatter %.2f ms  (%zu re-run)\n",
            G, t_enq - t_start, wall_ms() - t_enq, rerun.size());
  HIP_TRY(hipStreamSynchronize(s));
  coef.release(); px.release(); cst.release(); status.release(); upx.release(); upc.release();   // before the re-runs allocate
  for (int ji : rerun)
    results[ji] = run_job(jobs[ji], flags, niter, 0, nullptr, nullptr, /*eager=*/false);
  return QS_HIP_OK;
}

}  // namespace

static int do_quantsmooth_batch_impl(qs_hip_job* const* jobs, int njobs, int flags, int niter, int* results) {
  if (!jobs || !results || njobs < 0) return qs_fail(QS_HIP_EINVAL, "qs_hip_do_quantsmooth_batch: null argument");
  std::vector<int> fused, single;
  const int nit = niter < 0 ? 0 : niter > 100 ? 100 : niter;     // reference :2455-2456
  for (int j = 0; j < njobs; ++j) {
    int n1 = niter;
    const int todo = prepare_job(jobs[j], flags, &n1);
    results[j] = todo < 0 ? todo : 0;
    if (todo <= 0) continue;
    // a job large enough to be spread over several GPUs goes there on its own
    const bool fuse = job_fusable(jobs[j], flags) && shard_devices_for(jobs[j], flags, n1).empty();
    (fuse ? fused : single).push_back(j);
  }
  if (fused.empty() && single.empty()) return QS_HIP_OK;
  warm_wait();
  if (qs_hip_device_count() <= 0)
    return qs_fail(QS_HIP_ENODEV, "no HIP device available (this library has no CPU fallback)");
  // Jobs large enough to be cut over several GPUs run alone (run_sharded); everything else is spread
  // over the configured devices as WHOLE jobs -- independent objects, no exchange between devices:
  // every device gets a share of the plane-set jobs (one run_fused per device) and of the coupled /
  // special jobs (general route, a chain of small launches per job: up to four in flight per device,
  // each from its own host thread with its own stream set -- the job layer is thread-safe).
  std::vector<int> small, large;
  for (int j : single) (shard_devices_for(jobs[j], flags, nit).empty() ? small : large).push_back(j);
  std::vector<int> devs = configured_devices();
  if (devs.empty() || fused.size() + small.size() < 2) devs.assign(1, current_device());
  const size_t nd = devs.size();
  std::vector<std::vector<int>> fused_of(nd);
  {                                                           // greedy balance by block count
    std::vector<size_t> load(nd, 0);
    std::vector<int> order(fused);
    auto blocks_of = [&](int j) { size_t b = 0; for (int ci = 0; ci < jobs[j]->ncomp; ++ci) b += (size_t)jobs[j]->wblk[ci] * jobs[j]->hblk[ci]; return b; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return blocks_of(a) > blocks_of(b); });
    for (int j : order) {
      const size_t d = std::min_element(load.begin(), load.end()) - load.begin();
      fused_of[d].push_back(j); load[d] += blocks_of(j);
    }
    for (auto& v : fused_of) std::sort(v.begin(), v.end());
  }
  for (int j : fused) results[j] = QS_HIP_ENODEV;
  // the coupled YCbCr jobs among them advance in groups (run_coupled); what remains runs job by job
  const size_t kCoupleBlocks = env_size("QS_HIP_COUPLE_BLOCKS", (size_t)200 << 10);   // (read per call: the tests lower it)
  std::vector<std::vector<int>> tasks;
  {
    std::vector<int> cur;
    size_t cur_blocks = 0;
    for (int j : small) {
      if (!job_couplable(jobs[j], flags, nit)) { tasks.push_back({j}); continue; }
      size_t b = 0;
      for (int ci = 0; ci < jobs[j]->ncomp; ++ci) b += (size_t)jobs[j]->wblk[ci] * jobs[j]->hblk[ci];
      if (!cur.empty() && (cur_blocks + b > kCoupleBlocks || 2 * (cur.size() + 1) > (size_t)QS_MAX_PLANES)) {
        tasks.push_back(cur); cur.clear(); cur_blocks = 0;
      }
      cur.push_back(j); cur_blocks += b;
    }
    if (!cur.empty()) tasks.push_back(cur);
  }
  std::atomic<size_t> next_small{0};
  std::atomic<int> first_error{0};
  auto run_small = [&]() {
    for (size_t n; (n = next_small.fetch_add(1)) < tasks.size();) {
      const std::vector<int>& task = tasks[n];
      try {
        if (task.size() == 1) {
          results[task[0]] = do_quantsmooth_impl(jobs[task[0]], flags, niter, 0, nullptr, nullptr);
        } else {
          for (int j : task) results[j] = QS_HIP_ENODEV;
          if (int r = run_coupled(jobs, task, flags, nit, results)) { int z = 0; first_error.compare_exchange_strong(z, r); }
        }
      } catch (const std::bad_alloc&) {
        for (int j : task) if (task.size() == 1 || results[j] == QS_HIP_ENODEV) results[j] = QS_HIP_ENOMEM;
      } catch (...) {
        for (int j : task) if (task.size() == 1) results[j] = QS_HIP_ENODEV;
      }
    }
  };
  auto device_worker = [&](size_t d, bool with_fused) {
    (void)hipSetDevice(devs[d]);                              // (a new thread starts on device 0)
    try {
      if (with_fused && !fused_of[d].empty())
        if (int r = run_fused(jobs, fused_of[d], flags, nit, results)) { int z = 0; first_error.compare_exchange_strong(z, r); }
    } catch (const std::bad_alloc&) {
      int z = 0; first_error.compare_exchange_strong(z, (int)QS_HIP_ENOMEM);
    } catch (...) {
      int z = 0; first_error.compare_exchange_strong(z, (int)QS_HIP_ENODEV);
    }
    run_small();
  };
  const double t0 = wall_ms();
  {
    // the caller's current device is put back on every path
    struct Restore { int dev; ~Restore() { (void)hipSetDevice(dev); } } restore{current_device()};
    if (nd == 1 && tasks.size() < 2) {
      device_worker(0, true);                                 // the common case: everything on the calling thread
    } else {
      const size_t extra = tasks.size() >= 2 ? std::min<size_t>(3, tasks.size() - 1) : 0;   // more threads for the coupled jobs
      std::vector<std::thread> pool;
      pool.reserve(nd + nd * extra);                          // no reallocation (bad_alloc) once threads are running
      size_t started = 1;                                     // devices [0, started) have their plane-set worker
      try {
        for (size_t d = 1; d < nd; ++d, ++started) pool.emplace_back(device_worker, d, true);
        for (size_t d = 0; d < nd; ++d)
          for (size_t t = 0; t < extra; ++t) pool.emplace_back(device_worker, d, false);
      } catch (const std::system_error&) {                    // no more threads: the ones we have finish the work
      }
      device_worker(0, true);
      for (size_t d = started; d < nd; ++d) device_worker(d, true);   // (devices whose thread could not start)
      for (auto& t : pool) t.join();
    }
  }
  if (trace_on()) fprintf(stderr, "qs_hip trace: batch  %zu plane-set job(s) + %zu other job(s) in %zu task(s) on %zu device(s): %.2f ms\n",
                          fused.size(), small.size(), tasks.size(), nd, wall_ms() - t0);
  // (the jobs spread over several GPUs run whatever happened above: results[] stays truthful for every job)
  for (int j : large) results[j] = do_quantsmooth_impl(jobs[j], flags, niter, 0, nullptr, nullptr);
  if (first_error.load()) return qs_fail(first_error.load(), "qs_hip_do_quantsmooth_batch: a device worker failed (results[] carries the per-job codes)");
  return QS_HIP_OK;
}

extern "C" int qs_hip_do_quantsmooth_batch(qs_hip_job* const* jobs, int njobs, int flags, int niter, int* results) {
  struct Kick { ~Kick() { kick_background(); } } kick;
  try {
    return do_quantsmooth_batch_impl(jobs, njobs, flags, niter, results);
  } catch (const std::bad_alloc&) {
    return qs_fail(QS_HIP_ENOMEM, "out of host memory");
  } catch (...) {
    return qs_fail(QS_HIP_ENODEV, "unexpected internal error");
  }
}
