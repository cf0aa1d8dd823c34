// qs_job.cpp -- job layer of the flat C ABI (include/jpegqs_hip.h): the host-side
// semantics of the reference's plane driver (reference quantsmooth.h:2404-2878) --
// validation, early-outs, iteration loop with progress/cancel between launches, final
// clamp, quant tables := 1 -- with the per-plane passes running as gfx950 kernels.
// No CPU compute fallback exists.
#include <list>
#include <chrono>
#include <system_error>

#include "qs_jobint.h"

// ---------------------------------------------------------------------------
// job layer
//
// Two execution modes share one component routine:
//  * careful  -- the reference's order: one component after the other, host
//                sync after the first pass A of each (bad-coefficient stop,
//                reference :2610) and after every iteration that reports
//                progress.  Used whenever a progress callback is installed, and
//                as the re-run path below.
//  * eager    -- no callback: every component is enqueued without host syncs,
//                independent components on their own HIP streams ("one
//                component per stream", BASELINE config 1; chroma waits for
//                luma through an event when JOINT_YUV/UPSAMPLE_UV couple them).
//                The range-check flags are read once at the end; nothing is
//                copied back before that.  If any flag is set (crafted or
//                damaged file) the job is simply re-run in careful mode from
//                the untouched host input, which reproduces the reference's
//                stop semantics exactly.
// Device buffers come from a small process-wide cache (hipMalloc/hipFree of
// 100+ MiB cost milliseconds each); qs_hip_release_cache() empties it.

using namespace qsx;
using namespace qsj;

namespace {

struct Comp {            // per-component device state (kept until the job ends)
  DevBuf coef, plane, cst, status, up, px;
  PinnedBuf stage;           // pinned upload staging, held until the job's streams are drained
  PinnedBuf hstatus;         // range-check flag on its way back
  Download down, down_up;    // results on their way back
  bool processed = false, dequant_only = false, have_up = false;
  hipStream_t stream = nullptr;
};

}  // namespace

static thread_local int16_t* const* const* tl_rows = nullptr;
qsj::RowScope::RowScope(int16_t* const* const* rows) { tl_rows = rows; }
qsj::RowScope::~RowScope() { tl_rows = nullptr; }
bool qsj::rows_active() { return tl_rows != nullptr; }

void qsj::host_pieces(const qs_hip_job* job, int ci, int row0, int nrows, size_t arena_off, std::vector<Piece>& out) {
  const size_t rowbytes = (size_t)job->wblk[ci] * 128;
  if (nrows <= 0) return;
  if (!tl_rows) {
    out.push_back({job->coef[ci] + (size_t)row0 * job->wblk[ci] * 64, arena_off, (size_t)nrows * rowbytes});
    return;
  }
  int16_t* const* rows = tl_rows[ci];
  for (int y = 0; y < nrows;) {                              // runs of rows that are adjacent in memory
    int e = y + 1;
    while (e < nrows && reinterpret_cast<char*>(rows[row0 + e]) == reinterpret_cast<char*>(rows[row0 + e - 1]) + rowbytes) ++e;
    out.push_back({rows[row0 + y], arena_off + (size_t)y * rowbytes, (size_t)(e - y) * rowbytes});
    y = e;
  }
}

double qsj::wall_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
bool qsj::trace_on() { static const bool on = getenv("QS_HIP_TRACE") != nullptr; return on; }

int qsj::run_job(qs_hip_job* job, int flags, int niter, int progprec,
                 qs_hip_progress_fn progress, void* userdata, bool eager) {
  int stop = 0;
  const int need_lowres = job_needs_lowres(job, flags);  // reference :2447-2453

  // streams/events are pooled too (creating three streams costs ~1 ms)
  StreamLease lease;
  if (!lease.p) return qs_fail(QS_HIP_ENODEV, "could not create HIP streams: %s", hipGetErrorString(hipGetLastError()));
  Streams& st = *lease.p;
  const int nstreams = eager ? 3 : 1;

  int prog_next = 0, prog_max = 0, prog_thr = 0;
  if (progress) {                                        // reference :2474-2482
    for (int ci = 0; ci < job->ncomp; ++ci) prog_max += job->hblk[ci] * job->vsamp[ci] * niter;
    if (progprec == 0) progprec = 20;
    if (progprec < 0) progprec = prog_max;
    prog_thr = (int)((unsigned)(prog_max + progprec - 1) / (unsigned)progprec);
  }

  QsConsts* hc = new (std::nothrow) QsConsts[QS_HIP_MAXC];   // one per component: uploads are async
  if (!hc) return qs_fail(QS_HIP_ENOMEM, "out of host memory");
  struct HcFree { QsConsts* p; ~HcFree() { delete[] p; } } hc_free{hc};

  const double t_start = wall_ms();
  double t_upload = 0;
  Comp comp[QS_HIP_MAXC];
  // planes that outlive their component (reference image1 / image2, :2753-2815)
  DevBuf d_yfull, d_llow;          // full-res luma plane; luma at chroma resolution
  bool have_yfull = false, have_llow = false;
  int16_t* up_host[2] = { nullptr, nullptr };
  struct UpFree { int16_t** p; bool keep; ~UpFree() { if (!keep) { qs_hip_free(p[0]); qs_hip_free(p[1]); } } } up_free{up_host, false};
  // declared after every buffer above: on any early return the streams are drained first,
  // only then do the buffers go back to the shared pools (another thread may take them at once)
  DrainGuard drain{lease.p};

  for (int ci = 0; ci < job->ncomp; ++ci) {
    Comp& C = comp[ci];
    const int wb = job->wblk[ci], hb = job->hblk[ci];
    const size_t nblk = (size_t)wb * hb, cbytes = nblk * 64 * sizeof(int16_t);
    int iters = niter, extra = 0;
    int prog_cur = prog_next;
    const int prog_inc = job->vsamp[ci];
    const int luma = !ci || job->colorspace != 3;        // reference :2639
    prog_next += hb * prog_inc * niter;
    if (!job->has_quant[ci]) continue;                   // reference :2493
    if (have_yfull || (!ci && need_lowres)) extra = 1;   // reference :2495

    int acc = 0;
    for (int i = 0; i < 64; ++i) acc |= job->quant[ci][i];
    if (acc <= 1) iters = 0;                             // reference :2501
    if (acc >= 0x800) stop = 1;                          // reference :2504
    if (iters + extra == 0) continue;                    // reference :2542

    // stream: luma (and anything coupled to it) on stream 0; independent
    // components round-robin
    hipStream_t s = st.s[eager ? ci % nstreams : 0];
    C.stream = s; C.processed = true;
    HIP_TRY(C.coef.alloc(cbytes));
    HIP_TRY(C.cst.alloc(sizeof(QsConsts)));
    HIP_TRY(C.status.alloc(sizeof(int32_t)));
    if (int r = qs_hip_consts_build(&hc[ci], job->quant[ci], flags)) return r;
    HIP_TRY(hipMemcpyAsync(C.cst.p, &hc[ci], sizeof(QsConsts), hipMemcpyHostToDevice, s));
    {
      const double t0 = wall_ms();
      std::vector<Piece> src;
      host_pieces(job, ci, 0, hb, 0, src);
      HIP_TRY(upload_pieces(C.coef.p, src, cbytes, s, C.stage));
      t_upload += wall_ms() - t0;
    }
    HIP_TRY(hipMemsetAsync(C.status.p, 0, sizeof(int32_t), s));

    bool have_plane = false;
    if (!stop) {
      // the reference falls back to dequantise-only when the plane cannot be
      // allocated (reference :2551-2566); same here for device memory
      hipError_t e = C.plane.alloc(qs_hip_plane_bytes(wb, hb));
      if (e == hipSuccess) have_plane = true; else (void)hipGetLastError();
    }
    if (!have_plane) {
      C.dequant_only = true;
      if (int r = qs_hip_dequant_plane(C.cst.p, C.coef.as<int16_t>(), wb, hb, s)) return r;
      continue;
    }
    if (eager && ci > 0 && (have_llow || have_yfull))    // chroma reads planes produced on the luma stream
      HIP_TRY(hipStreamWaitEvent(s, st.luma_done, 0));

    const int rebalance = !(flags & QS_NO_REBALANCE) && (luma || !(flags & QS_NO_REBALANCE_UV));  // :1567-1568
    // JOINT_YUV acts through the low-res luma plane only (reference :2636)
    const bool joint = have_llow && (flags & QS_JOINT_YUV);
    const int plane_flags = flags & (QS_DIAGONALS | QS_NO_REBALANCE | QS_NO_REBALANCE_UV);
    bool clamped = false;
    for (int it = 0; it < iters + extra; ++it) {
      if (int r = qs_hip_idct_plane(C.cst.p, C.coef.as<int16_t>(), C.plane.as<uint8_t>(), wb, hb,
                                    it == 0, 1, 1, C.status.as<int32_t>(), s)) return r;
      if (it == 0 && !eager) {                           // reference :2610
        int32_t bad = 0;
        HIP_TRY(hipMemcpyAsync(&bad, C.status.p, sizeof(bad), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (bad) { stop = 1; break; }
      }
      if (it == iters) break;                            // refresh-only pass, reference :2622
      // pass B.  The +-1023 clamp rides on the last launch of the last iteration --
      // unless a refresh-only pass A follows: the reference clamps after its loop
      // (:2668-2689), so that refresh (the planes JOINT_YUV / UPSAMPLE_UV read) is the
      // IDCT of the unclamped coefficients.
      const int last = (it == iters - 1) && !extra;
      if (flags & QS_LOW_QUALITY) {                      // reference :924-938: never reaches the k-loop
        if (joint) {
          if (int r = qs_hip_joint_plane(C.cst.p, C.coef.as<int16_t>(), C.plane.as<uint8_t>(), d_llow.as<uint8_t>(),
                                         wb, hb, rebalance, last, s)) return r;
        } else {
          if (int r = qs_hip_lowq_plane(C.cst.p, C.coef.as<int16_t>(), C.plane.as<uint8_t>(), wb, hb,
                                        rebalance, last, s)) return r;
        }
      } else {
        if (joint)
          if (int r = qs_hip_joint_plane(C.cst.p, C.coef.as<int16_t>(), C.plane.as<uint8_t>(), d_llow.as<uint8_t>(),
                                         wb, hb, 0, 0, s)) return r;
        if (int r = qs_hip_smooth_plane(C.cst.p, C.coef.as<int16_t>(), C.plane.as<uint8_t>(), wb, hb,
                                        plane_flags, luma, last, s)) return r;
      }
      if (last) clamped = true;
      if (progress) {                                    // reference :2656-2664
        int cur = prog_cur += hb * prog_inc;
        if (cur >= prog_thr) {
          cur = (int)((long long)progprec * cur / prog_max);
          prog_thr = (int)(((long long)(cur + 1) * prog_max + progprec - 1) / progprec);
          HIP_TRY(hipStreamSynchronize(s));              // the pass is done when we report it
          stop = progress(userdata, cur, progprec);
        }
        if (stop) break;
      }
    }
    if (!clamped)                                        // reference :2668-2689
      if (int r = qs_hip_clamp_plane(C.coef.as<int16_t>(), wb, hb, s)) return r;

    if (!stop && have_yfull) {
      // UPSAMPLE_UV: chroma -> luma resolution, re-encoded (reference :2691-2752)
      const int ws = job->hsamp[0], hs = job->vsamp[0];
      const int uwb = job->wblk[0], uhb = job->hblk[0];
      const size_t ubytes = (size_t)uwb * uhb * 64 * sizeof(int16_t);
      HIP_TRY(C.px.alloc(qs_hip_upsample_bytes(job->image_width, job->image_height, ws, hs)));
      HIP_TRY(C.up.alloc(ubytes));
      if (int r = qs_hip_upsample_plane(C.plane.as<uint8_t>(), d_llow.as<uint8_t>(), wb, d_yfull.as<uint8_t>(),
                                        uwb, uhb, C.px.as<uint8_t>(), job->image_width, job->image_height,
                                        ws, hs, s)) return r;
      if (int r = qs_hip_fdct_plane(C.px.as<uint8_t>(), qs_hip_upsample_pitch(job->image_width, ws),
                                    C.up.as<int16_t>(), uwb, uhb, s)) return r;
      C.have_up = true;
    } else if (!stop && !ci && need_lowres) {
      // keep luma for the chroma passes (reference :2753-2815)
      const int ws = job->hsamp[0], hs = job->vsamp[0];
      if (ws == 1 && hs == 1) {
        d_llow.take(C.plane); have_llow = true;          // image2 = image
      } else {
        DevBuf d_l;
        HIP_TRY(d_l.alloc(qs_hip_plane_bytes(job->wblk[1], job->hblk[1])));
        if (int r = qs_hip_downsample_plane(C.plane.as<uint8_t>(), wb, hb, d_l.as<uint8_t>(),
                                            job->wblk[1], job->hblk[1], ws, hs, s)) return r;
        d_llow.take(d_l); have_llow = true;
        if (flags & QS_UPSAMPLE_UV) { d_yfull.take(C.plane); have_yfull = true; }   // image1 = image
      }
      HIP_TRY(hipEventRecord(st.luma_done, s));
    }
    if (!eager) HIP_TRY(hipStreamSynchronize(s));
  }

  // ---- behind each component's kernels: range-check flag and results into pinned memory
  const size_t ubytes = (size_t)job->wblk[0] * job->hblk[0] * 64 * sizeof(int16_t);
  for (int ci = 0; ci < job->ncomp; ++ci) {
    Comp& C = comp[ci];
    if (!C.processed) continue;
    const size_t cbytes = (size_t)job->wblk[ci] * job->hblk[ci] * 64 * sizeof(int16_t);
    if (eager && !C.dequant_only) {
      if (!C.hstatus.alloc(sizeof(int32_t))) return qs_fail(QS_HIP_ENOMEM, "out of pinned host memory");
      HIP_TRY(hipMemcpyAsync(C.hstatus.p, C.status.p, sizeof(int32_t), hipMemcpyDeviceToHost, C.stream));
    }
    HIP_TRY(C.down.issue(C.coef.p, cbytes, C.stream, rows_active()));
    if (C.have_up && !stop) HIP_TRY(C.down_up.issue(C.up.p, ubytes, C.stream));
  }

  // ---- everything is enqueued; eager mode reads the range-check flags now
  const double t_enq = wall_ms();
  for (int i = 0; i < nstreams; ++i) HIP_TRY(hipStreamSynchronize(st.s[i]));
  const double t_done = wall_ms();
  if (eager)
    for (int ci = 0; ci < job->ncomp; ++ci)
      if (comp[ci].processed && !comp[ci].dequant_only && *static_cast<const int32_t*>(comp[ci].hstatus.p))
        return JOB_RERUN_CAREFUL;                          // host input is still untouched

  // ---- scatter the results (the only place host memory is written).  Should a transfer fail after
  // earlier components have been written, their original blocks are put back from the pinned upload
  // staging: a reported failure leaves the image untouched.
  auto scatter = [&]() -> int {
    for (int ci = 0; ci < job->ncomp; ++ci) {
      Comp& C = comp[ci];
      if (!C.processed) continue;
      std::vector<Piece> dst;
      host_pieces(job, ci, 0, job->hblk[ci], 0, dst);
      HIP_TRY(C.down.finish(C.coef.p, dst, C.stream));
      if (C.have_up && !stop) {
        if (C.down_up.staged) {
          // the replacement array IS the pinned download buffer: it changes owner (qs_hip_free gives it
          // back to the pool) instead of being copied into fresh, page-faulting malloc memory
          HIP_TRY(C.down_up.finish(C.up.p, std::vector<Piece>{}, C.stream));     // (waits for its chunks)
          up_host[ci - 1] = static_cast<int16_t*>(pinned_handout(C.down_up.stage));
        } else {
          up_host[ci - 1] = static_cast<int16_t*>(malloc(ubytes));
          if (!up_host[ci - 1]) return qs_fail(QS_HIP_ENOMEM, "out of host memory");
          HIP_TRY(C.down_up.finish(C.up.p, std::vector<Piece>{{up_host[ci - 1], 0, ubytes}}, C.stream));
        }
      }
    }
    return QS_HIP_OK;
  };
  if (int r = scatter()) {
    for (int i = 0; i < nstreams; ++i) (void)hipStreamSynchronize(st.s[i]);
    for (int ci = 0; ci < job->ncomp; ++ci) {
      Comp& C = comp[ci];
      if (!C.processed || !C.stage.p) continue;
      std::vector<Piece> pcs;
      host_pieces(job, ci, 0, job->hblk[ci], 0, pcs);
      for (const Piece& pc : pcs) memcpy(pc.host, static_cast<const char*>(C.stage.p) + pc.off, pc.len);
    }
    return r;
  }
  if (trace_on())
    fprintf(stderr, "qs_hip trace: %s  enqueue %.2f ms (host->pinned->device issue %.2f)  drain %.2f ms  scatter %.2f ms\n",
            eager ? "eager" : "careful", t_enq - t_start, t_upload, t_done - t_enq, wall_ms() - t_done);

  if (!stop && have_yfull && up_host[0] && up_host[1]) {  // reference :2836-2849
    job->coef_up[0] = up_host[0]; job->coef_up[1] = up_host[1]; up_free.keep = true;
    job->up_wblk = job->wblk[0]; job->up_hblk = job->hblk[0];
    job->out_hsamp0 = job->out_vsamp0 = 1;
  }
  for (int ci = 0; ci < job->ncomp; ++ci)                // reference :2851-2859
    if (job->has_quant[ci]) for (int i = 0; i < 64; ++i) job->quant[ci][i] = 1;
  return stop;
}


// ---------------------------------------------------------------------------
// fused execution: jobs whose components are independent of each other (no
// JOINT_YUV / UPSAMPLE_UV coupling, no LOW_QUALITY, ordinary quant tables) run
// as plane sets -- ONE pass-A and ONE pass-B launch per iteration for all
// components of all jobs of a group (qs_*_set_kernel), so that small images
// fill the chip together and a job does not occupy three hardware queues.
// Everything else about the job semantics is as in run_job (eager mode): the
// range-check flags are read once at the end, a job with a set flag is re-run in
// the careful order from its untouched host input.

int qsj::comp_rebalance(const qs_hip_job* job, int ci, int flags) {
  const int luma = !ci || job->colorspace != 3;                                     // reference :2639
  return !(flags & QS_NO_REBALANCE) && (luma || !(flags & QS_NO_REBALANCE_UV));       // :1567-1568
}

// The luma/chroma coupling of JOINT_YUV / UPSAMPLE_UV (reference :2447-2453).  libjpeg only ever
// reports JCS_YCbCr with exactly three components; the flat ABI could be handed four, for which the
// reference's coef_up[ci - 1] indexing has no meaning (two replacement arrays exist), so the
// coupling is tied to ncomp == 3 here.
bool qsj::job_needs_lowres(const qs_hip_job* job, int flags) {
  return (flags & (QS_JOINT_YUV | QS_UPSAMPLE_UV)) && job->colorspace == 3 && job->ncomp == 3 &&
         job->hsamp[1] == 1 && job->vsamp[1] == 1 && job->hsamp[2] == 1 && job->vsamp[2] == 1;
}

// Tuning knobs from the environment.  Read ONCE per name in a production process (getenv on every call
// would race with a host application's setenv); with QS_HIP_TEST_HOOKS=1 they are re-read on every call,
// which is how the tests lower thresholds inside one process.
size_t qsj::env_size(const char* name, size_t dflt) {
  auto read = [&]() { const char* v = getenv(name); const long long n = v ? atoll(v) : 0; return n > 0 ? (size_t)n : dflt; };
  if (test_hooks_on()) return read();
  static std::mutex mu;
  static std::vector<std::pair<std::string, size_t>> seen;
  std::lock_guard<std::mutex> lk(mu);
  for (auto& e : seen) if (e.first == name) return e.second;
  seen.emplace_back(name, read());
  return seen.back().second;
}
bool qsj::job_fusable(const qs_hip_job* job, int flags) {
  static const bool off = getenv("QS_HIP_NO_FUSE") != nullptr;
  if (off || (flags & QS_LOW_QUALITY) || job_needs_lowres(job, flags)) return false;
  for (int ci = 0; ci < job->ncomp; ++ci) {
    if (!job->has_quant[ci]) return false;
    int acc = 0;
    for (int i = 0; i < 64; ++i) acc |= job->quant[ci][i];
    if (acc <= 1 || acc >= 0x800) return false;          // iterations skipped / stop: the general path knows how
  }
  return true;
}

namespace {
// One device plane of a set: a whole component, or a band of block rows of a very large
// one (rows [src_row0, src_row0 + hb) of the source, of which [keep0, keep1) are results:
// the rest is halo, see split_rows).
struct FPlane { int job, ci, wb, hb, cst; size_t coef_off, px_off, cbytes; int src_row0, keep0, keep1; };
struct FGroup {
  std::vector<FPlane> planes;
  std::vector<int> jobs;                  // indices into the caller's job list
  DevBuf coef, px, cst, status;
  PinnedBuf stage;
  std::vector<QsConsts> hc;               // host copies stay alive until the stream is drained
  PinnedBuf hstatus;                      // range-check flags
  Download down;                          // results on their way back
  hipStream_t s = nullptr;
  size_t blocks = 0, coef_bytes = 0;
  // everything queued on the group's stream has completed: give the arenas back
  void release_transients(bool keep_stage) {
    coef.release(); px.release(); cst.release(); status.release();
    hstatus.release(); down.reset();
    if (!keep_stage) stage.release();
  }
};

// a group of >= 3 waves per SIMD runs at the streaming rate; smaller groups let the upload of one
// overlap the kernels of the previous and the download of the one before (three streams)
static const size_t kGroupBlocks = (size_t)200 << 10;
// A plane above kSplitBlocks is cut into bands of about kBandBlocks that travel as separate
// groups, so its upload, kernels and download overlap as they do for a batch of small jobs.
// A block's result after n iterations depends only on blocks within n rows of it, so a band
// carries n extra block rows on each cut side (recomputed, not copied back): bit-exact.
// (QS_HIP_SPLIT_BLOCKS / QS_HIP_BAND_BLOCKS override the two sizes: the tests use them to run
// the band logic on small images.)
static const size_t kSplitBlocks = env_size("QS_HIP_SPLIT_BLOCKS", (size_t)512 << 10),
                    kBandBlocks = env_size("QS_HIP_BAND_BLOCKS", (size_t)256 << 10);

static int run_fused(qs_hip_job* const* jobs, const std::vector<int>& which, int flags, int niter, int* results) {
  StreamLease lease;
  if (!lease.p) return qs_fail(QS_HIP_ENODEV, "could not create HIP streams: %s", hipGetErrorString(hipGetLastError()));
  std::list<FGroup> groups;
  DrainGuard drain{lease.p};
  const double t_start = wall_ms();
  double t_enq = t_start;

  // ---- partition into groups (a job never straddles two, unless it is cut into bands)
  int maxj = 0;
  for (int ji : which) maxj = std::max(maxj, ji);
  std::vector<char> split(maxj + 1, 0), bad_job(maxj + 1, 0), scattered(maxj + 1, 0), defer(maxj + 1, 0);
  std::vector<int> ngroups(maxj + 1, 0), ndone(maxj + 1, 0);   // groups a job's planes live in / groups whose results are back
  for (int ji : which) {
    const qs_hip_job* job = jobs[ji];
    size_t jblocks = 0;
    bool big = false;
    for (int ci = 0; ci < job->ncomp; ++ci) {
      const size_t nb = (size_t)job->wblk[ci] * job->hblk[ci];
      jblocks += nb;
      const int bands = (int)((nb + kBandBlocks - 1) / kBandBlocks);
      if (nb > kSplitBlocks && (job->hblk[ci] + bands - 1) / bands >= 8 * niter) big = true;   // halo <= 25 %
    }
    if (big) {
      split[ji] = 1;
      for (int ci = 0; ci < job->ncomp; ++ci) {
        const int wb = job->wblk[ci], hb = job->hblk[ci];
        const int bands = std::max(1, (int)(((size_t)wb * hb + kBandBlocks - 1) / kBandBlocks));
        const int rows = (hb + bands - 1) / bands;
        for (int r0 = 0; r0 < hb; r0 += rows) {
          const int r1 = std::min(hb, r0 + rows), d0 = std::max(0, r0 - niter), d1 = std::min(hb, r1 + niter);
          groups.emplace_back();
          FGroup& G = groups.back();
          G.jobs.push_back(ji);
          G.blocks = (size_t)wb * (d1 - d0);
          G.planes.push_back({ji, ci, wb, d1 - d0, -1, 0, 0, (size_t)wb * (d1 - d0) * 128, d0, r0 - d0, r1 - d0});
        }
      }
      groups.emplace_back();                                 // the next job starts a fresh group
      continue;
    }
    if (groups.empty() || (int)groups.back().planes.size() + job->ncomp > QS_MAX_PLANES ||
        (groups.back().blocks && groups.back().blocks + jblocks > kGroupBlocks))
      groups.emplace_back();
    FGroup& G = groups.back();
    G.jobs.push_back(ji);
    G.blocks += jblocks;
    for (int ci = 0; ci < job->ncomp; ++ci)
      G.planes.push_back({ji, ci, job->wblk[ci], job->hblk[ci], -1, 0, 0, (size_t)job->wblk[ci] * job->hblk[ci] * 128,
                          0, 0, job->hblk[ci]});
  }
  groups.remove_if([](const FGroup& g) { return g.planes.empty(); });   // placeholders left by band jobs
  for (const FGroup& G : groups) for (int ji : G.jobs) ++ngroups[ji];

  // ---- per group: upload, niter x (pass A, pass B), status readback, download into pinned memory
  const int diag = (flags & QS_DIAGONALS) != 0;
  size_t gi = 0;
  auto enqueue = [&](FGroup& G) -> int {
    G.s = lease.p->s[gi++ % 3];
    const int np = (int)G.planes.size();
    size_t coef_bytes = 0, px_bytes = 0;
    std::vector<const uint16_t*> qtabs;
    for (FPlane& P : G.planes) {
      P.coef_off = coef_bytes; coef_bytes += P.cbytes;
      P.px_off = px_bytes; px_bytes += (qs_hip_plane_bytes(P.wb, P.hb) + 255) & ~(size_t)255;
      const uint16_t* q = jobs[P.job]->quant[P.ci];
      for (size_t k = 0; k < qtabs.size() && P.cst < 0; ++k)
        if (!memcmp(qtabs[k], q, 64 * sizeof(uint16_t))) P.cst = (int)k;
      if (P.cst < 0) { P.cst = (int)qtabs.size(); qtabs.push_back(q); }
    }
    HIP_TRY(G.coef.alloc(coef_bytes));
    HIP_TRY(G.px.alloc(px_bytes));
    HIP_TRY(G.cst.alloc(qtabs.size() * sizeof(QsConsts)));
    HIP_TRY(G.status.alloc((size_t)np * sizeof(int32_t)));
    G.hc.resize(qtabs.size());
    for (size_t k = 0; k < qtabs.size(); ++k)
      if (int r = qs_hip_consts_build(&G.hc[k], qtabs[k], flags)) return r;
    HIP_TRY(hipMemcpyAsync(G.cst.p, G.hc.data(), qtabs.size() * sizeof(QsConsts), hipMemcpyHostToDevice, G.s));
    std::vector<Piece> pieces;
    for (const FPlane& P : G.planes)
      host_pieces(jobs[P.job], P.ci, P.src_row0, P.hb, P.coef_off, pieces);
    G.coef_bytes = coef_bytes;
    HIP_TRY(upload_pieces(G.coef.p, pieces, coef_bytes, G.s, G.stage));
    HIP_TRY(hipMemsetAsync(G.status.p, 0, (size_t)np * sizeof(int32_t), G.s));

    QsPlaneSet set;
    memset(&set, 0, sizeof set);
    set.n = np;
    int w = 0;
    for (int i = 0; i < np; ++i) {
      const FPlane& P = G.planes[i];
      set.wave0[i] = w;
      w += (P.wb * P.hb + 63) / 64;
      QsPlaneRef& R = set.ref[i];
      R.cst = G.cst.as<QsConsts>() + P.cst;
      R.coef = reinterpret_cast<int16_t*>(G.coef.as<char>() + P.coef_off);
      R.plane = G.px.as<uint8_t>() + P.px_off;
      R.status = G.status.as<int32_t>() + i;
      R.wblk = P.wb; R.hblk = P.hb; R.pitch = qs_plane_pitch(P.wb);
      R.mode = QS_PLANE_REP_TOP | QS_PLANE_REP_BOT | (comp_rebalance(jobs[P.job], P.ci, flags) ? QS_PLANE_REBALANCE : 0);
    }
    for (int i = np; i < QS_MAX_PLANES + 2; ++i) set.wave0[i] = w;
    for (int it = 0; it < niter; ++it) {
      qs_launch_idct_set(set, it == 0, G.s);
      qs_launch_smooth_set(set, diag, it == niter - 1, G.s);
    }
    HIP_TRY(hipGetLastError());
    // pinned: a pageable destination would make this call wait for the whole stream
    if (!G.hstatus.alloc((size_t)np * sizeof(int32_t))) return qs_fail(QS_HIP_ENOMEM, "out of pinned host memory");
    HIP_TRY(hipMemcpyAsync(G.hstatus.p, G.status.p, (size_t)np * sizeof(int32_t), hipMemcpyDeviceToHost, G.s));
    HIP_TRY(G.down.issue(G.coef.p, coef_bytes, G.s, rows_active()));       // to pinned memory, right behind the kernels
    // a band job is scattered band by band; without the staging copy of its input (pinned memory
    // exhausted) nothing could be restored should a later band trip the range check: hold it back
    if (!G.stage.p) for (int ji : G.jobs) if (split[ji]) defer[ji] = 1;
    return QS_HIP_OK;
  };

  // ---- drain a group; results go back only for jobs whose range check passed.
  // A job cut into bands is scattered band by band before its later bands have been
  // checked: should one of those trip the range check after all (crafted file), the rows
  // already written are restored from the pinned upload staging, which still holds the
  // original input.  Without that staging copy the job's bands are held back until all of
  // them have been checked.
  auto result_pieces = [&](const FPlane& P, std::vector<Piece>& out) {   // the rows of P that are results (not halo)
    host_pieces(jobs[P.job], P.ci, P.src_row0 + P.keep0, P.keep1 - P.keep0, P.coef_off + (size_t)P.keep0 * P.wb * 128, out);
  };
  std::vector<FGroup*> held;
  auto drain_group = [&](FGroup& G) -> int {
    HIP_TRY(G.down.wait_first(G.s));
    const int32_t* hst = static_cast<const int32_t*>(G.hstatus.p);
    for (size_t i = 0; i < G.planes.size(); ++i) if (hst[i]) bad_job[G.planes[i].job] = 1;
    bool hold = false, banded = false;
    for (int ji : G.jobs) { hold |= (defer[ji] != 0); banded |= (split[ji] != 0); }
    if (hold) { held.push_back(&G); return QS_HIP_OK; }
    std::vector<Piece> back;
    for (const FPlane& P : G.planes)
      if (!bad_job[P.job]) { result_pieces(P, back); scattered[P.job] = 1; }
    HIP_TRY(G.down.finish(G.coef.p, back, G.s));
    for (int ji : G.jobs) ++ndone[ji];
    // the group's stream work is complete: recycle its device arenas and download staging now, so
    // that memory in flight is bounded by the window below and not by the size of the batch.  The
    // upload staging of a band job stays (it is the restore copy): that is one image's worth.
    G.release_transients(/*keep_stage=*/banded);
    return QS_HIP_OK;
  };

  // At most kWindow groups are in flight (about 200k blocks each: ~40 MiB of device memory and
  // ~50 MiB of pinned staging per group).
  static const size_t kWindow = env_size("QS_HIP_GROUP_WINDOW", 6);
  auto pump = [&]() -> int {
    std::deque<FGroup*> inflight;
    for (FGroup& G : groups) {
      if (inflight.size() >= kWindow) {
        if (int r = drain_group(*inflight.front())) return r;
        inflight.pop_front();
      }
      if (int r = enqueue(G)) return r;
      inflight.push_back(&G);
    }
    t_enq = wall_ms();
    for (FGroup* G : inflight)
      if (int r = drain_group(*G)) return r;
    for (FGroup* G : held) {
      std::vector<Piece> back;
      for (const FPlane& P : G->planes) if (!bad_job[P.job]) result_pieces(P, back);
      HIP_TRY(G->down.finish(G->coef.p, back, G->s));
      for (int ji : G->jobs) ++ndone[ji];
    }
    return QS_HIP_OK;
  };
  // the caller's rows of job ji <- the original input kept in the pinned upload staging
  auto restore_job = [&](int ji) {
    for (FGroup& G : groups)
      for (const FPlane& P : G.planes)
        if (P.job == ji && G.stage.p) {
          std::vector<Piece> pcs;
          result_pieces(P, pcs);
          for (const Piece& pc : pcs) memcpy(pc.host, static_cast<const char*>(G.stage.p) + pc.off, pc.len);
        }
  };
  if (int r = pump()) {
    // Error exit (device out of memory, HIP failure) with groups in flight.  A banded job is scattered
    // band by band, so some of its rows may already hold results while the call reports a failure:
    // "image left untouched" must hold for callers that ignore the return value, as the reference's
    // applications do.  Wait for everything queued, then either finish a job whose every group came
    // back (its result is complete and checked) or put the original rows back.
    for (auto& x : lease.p->s) (void)hipStreamSynchronize(x);
    for (int ji : which) {
      if (!scattered[ji]) continue;
      if (!bad_job[ji] && ndone[ji] == ngroups[ji]) {
        results[ji] = 0;
        for (int ci = 0; ci < jobs[ji]->ncomp; ++ci)
          for (int i = 0; i < 64; ++i) jobs[ji]->quant[ci][i] = 1;
      } else {
        restore_job(ji);
      }
    }
    return r;
  }
  std::vector<int> rerun;
  for (int ji : which) {
    if (!bad_job[ji]) { results[ji] = 0; continue; }
    rerun.push_back(ji);
    if (scattered[ji]) restore_job(ji);                      // (otherwise the host input is still untouched)
  }
  if (trace_on())
    fprintf(stderr, "qs_hip trace: fused  %zu job(s) in %zu group(s)  enqueue %.2f ms  drain+download %.2f ms  (%zu re-run)\n",
            which.size(), groups.size(), t_enq - t_start, wall_ms() - t_enq, rerun.size());
  for (int ji : which) {
    if (bad_job[ji]) continue;
    for (int ci = 0; ci < jobs[ji]->ncomp; ++ci)           // reference :2851-2859
      for (int i = 0; i < 64; ++i) jobs[ji]->quant[ci][i] = 1;
  }
  const double t_clear = wall_ms();
  groups.clear();                                            // give the arenas back before the re-runs allocate
  if (trace_on()) fprintf(stderr, "qs_hip trace: fused  release %.2f ms\n", wall_ms() - t_clear);
  for (int ji : rerun)
    results[ji] = run_job(jobs[ji], flags, niter, 0, nullptr, nullptr, /*eager=*/false);
  return QS_HIP_OK;
}


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
  size_t coef_bytes = 0, px_bytes = 0, upx_bytes = 0, upc_bytes = 0, luma_blocks = 0, chroma_blocks = 0;
  std::vector<const uint16_t*> qtabs;
  for (int pass = 0; pass < 2; ++pass)
    for (int g = 0; g < G; ++g) {
      const qs_hip_job* job = jobs[which[g]];
      CJob& J = cj[g];
      for (int ci = pass ? 1 : 0; ci < (pass ? 3 : 1); ++ci) {
        const size_t nb = (size_t)job->wblk[ci] * job->hblk[ci];
        J.coef_off[ci] = coef_bytes; coef_bytes += nb * 128;
        (ci ? chroma_blocks : luma_blocks) += nb;
        J.px_off[ci] = px_bytes; px_bytes += (qs_hip_plane_bytes(job->wblk[ci], job->hblk[ci]) + 255) & ~(size_t)255;
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
  auto plane_of = [&](int g, int ci) { return px.as<uint8_t>() + cj[g].px_off[ci]; };
  auto lowres_of = [&](int g) { return cj[g].sub ? px.as<uint8_t>() + cj[g].l_off : plane_of(g, 0); };
  auto make_set = [&](QsPlaneSet& set, int ci0, int ci1) {
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
        R.status = status.as<int32_t>() + g * 3 + ci;
        R.wblk = job->wblk[ci]; R.hblk = job->hblk[ci]; R.pitch = qs_plane_pitch(job->wblk[ci]);
        R.mode = QS_PLANE_REP_TOP | QS_PLANE_REP_BOT | (comp_rebalance(job, ci, flags) ? QS_PLANE_REBALANCE : 0);
      }
    set.n = n;
    for (int i = n; i < QS_MAX_PLANES + 2; ++i) set.wave0[i] = w;
  };
  QsPlaneSet set;

  // ---- luma: niter iterations, then the refresh pass the chroma stages read (reference :2495, :2622).
  // The +-1023 clamp comes after that refresh (reference :2668-2689 sits behind the loop).
  make_set(set, 0, 1);
  for (int it = 0; it < niter; ++it) {
    qs_launch_idct_set(set, it == 0, s);
    qs_launch_smooth_set(set, diag, 0, s);
  }
  qs_launch_idct_set(set, 0, s);
  qs_launch_clamp(coef_of(0, 0), luma_blocks, s);
  for (int g = 0; g < G; ++g) {                              // image2 (reference :2753-2815)
    const qs_hip_job* job = jobs[which[g]];
    if (cj[g].sub)
      qs_launch_downsample(plane_of(g, 0), job->wblk[0], job->hblk[0], lowres_of(g), job->wblk[1], job->hblk[1],
                           job->hsamp[0], job->vsamp[0], s);
  }

  // ---- chroma.  A job that is upsampled afterwards takes one more refresh pass (and its clamp moves
  // behind it); jobs of both kinds may share a group, so the extra pass runs on a set of its own.
  make_set(set, 1, 3);
  QsPlaneAux lowres;
  memset(&lowres, 0, sizeof lowres);
  for (int g = 0; g < G; ++g) lowres.p[2 * g] = lowres.p[2 * g + 1] = lowres_of(g);
  bool any_up = false, all_up = true;
  for (int g = 0; g < G; ++g) { any_up |= cj[g].upsample; all_up &= cj[g].upsample; }
  for (int it = 0; it < niter; ++it) {
    qs_launch_idct_set(set, it == 0, s);
    if (joint)                                               // JOINT_YUV acts through the low-res luma plane (reference :2636)
      qs_launch_joint_set(set, lowres, 0, 0, s);
    qs_launch_smooth_set(set, diag, it == niter - 1 && !any_up, s);
  }
  if (any_up) {
    if (all_up) {
      qs_launch_idct_set(set, 0, s);
      qs_launch_clamp(coef_of(0, 1), chroma_blocks, s);
    } else {
      for (int g = 0; g < G; ++g)
        for (int ci = 1; ci < 3; ++ci) {
          const qs_hip_job* job = jobs[which[g]];
          if (cj[g].upsample)
            qs_launch_idct_plane(cst.as<QsConsts>() + cj[g].cst[ci], coef_of(g, ci), plane_of(g, ci), job->wblk[ci], job->hblk[ci],
                                 0, 1, 1, status.as<int32_t>() + g * 3 + ci, s);
        }
      qs_launch_clamp(coef_of(0, 1), chroma_blocks, s);     // (clamping is idempotent and independent of the refresh order
                                                              //  for the jobs without one)
    }
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
  if (hipError_t e = down.finish(coef.p, back, s)) {          // a late failure: put the original blocks back
    (void)hipStreamSynchronize(s);
    if (stage.p) for (const Piece& pc : back) memcpy(pc.host, static_cast<const char*>(stage.p) + pc.off, pc.len);
    return qs_fail(QS_HIP_ENODEV, "download failed: %s", hipGetErrorString(e));
  }
  // replacement arrays: first every transfer is completed (nothing handed out yet, so an error on the
  // way leaves no job half-updated), then ownership moves to the jobs
  struct UpArrays {
    std::vector<int16_t*> p;
    ~UpArrays() { for (int16_t* q : p) if (q) free(q); }                    // (only malloc'ed ones are kept here)
  } ups;
  ups.p.assign((size_t)G * 2, nullptr);
  for (int g = 0; g < G; ++g) {
    if (!cj[g].upsample) continue;
    const bool bad = (hst[g * 3] | hst[g * 3 + 1] | hst[g * 3 + 2]) != 0;
    for (int k = 0; k < 2; ++k) {
      Download& D = *down_up_of[(size_t)g * 2 + k];
      const char* src = upc.as<char>() + cj[g].upc_off[k];
      if (D.staged) {
        HIP_TRY(D.finish(src, std::vector<Piece>{}, s));                    // (waits for its chunks)
      } else if (!bad) {
        int16_t* q = static_cast<int16_t*>(malloc(cj[g].ubytes));
        if (!q) return qs_fail(QS_HIP_ENOMEM, "out of host memory");
        ups.p[(size_t)g * 2 + k] = q;
        HIP_TRY(D.finish(src, std::vector<Piece>{{q, 0, cj[g].ubytes}}, s));
      }
    }
  }
  for (int g = 0; g < G; ++g) {
    qs_hip_job* job = jobs[which[g]];
    if (hst[g * 3] | hst[g * 3 + 1] | hst[g * 3 + 2]) continue;
    if (cj[g].upsample) {                                    // reference :2836-2849
      for (int k = 0; k < 2; ++k) {
        Download& D = *down_up_of[(size_t)g * 2 + k];
        // a staged array IS the pinned download buffer: it changes owner (qs_hip_free gives it back to the pool)
        job->coef_up[k] = D.staged ? static_cast<int16_t*>(pinned_handout(D.stage)) : ups.p[(size_t)g * 2 + k];
        ups.p[(size_t)g * 2 + k] = nullptr;
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

}  // namespace (fused and coupled routes)

extern "C" void qs_hip_release_cache(void) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  const int cur = current_device();
  for (auto& c : g_cache) {                                  // every device's blocks, each under its own device
    if (c.dev != current_device()) (void)hipSetDevice(c.dev);
    (void)hipFree(c.p);
  }
  g_cache.clear();
  for (auto* sp : g_stream_pool) delete sp;                  // (~Streams switches to the owning device itself)
  g_stream_pool.clear();
  for (auto& c : PinnedBuf::pool()) (void)hipHostFree(c.p);
  PinnedBuf::pool().clear();
  if (cur != current_device()) (void)hipSetDevice(cur);
}

// validation and the reference's early-outs; returns 1 when there is work to do,
// 0 when the job is already finished (result 0), < 0 on a bad job
static int prepare_job(qs_hip_job* job, int flags, int* niter) {
  if (!job || job->ncomp < 1 || job->ncomp > QS_HIP_MAXC)
    return qs_fail(QS_HIP_EINVAL, "qs_hip_do_quantsmooth: bad job");
  for (int ci = 0; ci < job->ncomp; ++ci) {
    if ((!tl_rows && !job->coef[ci]) || job->wblk[ci] <= 0 || job->hblk[ci] <= 0)
      return qs_fail(QS_HIP_EINVAL, "qs_hip_do_quantsmooth: component %d has no data", ci);
    if (job->hsamp[ci] < 1 || job->hsamp[ci] > 4 || job->vsamp[ci] < 1 || job->vsamp[ci] > 4)   // JPEG: 1..4
      return qs_fail(QS_HIP_EINVAL, "qs_hip_do_quantsmooth: component %d has sampling factors %dx%d",
                     ci, job->hsamp[ci], job->vsamp[ci]);
    if ((long long)job->wblk[ci] * job->hblk[ci] > (1ll << 27))     // 65500 px / 8 squared is 2^26: int block indices are safe
      return qs_fail(QS_HIP_EINVAL, "qs_hip_do_quantsmooth: component %d is too large", ci);
  }
  job->up_wblk = job->up_hblk = 0; job->coef_up[0] = job->coef_up[1] = nullptr;
  job->out_hsamp0 = job->hsamp[0]; job->out_vsamp0 = job->vsamp[0];
  if (*niter < 0) *niter = 0;
  if (*niter > 100) *niter = 100;                          // reference :2455-2456
  if (*niter <= 0 && !((flags & QS_UPSAMPLE_UV) && job_needs_lowres(job, flags))) return 0;  // reference :2458
  return 1;
}

static int do_quantsmooth_impl(qs_hip_job* job, int flags, int niter, int progprec,
                               qs_hip_progress_fn progress, void* userdata) {
  const int todo = prepare_job(job, flags, &niter);
  if (todo <= 0) return todo;
  if (qs_hip_device_count() <= 0)
    return qs_fail(QS_HIP_ENODEV, "no HIP device available (this library has no CPU fallback)");

  if (!progress) {                                             // several GPUs and a job worth spreading over them
    const std::vector<int> devs = shard_devices_for(job, flags, niter);
    if (!devs.empty()) {
      int r = run_sharded(job, flags, niter, devs);
      if (r == JOB_RERUN_CAREFUL)
        r = run_job(job, flags, niter, progprec, progress, userdata, /*eager=*/false);
      return r;
    }
  }
  if (!progress && job_fusable(job, flags)) {
    int result = QS_HIP_ENODEV;
    qs_hip_job* one[1] = { job };
    if (int r = run_fused(one, std::vector<int>{0}, flags, niter, &result)) return r;
    return result;
  }
  int r = run_job(job, flags, niter, progprec, progress, userdata, /*eager=*/progress == nullptr);
  if (r == JOB_RERUN_CAREFUL)
    r = run_job(job, flags, niter, progprec, progress, userdata, /*eager=*/false);
  return r;
}

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

// The C ABI never lets a C++ exception (std::bad_alloc from the host-side containers) escape.
extern "C" int qs_hip_do_quantsmooth(qs_hip_job* job, int flags, int niter, int progprec,
                                     qs_hip_progress_fn progress, void* userdata) {
  try {
    return do_quantsmooth_impl(job, flags, niter, progprec, progress, userdata);
  } catch (const std::bad_alloc&) {
    return qs_fail(QS_HIP_ENOMEM, "out of host memory");
  } catch (...) {
    return qs_fail(QS_HIP_ENODEV, "unexpected internal error");
  }
}

extern "C" int qs_hip_do_quantsmooth_rows(qs_hip_job* job, int16_t* const* const* rows, int flags, int niter, int progprec,
                                          qs_hip_progress_fn progress, void* userdata) {
  try {
    if (!job || !rows || job->ncomp < 1 || job->ncomp > QS_HIP_MAXC)
      return qs_fail(QS_HIP_EINVAL, "qs_hip_do_quantsmooth_rows: bad job");
    for (int ci = 0; ci < job->ncomp; ++ci) {
      if (!rows[ci] || job->hblk[ci] <= 0) return qs_fail(QS_HIP_EINVAL, "qs_hip_do_quantsmooth_rows: component %d has no rows", ci);
      for (int y = 0; y < job->hblk[ci]; ++y)
        if (!rows[ci][y]) return qs_fail(QS_HIP_EINVAL, "qs_hip_do_quantsmooth_rows: component %d row %d is null", ci, y);
    }
    RowScope scope(rows);
    return do_quantsmooth_impl(job, flags, niter, progprec, progress, userdata);
  } catch (const std::bad_alloc&) {
    return qs_fail(QS_HIP_ENOMEM, "out of host memory");
  } catch (...) {
    return qs_fail(QS_HIP_ENODEV, "unexpected internal error");
  }
}

extern "C" int qs_hip_do_quantsmooth_sharded(qs_hip_job* job, int flags, int niter, const int* devices, int ndev) {
  try {
    if (!devices || ndev < 1) return qs_fail(QS_HIP_EINVAL, "qs_hip_do_quantsmooth_sharded: empty device list");
    const int todo = prepare_job(job, flags, &niter);
    if (todo <= 0) return todo;
    if (qs_hip_device_count() <= 0)
      return qs_fail(QS_HIP_ENODEV, "no HIP device available (this library has no CPU fallback)");
    int r = run_sharded(job, flags, niter, std::vector<int>(devices, devices + ndev));
    if (r == JOB_RERUN_CAREFUL)
      r = run_job(job, flags, niter, 0, nullptr, nullptr, /*eager=*/false);
    return r;
  } catch (const std::bad_alloc&) {
    return qs_fail(QS_HIP_ENOMEM, "out of host memory");
  } catch (...) {
    return qs_fail(QS_HIP_ENODEV, "unexpected internal error");
  }
}

extern "C" int qs_hip_do_quantsmooth_batch(qs_hip_job* const* jobs, int njobs, int flags, int niter, int* results) {
  try {
    return do_quantsmooth_batch_impl(jobs, njobs, flags, niter, results);
  } catch (const std::bad_alloc&) {
    return qs_fail(QS_HIP_ENOMEM, "out of host memory");
  } catch (...) {
    return qs_fail(QS_HIP_ENODEV, "unexpected internal error");
  }
}
