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
  DevBuf coef, plane, plane2, cst, status, up, px;   // plane2: the second pixel plane of the fused schedule (pass B writes the next iteration's)
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
    hipStream_t s = st.get_ready(eager ? ci % nstreams : 0);   // (extra streams only once they exist: see Streams::get_ready)
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
    // Fused schedule: every pass B that is followed by another pass A (the next iteration's, or the refresh-only
    // pass) writes that pass's pixel plane itself, into a second plane (C.plane and C.plane2 swap roles); the
    // stand-alone pass A then only runs for iteration 0.  Not for LOW_QUALITY (no recovery kernel), and not when the
    // second plane cannot be had (the unfused order is the fall-back, like the reference's own, :2551-2566).
    bool fuse = !(flags & QS_LOW_QUALITY) && iters + extra > 1;
    if (fuse && C.plane2.alloc(qs_hip_plane_bytes(wb, hb)) != hipSuccess) { (void)hipGetLastError(); fuse = false; }
    bool have_next = false;                              // this iteration's plane was written by the previous pass B
    for (int it = 0; it < iters + extra; ++it) {
      if (!have_next)
        if (int r = qs_hip_idct_plane(C.cst.p, C.coef.as<int16_t>(), C.plane.as<uint8_t>(), wb, hb,
                                      it == 0, 1, 1, C.status.as<int32_t>(), s)) return r;
      have_next = false;
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
        if (fuse && it + 1 < iters + extra) {
          // (the +-1023 clamp may ride on the last iteration's launch even when the refresh-only pass follows: the
          //  fused IDCT reads the unclamped coefficients the kernel holds, the clamp applies to what it stores)
          const int clamp_now = it == iters - 1;
          if (int r = qs_hip_smooth_plane_next(C.cst.p, C.coef.as<int16_t>(), C.plane.as<uint8_t>(), C.plane2.as<uint8_t>(),
                                               wb, hb, plane_flags, luma, clamp_now, 1, 1, s)) return r;
          DevBuf t; t.take(C.plane); C.plane.take(C.plane2); C.plane2.take(t);
          have_next = true;
          if (clamp_now) clamped = true;
        } else if (int r = qs_hip_smooth_plane(C.cst.p, C.coef.as<int16_t>(), C.plane.as<uint8_t>(), wb, hb,
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
  for (int i = 0; i < nstreams; ++i) if (st.s[i]) HIP_TRY(hipStreamSynchronize(st.s[i]));
  const double t_done = wall_ms();
  if (eager)
    for (int ci = 0; ci < job->ncomp; ++ci)
      if (comp[ci].processed && !comp[ci].dequant_only && *static_cast<const int32_t*>(comp[ci].hstatus.p))
        return JOB_RERUN_CAREFUL;                          // host input is still untouched

  // ---- scatter the results (the only place host memory is written).  Should a transfer fail after
  // earlier components have been written, their original blocks are put back from the pinned upload
  // staging: a reported failure leaves the image untouched.
  auto scatter = [&]() -> int {
    // without the pinned upload staging of EVERY component there is nothing to restore from: then everything
    // lands in library-owned memory first (the only step that can fail) and is copied to the caller afterwards
    bool have_copy = true;
    for (int ci = 0; ci < job->ncomp; ++ci) have_copy = have_copy && (!comp[ci].processed || comp[ci].stage.p);
    if (!have_copy)
      for (int ci = 0; ci < job->ncomp; ++ci) {
        Comp& C = comp[ci];
        if (!C.processed) continue;
        HIP_TRY(C.down.land(C.coef.p, C.stream));
        if (C.have_up && !stop) HIP_TRY(C.down_up.land(C.up.p, C.stream));
      }
    for (int ci = 0; ci < job->ncomp; ++ci) {
      Comp& C = comp[ci];
      if (!C.processed) continue;
      std::vector<Piece> dst;
      host_pieces(job, ci, 0, job->hblk[ci], 0, dst);
      HIP_TRY(C.down.finish(C.coef.p, dst, C.stream, have_copy));
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
    st.sync_all();
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
// ---- the reference's progress calls as a function of the geometry (see qs_jobint.h)
void qsj::ProgressPlan::init(const qs_hip_job* job, int niter, int progprec_arg, qs_hip_progress_fn f, void* ud) {
  fn = f; userdata = ud; progprec = progprec_arg;
  calls.clear(); made = 0; cancelled = false; replayed = 0;
  int prog_max = 0;
  for (int ci = 0; ci < job->ncomp; ++ci) prog_max += job->hblk[ci] * job->vsamp[ci] * niter;   // reference :2474-2478
  int pp = progprec_arg;
  if (pp == 0) pp = 20;
  if (pp < 0) pp = prog_max;
  progprec_eff = pp;
  if (prog_max <= 0 || pp <= 0) return;
  int prog_thr = (int)((unsigned)(prog_max + pp - 1) / (unsigned)pp);
  long long units = 0;
  for (int ci = 0; ci < job->ncomp; ++ci)                  // every component of a plane-set job runs niter iterations
    for (int it = 0; it < niter; ++it) {                   // reference :2656-2664
      units += (long long)job->hblk[ci] * job->vsamp[ci];
      int cur = (int)units;
      if (cur >= prog_thr) {
        cur = (int)((long long)pp * cur / prog_max);
        prog_thr = (int)(((long long)(cur + 1) * prog_max + pp - 1) / pp);
        calls.push_back({units, cur});
      }
    }
}

bool qsj::ProgressPlan::advance(long long units_done) {
  while (!cancelled && made < calls.size() && calls[made].units <= units_done) {
    const int stop = fn(userdata, calls[made].cur, progprec_eff);
    ++made;
    if (stop) cancelled = true;
  }
  return cancelled;
}

int qsj::ProgressPlan::replay(void* self, int cur, int max) {
  ProgressPlan* P = static_cast<ProgressPlan*>(self);
  if (P->replayed < P->made) {                              // already reported by the pipelined route: answer from the record
    const size_t k = P->replayed++;
    return (P->cancelled && k == P->made - 1) ? 1 : 0;
  }
  return P->fn(P->userdata, cur, max);                      // beyond the record: live
}

extern "C" int qs_hip_progress_calls(const qs_hip_job* geometry, int niter, int progprec, int* cur_out, int max_calls, int* max_out) {
  if (!geometry || geometry->ncomp < 1 || geometry->ncomp > QS_HIP_MAXC) return qs_fail(QS_HIP_EINVAL, "qs_hip_progress_calls: bad job");
  niter = niter < 0 ? 0 : niter > 100 ? 100 : niter;       // reference :2455-2456
  ProgressPlan plan;
  plan.init(geometry, niter, progprec, nullptr, nullptr);
  if (max_out) *max_out = plan.progprec_eff;
  for (size_t k = 0; k < plan.calls.size() && cur_out && (int)k < max_calls; ++k) cur_out[k] = plan.calls[k].cur;
  return (int)plan.calls.size();
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

// ---------------------------------------------------------------------------
// prewarm: a FRESH process pays ~50 ms for hipInit, ~35 ms for the device context and its first hardware queue,
// ~16 ms for the first DMA, ~10 ms for the code object and ~15 ms per 64 MiB of pinned staging (tools/cold_phases,
// profiles/r03*) -- none of it depends on the coefficients, all of it can run while the application is still
// entropy-decoding the file (libjpeg needs 0.4 s for an 8192^2 image).  qs_hip_prewarm() does it on a detached
// thread; the first do_quantsmooth waits for that thread instead of repeating its work.
namespace {
std::mutex g_warm_mu;
std::condition_variable g_warm_cv;
int g_warm_running = 0;                 // prewarm threads at work (guarded by g_warm_mu)
unsigned long long g_warm_runtime_done = 0;   // bit d: device d's context / queues / code object were warmed (guarded by g_warm_mu)

// the runtime part of the warm-up runs once PER DEVICE (contexts, queues and the code object are per device)
bool warm_runtime_claim(int dev) {
  std::lock_guard<std::mutex> lk(g_warm_mu);
  const unsigned long long bit = 1ull << (dev & 63);
  if (g_warm_runtime_done & bit) return false;
  g_warm_runtime_done |= bit;
  return true;
}

void warm_runtime() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return; }
  const double t0 = wall_ms();
  { StreamLease lease; if (lease.p) lease.p->warm_all(); }   // device context + the three queues, each with its first copy done; parked in the pool
  const double t1 = wall_ms();
  DevBuf d; PinnedBuf h;
  if (d.alloc(1 << 20) == hipSuccess && h.alloc(1 << 20)) {
    memset(h.p, 0, 1 << 20);
    (void)hipMemcpy(d.p, h.p, 1 << 20, hipMemcpyHostToDevice);          // first DMA
    qs_launch_clamp(d.as<int16_t>(), 64, nullptr);                       // first kernel: loads the code object
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
  }
  (void)HostPool::get();                                     // helper threads
  if (trace_on()) fprintf(stderr, "qs_hip trace: prewarm  context+queue %.2f ms  first copy+kernel+helpers %.2f ms\n", t1 - t0, wall_ms() - t1);
}

void warm_pools(std::vector<size_t> pinned, std::vector<size_t> device) {
  const double t0 = wall_ms();
  {
    std::vector<std::unique_ptr<DevBuf>> hold;               // cold hipMalloc of a band's arenas costs 3-9 ms per group
    for (size_t n : device) {
      std::unique_ptr<DevBuf> b(new DevBuf);
      if (b->alloc(n) != hipSuccess) { (void)hipGetLastError(); break; }
      hold.push_back(std::move(b));
    }
  }
  const double t1 = wall_ms();
  std::vector<std::unique_ptr<PinnedBuf>> hold;              // all at once: distinct blocks; released to the pool together
  size_t total = 0;
  for (size_t n : pinned) {
    if (n < kStageMin || total + round_size(n) > ((size_t)2 << 30)) continue;
    std::unique_ptr<PinnedBuf> b(new PinnedBuf);
    if (!b->alloc(n)) break;
    total += b->n;
    hold.push_back(std::move(b));
  }
  if (trace_on()) fprintf(stderr, "qs_hip trace: prewarm  %zu device buffer(s) %.2f ms, %zu pinned block(s), %.0f MiB: %.2f ms\n",
                          device.size(), t1 - t0, hold.size(), total / 1048576.0, wall_ms() - t1);
}
}  // namespace

// Process exit: the detached background threads (prewarm, kick_background) must not be inside a HIP call when the
// HIP runtime's own static destructors run.  This library is unloaded before libamdhip64 (it depends on it), so its
// static destructor is the place to wait for them (bounded: they finish in tens of milliseconds).
namespace {
void wait_for_background() {
  for (int i = 0; i < 5000; ++i) {
    bool warm;
    { std::lock_guard<std::mutex> lk(g_warm_mu); warm = g_warm_running != 0; }
    if (!warm && !g_bg_busy.load()) return;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
}
struct BackgroundJoin {
  ~BackgroundJoin() { wait_for_background(); }
} g_background_join;
// That destructor is not early enough for a process that EXITS while a prewarm thread is still at work (an
// application whose libjpeg error handler fires right after jpegqs_start_decompress() began -- the reference's
// example.c on a CMYK file -- and returns from main()): the HIP runtime registers exit handlers of its own lazily,
// from inside hipInit / context creation, i.e. AFTER this library's static objects, so they run BEFORE the destructor
// above and tear the runtime down under the thread (seen on MI355X as a glibc heap-corruption abort at exit).  exit()
// runs handlers newest first, so the wait is registered with atexit() at three points, once each per process: when the
// first prewarm thread is started (newest while the thread is still inside hipInit), and from the thread itself right
// after hipInit and again after the context and queues exist (newer than anything the runtime registered by then).
void register_exit_wait(int point) {
  static std::atomic<unsigned> done{0};
  const unsigned bit = 1u << point;
  if (done.fetch_or(bit) & bit) return;
  (void)atexit(wait_for_background);
}
}  // namespace

void qsj::warm_wait() {
  std::unique_lock<std::mutex> lk(g_warm_mu);
  g_warm_cv.wait(lk, [] { return g_warm_running == 0; });
}

extern "C" int qs_hip_prewarm(const qs_hip_job* geometry, int flags, int niter) {
  try {
    std::vector<size_t> sizes, device;
    if (geometry && geometry->ncomp >= 1 && geometry->ncomp <= QS_HIP_MAXC) {
      qs_hip_job g = *geometry;
      int nit = niter;
      bool ok = true;
      for (int ci = 0; ci < g.ncomp; ++ci) ok = ok && g.wblk[ci] > 0 && g.hblk[ci] > 0 && (long long)g.wblk[ci] * g.hblk[ci] <= (1ll << 27);
      if (ok) {
        nit = nit < 0 ? 0 : nit > 100 ? 100 : nit;
        std::vector<size_t> one, px;
        if (job_fusable(&g, flags)) fused_stage_sizes(&g, nit, one, px);
        else for (int ci = 0; ci < g.ncomp; ++ci) {
          one.push_back((size_t)g.wblk[ci] * g.hblk[ci] * 128);
          px.push_back(qs_hip_plane_bytes(g.wblk[ci], g.hblk[ci]));
        }
        for (size_t k = 0; k < one.size(); ++k) {
          sizes.push_back(one[k]); sizes.push_back(one[k]);   // upload staging + download landing buffer
          device.push_back(one[k]); device.push_back(px[k]);  // coefficient arena + pixel planes
        }
        if ((flags & QS_UPSAMPLE_UV) && job_needs_lowres(&g, flags) && !(g.hsamp[0] == 1 && g.vsamp[0] == 1))
          for (int k = 0; k < 2; ++k) sizes.push_back((size_t)g.wblk[0] * g.hblk[0] * 128);   // the replacement arrays
      }
    }
    // The warm-up belongs to the CALLER's device: a new thread starts on HIP device 0, and the stream pool, the device
    // arenas and the context are all keyed on the current device -- warming device 0 for a caller that works on device
    // 3 would create a context and cache up to an image of memory on another worker's GPU and leave the caller's cold.
    const int dev = current_device();
    // "a prewarm thread is at work" is counted before the thread exists and given back on EVERY path on which the
    // thread does not come to exist (std::system_error from the constructor, bad_alloc copying the captures): a count
    // left behind would block every later warm_wait() -- i.e. every later job -- forever.
    struct WarmCount {
      bool armed = true;
      WarmCount() { std::lock_guard<std::mutex> lk(g_warm_mu); ++g_warm_running; }
      static void done() { { std::lock_guard<std::mutex> lk(g_warm_mu); --g_warm_running; } g_warm_cv.notify_all(); }
      ~WarmCount() { if (armed) done(); }
    } count;
    register_exit_wait(0);
    std::thread([sizes, device, dev]() {
      try {
        int n = 0;
        const bool have = hipGetDeviceCount(&n) == hipSuccess && n > 0;
        register_exit_wait(1);
        if (have && hipSetDevice(dev) == hipSuccess) {
          if (warm_runtime_claim(dev)) { warm_runtime(); register_exit_wait(2); }
          if (!sizes.empty()) warm_pools(sizes, device);
        } else (void)hipGetLastError();
      } catch (...) {}
      WarmCount::done();
    }).detach();
    count.armed = false;                                     // the thread owns the count now
    return QS_HIP_OK;
  } catch (...) {
    return QS_HIP_ENOMEM;                                    // (no thread, no memory: the job will simply start cold)
  }
}

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
  PinnedBuf::pending().clear();                              // (blocks wanted for next time: not any more)
  if (cur != current_device()) (void)hipSetDevice(cur);
}

// validation and the reference's early-outs; returns 1 when there is work to do,
// 0 when the job is already finished (result 0), < 0 on a bad job
int qsj::prepare_job(qs_hip_job* job, int flags, int* niter) {
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

int qsj::do_quantsmooth_impl(qs_hip_job* job, int flags, int niter, int progprec,
                               qs_hip_progress_fn progress, void* userdata) {
  const int todo = prepare_job(job, flags, &niter);
  if (todo <= 0) return todo;
  warm_wait();                                                 // a prewarm thread is doing what this call would do first
  if (qs_hip_device_count() <= 0)
    return qs_fail(QS_HIP_ENODEV, "no HIP device available (this library has no CPU fallback)");

  // A progress callback does not send the job to the slow route any more (missing 4): the plane-set
  // routes report completed work through a ProgressPlan; only the coupled-colour routes (whose component order IS the
  // reference's) keep the host-synchronised general route for callers that want progress.
  ProgressPlan plan;
  const bool fusable = job_fusable(job, flags);
  if (progress && fusable) plan.init(job, niter, progprec, progress, userdata);
  ProgressPlan* pl = (progress && fusable) ? &plan : nullptr;
  auto careful = [&]() {                                       // the reference's own order; calls already made are not repeated
    if (pl) return run_job(job, flags, niter, progprec, &ProgressPlan::replay, pl, /*eager=*/false);
    return run_job(job, flags, niter, progprec, progress, userdata, /*eager=*/false);
  };
  if (!progress || fusable) {                                  // several GPUs and a job worth spreading over them
    const std::vector<int> devs = shard_devices_for(job, flags, niter);
    if (!devs.empty() && (!progress || fusable)) {
      int r = run_sharded(job, flags, niter, devs, pl);
      if (r == JOB_RERUN_CAREFUL) r = careful();
      return r;
    }
  }
  if (fusable) {
    int result = QS_HIP_ENODEV;
    qs_hip_job* one[1] = { job };
    if (int r = run_fused(one, std::vector<int>{0}, flags, niter, &result, pl)) return r;
    return result;
  }
  int r = run_job(job, flags, niter, progprec, progress, userdata, /*eager=*/progress == nullptr);
  if (r == JOB_RERUN_CAREFUL) r = careful();
  return r;
}

// The C ABI never lets a C++ exception (std::bad_alloc from the host-side containers) escape.
extern "C" int qs_hip_do_quantsmooth(qs_hip_job* job, int flags, int niter, int progprec,
                                     qs_hip_progress_fn progress, void* userdata) {
  struct Kick { ~Kick() { kick_background(); } } kick;   // staging blocks this call missed are pinned afterwards
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
    struct Kick { ~Kick() { kick_background(); } } kick;
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
    struct Kick { ~Kick() { kick_background(); } } kick;
    const int todo = prepare_job(job, flags, &niter);
    if (todo <= 0) return todo;
    warm_wait();
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
