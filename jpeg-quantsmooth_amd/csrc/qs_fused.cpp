// qs_fused.cpp -- the plane-set route of the job layer: jobs whose components are independent of each
// other run as ONE pass-A and ONE pass-B launch per iteration over all their planes; a very large plane
// is cut into pipelined bands.  (Split out of qs_job.cpp; semantics in qs_jobint.h.)
#include <list>

#include "qs_jobint.h"

using namespace qsx;
using namespace qsj;

// ---------------------------------------------------------------------------
// fused execution: jobs whose components are independent of each other (no
// JOINT_YUV / UPSAMPLE_UV coupling, no LOW_QUALITY, ordinary quant tables) run
// as plane sets -- ONE pass-A and ONE pass-B launch per iteration for all
// components of all jobs of a group (qs_*_set_kernel), so that small images
// fill the chip together and a job does not occupy three hardware queues.
// Everything else about the job semantics is as in run_job (eager mode): the
// range-check flags are read once at the end, a job with a set flag is re-run in
// the careful order from its untouched host input.

namespace {
static size_t plane_stride(int wb, int hb) { return (qs_hip_plane_bytes(wb, hb) + 255) & ~(size_t)255; }
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
  // progress callback installed: one event behind every iteration's launch, and what an iteration of this group is
  // worth in the reference's progress units (own block rows x v_samp; halo rows of a band are somebody else's work)
  std::vector<hipEvent_t> it_ev;
  long long units = 0;
  // ... and the range-check flags right behind pass A of iteration 0 (the only pass that sets them): the reference makes
  // no progress call for a component whose range check trips (quantsmooth.h:2610 leaves the loop first), so the calls of
  // this group wait until ITS flags are known to be clear
  PinnedBuf hstatus0;
  hipEvent_t ev_status0 = nullptr;
  // QS_HIP_TRACE only: device timestamps "input is on the device" / "kernels done" / "results are in pinned host memory"
  hipEvent_t tev[3] = {nullptr, nullptr, nullptr};
  float t_ms[3] = {0, 0, 0};              // ... in ms after the call's first event, read when the group is drained
  FGroup() = default;
  FGroup(const FGroup&) = delete;
  FGroup& operator=(const FGroup&) = delete;
  ~FGroup() {
    for (hipEvent_t& e : tev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (hipEvent_t& e : it_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (ev_status0) { (void)hipEventDestroy(ev_status0); ev_status0 = nullptr; }
  }
  // everything queued on the group's stream has completed: give the arenas back
  void release_transients(bool keep_stage) {
    coef.release(); px.release(); cst.release(); status.release();
    hstatus.release(); hstatus0.release(); down.reset();
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

// ---- partition into groups (a job never straddles two, unless it is cut into bands)
static void partition(const qs_hip_job* const* jobs, const std::vector<int>& which, int niter,
                      std::list<FGroup>& groups, std::vector<char>& split) {
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
        std::vector<int> cut(1, 0);                         // band k = block rows [cut[k], cut[k + 1]): equal bands
        {                                                    // (a small first and last band was measured: slower, profiles/r04c_route)
          const int rows = (hb + bands - 1) / bands;
          for (int r = rows; r < hb; r += rows) cut.push_back(r);
        }
        cut.push_back(hb);
        for (size_t k = 0; k + 1 < cut.size(); ++k) {
          const int r0 = cut[k], r1 = cut[k + 1], d0 = std::max(0, r0 - niter), d1 = std::min(hb, r1 + niter);
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
}

}  // namespace

// what a plane-set run of this one job will ask for, one entry per group: the coefficient bytes (= the size of its
// pinned upload staging, of its pinned download landing buffer and of its device arena) and the bytes of its pixel planes
void qsj::fused_stage_sizes(const qs_hip_job* job, int niter, std::vector<size_t>& coef_bytes, std::vector<size_t>& px_bytes) {
  std::list<FGroup> groups;
  std::vector<char> split(1, 0);
  const qs_hip_job* one[1] = { job };
  partition(one, std::vector<int>{0}, niter, groups, split);
  for (const FGroup& G : groups) {
    size_t n = 0, px = 0;
    for (const FPlane& P : G.planes) { n += P.cbytes; px += 2 * plane_stride(P.wb, P.hb); }
    coef_bytes.push_back(n);
    px_bytes.push_back(px);
  }
}

int qsj::run_fused(qs_hip_job* const* jobs, const std::vector<int>& which, int flags, int niter, int* results, ProgressPlan* plan) {
  if (plan && which.size() != 1) plan = nullptr;             // (progress is a single-job matter)
  StreamLease lease;
  if (!lease.p) return qs_fail(QS_HIP_ENODEV, "could not create HIP streams: %s", hipGetErrorString(hipGetLastError()));
  std::list<FGroup> groups;
  DrainGuard drain{lease.p};
  const double t_start = wall_ms();
  double t_enq = t_start;

  int maxj = 0;
  for (int ji : which) maxj = std::max(maxj, ji);
  std::vector<char> split(maxj + 1, 0), bad_job(maxj + 1, 0), scattered(maxj + 1, 0), defer(maxj + 1, 0);
  std::vector<int> ngroups(maxj + 1, 0), ndone(maxj + 1, 0);   // groups a job's planes live in / groups whose results are back
  partition(jobs, which, niter, groups, split);
  for (const FGroup& G : groups) for (int ji : G.jobs) ++ngroups[ji];

  // ---- per group: upload, niter x (pass A, pass B), status readback, download into pinned memory
  const int diag = (flags & QS_DIAGONALS) != 0;
  size_t gi = 0;
  hipEvent_t t_first = nullptr;                              // QS_HIP_TRACE: time zero of the per-group device timestamps
  struct EvGuard { hipEvent_t& e; ~EvGuard() { if (e) (void)hipEventDestroy(e); } } t_first_guard{t_first};
  auto enqueue = [&](FGroup& G) -> int {
    const double t_g0 = wall_ms();
    G.s = lease.p->get_ready((int)(gi++ % 3));
    if (trace_on() && !t_first) { HIP_TRY(hipEventCreate(&t_first)); HIP_TRY(hipEventRecord(t_first, G.s)); }   // time zero: before the first upload
    const int np = (int)G.planes.size();
    size_t coef_bytes = 0, px_bytes = 0;
    std::vector<const uint16_t*> qtabs;
    for (FPlane& P : G.planes) {
      P.coef_off = coef_bytes; coef_bytes += P.cbytes;
      P.px_off = px_bytes; px_bytes += 2 * plane_stride(P.wb, P.hb);       // two planes: pass B writes the next iteration's
      const uint16_t* q = jobs[P.job]->quant[P.ci];
      for (size_t k = 0; k < qtabs.size() && P.cst < 0; ++k)
        if (!memcmp(qtabs[k], q, 64 * sizeof(uint16_t))) P.cst = (int)k;
      if (P.cst < 0) { P.cst = (int)qtabs.size(); qtabs.push_back(q); }
    }
    const double t_a0 = wall_ms();
    HIP_TRY(G.coef.alloc(coef_bytes));
    HIP_TRY(G.px.alloc(px_bytes));
    HIP_TRY(G.cst.alloc(qtabs.size() * sizeof(QsConsts)));
    HIP_TRY(G.status.alloc((size_t)np * sizeof(int32_t)));
    const double t_a1 = wall_ms();
    G.hc.resize(qtabs.size());
    for (size_t k = 0; k < qtabs.size(); ++k)
      if (int r = qs_hip_consts_build(&G.hc[k], qtabs[k], flags)) return r;
    const double t_a2 = wall_ms();
    HIP_TRY(hipMemcpyAsync(G.cst.p, G.hc.data(), qtabs.size() * sizeof(QsConsts), hipMemcpyHostToDevice, G.s));
    if (trace_on())
      fprintf(stderr, "qs_hip trace: fused  group %zu: stream %.2f ms, device buffers %.2f ms, consts build %.2f ms, consts copy %.2f ms\n",
              gi, t_a0 - t_g0, t_a1 - t_a0, t_a2 - t_a1, wall_ms() - t_a2);
    std::vector<Piece> pieces;
    for (const FPlane& P : G.planes)
      host_pieces(jobs[P.job], P.ci, P.src_row0, P.hb, P.coef_off, pieces);
    G.coef_bytes = coef_bytes;
    const double t_up0 = wall_ms();
    HIP_TRY(upload_pieces(G.coef.p, pieces, coef_bytes, G.s, G.stage));
    if (trace_on())
      fprintf(stderr, "qs_hip trace: fused  group %zu: alloc+consts %.2f ms, upload of %.1f MiB in %zu piece(s) %.2f ms (%s)\n",
              gi, t_up0 - t_g0, coef_bytes / 1048576.0, pieces.size(), wall_ms() - t_up0, G.stage.p ? "staged" : "direct");
    HIP_TRY(hipMemsetAsync(G.status.p, 0, (size_t)np * sizeof(int32_t), G.s));
    if (trace_on()) {
      for (hipEvent_t& e : G.tev) HIP_TRY(hipEventCreate(&e));
      HIP_TRY(hipEventRecord(G.tev[0], G.s));
    }

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
    // pass A once (dequantise, range check, first pixel planes); every pass B but the last writes the next
    // iteration's planes itself (fused pass A, ping-pong between the two planes of each FPlane)
    qs_launch_idct_set(set, 1, G.s);
    if (plan) {                                              // the flags as pass A left them, for the progress calls (see FGroup)
      if (!G.hstatus0.alloc((size_t)np * sizeof(int32_t))) return qs_fail(QS_HIP_ENOMEM, "out of pinned host memory");
      HIP_TRY(hipMemcpyAsync(G.hstatus0.p, G.status.p, (size_t)np * sizeof(int32_t), hipMemcpyDeviceToHost, G.s));
      HIP_TRY(hipEventCreateWithFlags(&G.ev_status0, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(G.ev_status0, G.s));
    }
    for (int it = 0; it < niter; ++it) {
      for (int i = 0; i < np; ++i) {
        uint8_t* a = G.px.as<uint8_t>() + G.planes[i].px_off;
        uint8_t* b = a + plane_stride(G.planes[i].wb, G.planes[i].hb);
        set.ref[i].plane = (it & 1) ? b : a;
        set.ref[i].plane_next = it == niter - 1 ? nullptr : (it & 1) ? a : b;
      }
      qs_launch_smooth_set(set, diag, it == niter - 1, G.s);
      if (plan) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        G.it_ev.push_back(e);
        HIP_TRY(hipEventRecord(e, G.s));
      }
    }
    if (plan) for (const FPlane& P : G.planes) G.units += (long long)(P.keep1 - P.keep0) * jobs[P.job]->vsamp[P.ci];
    HIP_TRY(hipGetLastError());
    if (G.tev[1]) HIP_TRY(hipEventRecord(G.tev[1], G.s));
    // pinned: a pageable destination would make this call wait for the whole stream
    if (!G.hstatus.alloc((size_t)np * sizeof(int32_t))) return qs_fail(QS_HIP_ENOMEM, "out of pinned host memory");
    HIP_TRY(hipMemcpyAsync(G.hstatus.p, G.status.p, (size_t)np * sizeof(int32_t), hipMemcpyDeviceToHost, G.s));
    HIP_TRY(G.down.issue(G.coef.p, coef_bytes, G.s, rows_active()));       // to pinned memory, right behind the kernels
    if (G.tev[2]) HIP_TRY(hipEventRecord(G.tev[2], G.s));
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
  long long units_done = 0;                                  // progress: completed group-iterations, in the reference's units
  auto drain_group = [&](FGroup& G) -> int {
    // Progress (reference :2656-2664): every iteration of this group that has completed on the device counts; the
    // calls whose share of the work is done are made now, in the reference's sequence.  A cancel is handled like a
    // tripped range check: nothing more of the job is written, rows already written are restored, and the job is
    // re-run in the reference's order with the recorded answers (below).
    bool tripped = false;
    if (plan && G.ev_status0) {                              // the group's range check first: a tripped group reports nothing
      HIP_TRY(hipEventSynchronize(G.ev_status0));
      const int32_t* h0 = static_cast<const int32_t*>(G.hstatus0.p);
      for (size_t i = 0; i < G.planes.size(); ++i) if (h0[i]) { bad_job[G.planes[i].job] = 1; tripped = true; }
    }
    if (plan && !plan->cancelled && !tripped)
      for (hipEvent_t e : G.it_ev) {
        HIP_TRY(hipEventSynchronize(e));
        units_done += G.units;
        if (plan->advance(units_done)) { for (int ji : G.jobs) bad_job[ji] = 1; break; }
      }
    HIP_TRY(G.down.wait_first(G.s));
    const int32_t* hst = static_cast<const int32_t*>(G.hstatus.p);
    for (size_t i = 0; i < G.planes.size(); ++i) if (hst[i]) bad_job[G.planes[i].job] = 1;
    bool hold = false, banded = false;
    for (int ji : G.jobs) { hold |= (defer[ji] != 0); banded |= (split[ji] != 0); }
    if (hold) { held.push_back(&G); return QS_HIP_OK; }
    std::vector<Piece> back;
    for (const FPlane& P : G.planes)
      if (!bad_job[P.job]) { result_pieces(P, back); scattered[P.job] = 1; }
    if (!G.stage.p) HIP_TRY(G.down.land(G.coef.p, G.s));        // no restore copy: land first, write afterwards
    HIP_TRY(G.down.finish(G.coef.p, back, G.s, G.stage.p != nullptr));
    for (int ji : G.jobs) ++ndone[ji];
    if (G.tev[2] && t_first && hipEventSynchronize(G.tev[2]) == hipSuccess)
      for (int k = 0; k < 3; ++k) if (hipEventElapsedTime(&G.t_ms[k], t_first, G.tev[k]) != hipSuccess) { (void)hipGetLastError(); G.t_ms[k] = -1; }
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
      if (plan && plan->cancelled) break;                    // cancelled: nothing further is started
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
    for (FGroup* G : held) HIP_TRY(G->down.land(G->coef.p, G->s));   // held back = no restore copy: all land, then all are written
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
    lease.p->sync_all();
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
  if (trace_on()) {
    fprintf(stderr, "qs_hip trace: fused  %zu job(s) in %zu group(s)  enqueue %.2f ms  drain+download %.2f ms  (%zu re-run)%s\n",
            which.size(), groups.size(), t_enq - t_start, wall_ms() - t_enq, rerun.size(),
            plan ? (plan->cancelled ? "  progress: cancelled by the callback" : "  progress: callback served from this route") : "");
    // device timeline per group, ms after the first group's stream work began: input on the device / kernels done / results in pinned memory
    fprintf(stderr, "qs_hip trace: fused  device timeline (upload done, kernels done, download done):");
    for (const FGroup& G : groups) fprintf(stderr, " [%.2f %.2f %.2f]", G.t_ms[0], G.t_ms[1], G.t_ms[2]);
    fprintf(stderr, "\n");
  }
  for (int ji : which) {
    if (bad_job[ji]) continue;
    for (int ci = 0; ci < jobs[ji]->ncomp; ++ci)           // reference :2851-2859
      for (int i = 0; i < 64; ++i) jobs[ji]->quant[ci][i] = 1;
  }
  const double t_clear = wall_ms();
  groups.clear();                                            // give the arenas back before the re-runs allocate
  if (trace_on()) fprintf(stderr, "qs_hip trace: fused  release %.2f ms\n", wall_ms() - t_clear);
  for (int ji : rerun)
    results[ji] = plan ? run_job(jobs[ji], flags, niter, plan->progprec, &ProgressPlan::replay, plan, /*eager=*/false)
                       : run_job(jobs[ji], flags, niter, 0, nullptr, nullptr, /*eager=*/false);
  return QS_HIP_OK;
}
