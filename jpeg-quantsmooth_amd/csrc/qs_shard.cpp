// qs_shard.cpp -- one job spread over several GPUs from ONE host process, inside the C ABI.
//
// The reference parallelises do_quantsmooth() internally (OpenMP over block rows,
// reference quantsmooth.h:2587-2640); this is the same idea one level up: every component is
// cut into contiguous block-row BANDS, band d lives on device d for the whole job, and after
// each pass A a band pulls ONE PIXEL ROW from each neighbouring band into its apron row --
// the only data the recovery loop reads across a band edge (reference :1396-1401; SURVEY.md
// section 8e).  Image-edge bands replicate instead (QS_PLANE_REP_TOP / _BOT of the plane set).
// Bit-exact with the unsharded result: no arithmetic changes, only where a block runs.
//
// Mechanics: one stream per band; all launches come from the calling thread, which walks
// over the devices with hipSetDevice (no helper processes, no launcher).  The halo rows
// move device-to-device with hipMemcpyPeerAsync over xGMI (a plain device copy when two
// bands share a GPU), ordered by events:
//     A[d]  recorded after pass A of band d       -> neighbours wait for it before they pull
//     X[d]  recorded after band d's pulls          -> neighbours wait for it before their next
//                                                     pass A overwrites the rows it read
// The host never waits inside the iteration loop.  Range-check flags are read once at the end
// (nothing reaches caller memory before that); a set flag sends the job to the careful
// single-device route, as in qs_job.cpp.
//
// Routes: independent components (no JOINT_YUV / UPSAMPLE_UV coupling, no LOW_QUALITY; CLI
// --quality 3/4, any colour layout) run as ONE plane set per band and pass -- run_sharded_set.
// That route has a second, COMMUNICATION-AVOIDING schedule (qs_hip_set_shard_schedule(1) /
// QS_HIP_SHARD_SCHEDULE=deep): a band carries `niter` extra block rows of its neighbours on each cut
// side and runs all iterations without any exchange -- the error of treating the cut as an image edge
// travels one block row per iteration and stops short of the rows the band owns (what the banded
// one-device route of qs_fused.cpp does between its bands).  +2 * niter block rows of work per inner
// band (+4.7 % at 8192^2 over 8 devices, niter 3) against niter exchanges whose cost is pure latency.
// Coupled YCbCr jobs (--quality 5/6) are cut on chroma block rows (the luma band is the
// v_samp-times taller range of the same image rows) and add three one-off exchanges: the
// low-res luma plane, the refreshed chroma planes -- run_sharded_colour.
#include <string>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library does not link librccl (see qs_hip_do_quantsmooth_band)

#include "qs_jobint.h"

using namespace qsx;
using namespace qsj;

namespace {

// block rows [r0, r1) of band `d` of `n`: the exported definition (qs_planes.cpp), shared with bands.py
static void band_rows(int hblk, int n, int d, int& r0, int& r1) {
  (void)qs_hip_band_rows(hblk, n, d, 1, &r0, &r1);
}

struct BandPlane {       // one component's band on one device
  int ci = 0, wb = 0, hb = 0, r0 = 0;    // the block rows [r0, r0 + hb) live on the device
  int own0 = 0, own_hb = -1;             // ... of which [r0 + own0, r0 + own0 + own_hb) are the band's own (written back);
                                         // -1: all of them (the exchange schedule); fewer with the deep-halo schedule
  int own_rows() const { return own_hb < 0 ? hb : own_hb; }
  size_t own_off() const { return coef_off + (size_t)own0 * wb * 128; }   // arena offset of the first owned row
  bool halo_top = false, halo_bot = false;
  size_t coef_off = 0, px_off = 0, cbytes = 0;
  int cst = -1;
};

struct Band {            // one (logical) device
  int dev = 0, index = 0;
  Streams* st = nullptr;                 // leased under `dev`
  hipStream_t s = nullptr;
  hipEvent_t evA = nullptr, evX[2] = {nullptr, nullptr};   // evX: "my pulls of exchange n are done", alternating (see before_overwrite)
  hipEvent_t evT0 = nullptr, evT1 = nullptr;   // QS_HIP_TRACE only: first pass A .. last pass B on this band's stream
  std::vector<BandPlane> planes;
  DevBuf coef, px, cst, status, aux[8];   // aux: route-specific planes (colour route)
  PinnedBuf stage, hstatus, hstatus0;      // hstatus0 / evS: the range-check flags right behind pass A of iteration 0 (progress route)
  hipEvent_t evS = nullptr;
  Download down, down_up[2];
  std::vector<QsConsts> hc;
  QsPlaneSet set;
  int x_count = 0;                                           // exchanges this band has recorded an evX for
};

// All per-device resources of a sharded job.  Destruction order matters: first drain every
// stream (under its own device), then release buffers, then return the stream sets.
struct Bands {
  std::vector<Band> b;
  int home = 0;
  explicit Bands(size_t n) : b(n) { home = current_device(); }
  ~Bands() {
    for (Band& B : b) {
      if (!B.st) continue;
      (void)hipSetDevice(B.dev);
      B.st->sync_all();
    }
    for (Band& B : b) {
      (void)hipSetDevice(B.dev);
      if (B.evA) (void)hipEventDestroy(B.evA);
      for (hipEvent_t e : B.evX) if (e) (void)hipEventDestroy(e);
      if (B.evT0) (void)hipEventDestroy(B.evT0);
      if (B.evT1) (void)hipEventDestroy(B.evT1);
      if (B.evS) (void)hipEventDestroy(B.evS);
      B.down.reset(); B.down_up[0].reset(); B.down_up[1].reset();
      B.coef.release(); B.px.release(); B.cst.release(); B.status.release();
      for (auto& a : B.aux) a.release();
      if (B.st) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        g_stream_pool.push_back(B.st);
      }
    }
    (void)hipSetDevice(home);
  }
};

static int open_bands(Bands& bands, const std::vector<int>& devices) {
  const int ndev_visible = qs_hip_device_count();
  for (size_t d = 0; d < devices.size(); ++d) {
    Band& B = bands.b[d];
    B.dev = devices[d]; B.index = (int)d;
    if (B.dev < 0 || B.dev >= ndev_visible) return qs_fail(QS_HIP_EINVAL, "sharded job: no HIP device %d", B.dev);
    HIP_TRY(hipSetDevice(B.dev));
    StreamLease lease;                                       // (streams of the now-current device)
    if (!lease.p) return qs_fail(QS_HIP_ENODEV, "could not create HIP streams on device %d", B.dev);
    B.st = lease.p; lease.p = nullptr;                       // Bands::~Bands gives it back
    B.s = B.st->s[0];
    HIP_TRY(hipEventCreateWithFlags(&B.evA, hipEventDisableTiming));
    for (hipEvent_t& e : B.evX) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (trace_on()) { HIP_TRY(hipEventCreate(&B.evT0)); HIP_TRY(hipEventCreate(&B.evT1)); }
  }
  // direct xGMI access between neighbouring devices (without it the runtime stages peer copies
  // through host memory); "already enabled" is not an error
  for (size_t d = 0; d + 1 < devices.size(); ++d) {
    const int a = devices[d], c = devices[d + 1];
    if (a == c) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, a, c) == hipSuccess && can) {
      (void)hipSetDevice(a); (void)hipDeviceEnablePeerAccess(c, 0); (void)hipGetLastError();
      (void)hipSetDevice(c); (void)hipDeviceEnablePeerAccess(a, 0); (void)hipGetLastError();
    }
  }
  return QS_HIP_OK;
}

// dst (on band D's device) <- src (on band S's device), queued on D's stream
static hipError_t pull(Band& D, void* dst, const Band& S, const void* src, size_t n) {
  if (D.dev == S.dev) return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, D.s);
  return hipMemcpyPeerAsync(dst, D.dev, src, S.dev, n, D.s);
}

// One halo exchange: `nplanes` plane roles at once.  `plane(d, k, &p, &wb, &hb)` yields band d's
// plane of role k (device pointer, width / height in blocks), or false when the band holds no
// such plane.  Bands pull the last pixel row of the band above into their y = -1 apron row and
// the first pixel row of the band below into their y = h apron row.
// Caller contract: every band's producer kernels for these planes are already queued on its
// stream; the call records A[d] itself.
template <class PlaneFn>
static int exchange(Bands& bands, int nplanes, PlaneFn plane) {
  const size_t n = bands.b.size();
  for (Band& B : bands.b) {
    HIP_TRY(hipSetDevice(B.dev));
    HIP_TRY(hipEventRecord(B.evA, B.s));
  }
  for (size_t d = 0; d < n; ++d) {
    Band& B = bands.b[d];
    HIP_TRY(hipSetDevice(B.dev));
    bool wait_up = false, wait_dn = false;
    for (int k = 0; k < nplanes; ++k) {
      uint8_t *mine, *theirs; int wb, hb, wb2, hb2;
      if (!plane((int)d, k, &mine, &wb, &hb)) continue;
      size_t recv_top, recv_bot, pitch, their_top, their_bot;
      (void)qs_hip_band_halo_rows(wb, hb, nullptr, nullptr, &recv_top, &recv_bot, &pitch);
      if (d > 0 && plane((int)d - 1, k, &theirs, &wb2, &hb2)) {
        Band& U = bands.b[d - 1];
        (void)qs_hip_band_halo_rows(wb2, hb2, nullptr, &their_bot, nullptr, nullptr, nullptr);
        if (!wait_up) { HIP_TRY(hipStreamWaitEvent(B.s, U.evA, 0)); wait_up = true; }
        HIP_TRY(pull(B, mine + recv_top, U, theirs + their_bot, pitch));
      }
      if (d + 1 < n && plane((int)d + 1, k, &theirs, &wb2, &hb2)) {
        Band& L = bands.b[d + 1];
        (void)qs_hip_band_halo_rows(wb2, hb2, &their_top, nullptr, nullptr, nullptr, nullptr);
        if (!wait_dn) { HIP_TRY(hipStreamWaitEvent(B.s, L.evA, 0)); wait_dn = true; }
        HIP_TRY(pull(B, mine + recv_bot, L, theirs + their_top, pitch));
      }
    }
  }
  for (Band& B : bands.b) {                                  // "my pulls are done" -- see before_overwrite
    HIP_TRY(hipSetDevice(B.dev));
    HIP_TRY(hipEventRecord(B.evX[B.x_count & 1], B.s));
    ++B.x_count;
  }
  return QS_HIP_OK;
}

// Before band d overwrites planes its neighbours may still be pulling from: wait for the neighbours' "pulls done".
// lag = 0: of the latest exchange.  lag = 1: of the exchange BEFORE the latest -- for a fused pass B, which reads the
// planes the latest exchange filled and overwrites the OTHER set (ping-pong), last pulled from one exchange earlier:
// the long kernel then does not queue behind the neighbours' current halo pulls.  Two alternating
// events per band: a wait captures the record it was queued behind, the event is free again two exchanges later.
static int before_overwrite(Bands& bands, size_t d, int lag = 0) {
  Band& B = bands.b[d];
  for (size_t n : {d - 1, d + 1}) {
    if (n >= bands.b.size()) continue;                       // (d - 1 wraps for d = 0)
    const Band& N = bands.b[n];
    if (N.x_count > lag) HIP_TRY(hipStreamWaitEvent(B.s, N.evX[(N.x_count - 1 - lag) & 1], 0));
  }
  return QS_HIP_OK;
}

static size_t plane_stride(int wb, int hb) { return (qs_hip_plane_bytes(wb, hb) + 255) & ~(size_t)255; }

// the per-band constant blocks: one per distinct quant table among the band's planes
static int upload_consts(Band& B, const qs_hip_job* job, int flags) {
  std::vector<const uint16_t*> qtabs;
  for (BandPlane& P : B.planes) {
    const uint16_t* q = job->quant[P.ci];
    P.cst = -1;
    for (size_t k = 0; k < qtabs.size() && P.cst < 0; ++k)
      if (!memcmp(qtabs[k], q, 64 * sizeof(uint16_t))) P.cst = (int)k;
    if (P.cst < 0) { P.cst = (int)qtabs.size(); qtabs.push_back(q); }
  }
  HIP_TRY(B.cst.alloc(std::max<size_t>(1, qtabs.size()) * sizeof(QsConsts)));
  B.hc.resize(qtabs.size());
  for (size_t k = 0; k < qtabs.size(); ++k)
    if (int r = qs_hip_consts_build(&B.hc[k], qtabs[k], flags)) return r;
  if (!qtabs.empty())
    HIP_TRY(hipMemcpyAsync(B.cst.p, B.hc.data(), qtabs.size() * sizeof(QsConsts), hipMemcpyHostToDevice, B.s));
  return QS_HIP_OK;
}

// arenas + upload of the band's coefficient rows + plane set
static int stage_band(Band& B, const qs_hip_job* job, int flags) {
  size_t coef_bytes = 0, px_bytes = 0;
  for (BandPlane& P : B.planes) {
    P.cbytes = (size_t)P.wb * P.hb * 128;
    P.coef_off = coef_bytes; coef_bytes += P.cbytes;
    P.px_off = px_bytes; px_bytes += 2 * plane_stride(P.wb, P.hb);   // two planes each (the set route ping-pongs: fused pass A)
  }
  const int np = (int)B.planes.size();
  HIP_TRY(B.coef.alloc(std::max<size_t>(coef_bytes, 64)));
  HIP_TRY(B.px.alloc(std::max<size_t>(px_bytes, 64)));
  HIP_TRY(B.status.alloc((size_t)std::max(np, 1) * sizeof(int32_t)));
  if (int r = upload_consts(B, job, flags)) return r;
  std::vector<Piece> pieces;
  for (const BandPlane& P : B.planes)
    host_pieces(job, P.ci, P.r0, P.hb, P.coef_off, pieces);
  if (coef_bytes) HIP_TRY(upload_pieces(B.coef.p, pieces, coef_bytes, B.s, B.stage));
  HIP_TRY(hipMemsetAsync(B.status.p, 0, (size_t)std::max(np, 1) * sizeof(int32_t), B.s));

  QsPlaneSet& set = B.set;
  memset(&set, 0, sizeof set);
  set.n = np;
  int w = 0;
  for (int i = 0; i < np; ++i) {
    const BandPlane& P = B.planes[i];
    set.wave0[i] = w;
    w += (P.wb * P.hb + 63) / 64;
    QsPlaneRef& R = set.ref[i];
    R.cst = B.cst.as<QsConsts>() + P.cst;
    R.coef = reinterpret_cast<int16_t*>(B.coef.as<char>() + P.coef_off);
    R.plane = B.px.as<uint8_t>() + P.px_off;
    R.status = B.status.as<int32_t>() + i;
    R.wblk = P.wb; R.hblk = P.hb; R.pitch = qs_plane_pitch(P.wb);
    R.mode = (comp_rebalance(job, P.ci, flags) ? QS_PLANE_REBALANCE : 0) |
             (P.halo_top ? 0 : QS_PLANE_REP_TOP) | (P.halo_bot ? 0 : QS_PLANE_REP_BOT);
  }
  for (int i = np; i < QS_MAX_PLANES + 2; ++i) set.wave0[i] = w;
  return QS_HIP_OK;
}

static const BandPlane* find_plane(const Band& B, int ci) {
  for (const BandPlane& P : B.planes) if (P.ci == ci) return &P;
  return nullptr;
}

// range-check flags of every band -> true when any plane tripped
static int read_flags(Bands& bands, bool& bad) {
  bad = false;
  for (Band& B : bands.b) {
    HIP_TRY(hipSetDevice(B.dev));
    HIP_TRY(B.down.wait_first(B.s));
    const int32_t* hst = static_cast<const int32_t*>(B.hstatus.p);
    for (size_t i = 0; i < B.planes.size(); ++i) bad |= hst[i] != 0;
  }
  return QS_HIP_OK;
}

// Error exit while the bands' results are being scattered: bands already written must not stay
// behind in caller memory when the call reports a failure ("image left untouched").  Every band's
// pinned upload staging still holds its original rows; put them back.  (A band whose upload went
// straight from caller memory -- under 1 MiB, or pinned memory exhausted -- has no such copy.)
static void restore_bands(Bands& bands, const qs_hip_job* job) {
  for (Band& B : bands.b) {
    (void)hipSetDevice(B.dev);
    (void)hipStreamSynchronize(B.s);
    if (!B.stage.p) continue;
    std::vector<Piece> pcs;
    for (const BandPlane& P : B.planes) host_pieces(job, P.ci, P.r0 + P.own0, P.own_rows(), P.own_off(), pcs);   // (only these are ever written)
    for (const Piece& pc : pcs) memcpy(pc.host, static_cast<const char*>(B.stage.p) + pc.off, pc.len);
  }
}
#define HIP_TRY_RESTORE(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { restore_bands(bands, job); \
  return qs_fail(e_ == hipErrorOutOfMemory ? QS_HIP_ENOMEM : QS_HIP_ENODEV, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)

// Without the pinned upload staging of EVERY band there is nothing to restore from: then all results land in
// library-owned memory first (the only step that can fail) and are copied to the caller afterwards.
static int land_all_if_no_copy(Bands& bands, bool upsample) {
  bool have_copy = true;
  for (Band& B : bands.b) have_copy = have_copy && B.stage.p;
  if (have_copy) return QS_HIP_OK;
  for (Band& B : bands.b) {
    HIP_TRY(hipSetDevice(B.dev));
    HIP_TRY(B.down.land(B.coef.p, B.s));
    if (upsample) for (int j = 0; j < 2; ++j) HIP_TRY(B.down_up[j].land(B.aux[3 + j].p, B.s));
  }
  return QS_HIP_OK;
}

// ---------------------------------------------------------------------------
// independent components: one plane set per band
static int run_sharded_set(qs_hip_job* job, int flags, int niter, const std::vector<int>& devices, ProgressPlan* plan) {
  const double t_start = wall_ms();
  Bands bands(devices.size());
  if (int r = open_bands(bands, devices)) return r;
  const int n = (int)devices.size();

  // every component is cut into min(n, hblk / 8) bands (a band keeps at least 8 block rows),
  // which go to the first devices of the list
  // deep: the communication-avoiding schedule (see the top of this file) -- every cut side of a band carries niter
  // block rows of the neighbouring band, both plane edges replicate like image edges, nothing is exchanged
  // (only while the extra rows stay a fraction of a band: 4 * niter <= the shortest band; beyond that -- tiny images, or
  //  niter in the dozens -- the exchange schedule is the cheaper one)
  bool deep = shard_schedule_deep();
  for (int ci = 0; ci < job->ncomp && deep; ++ci) {
    const int nb = std::max(1, std::min(n, job->hblk[ci] / 8));
    if (nb > 1 && job->hblk[ci] / nb < 4 * niter) deep = false;
  }
  for (int ci = 0; ci < job->ncomp; ++ci) {
    const int hb = job->hblk[ci];
    const int nb = std::max(1, std::min(n, hb / 8));
    for (int d = 0; d < nb; ++d) {
      BandPlane P;
      int r0, r1;
      band_rows(hb, nb, d, r0, r1);
      P.ci = ci; P.wb = job->wblk[ci]; P.hb = r1 - r0; P.r0 = r0;
      P.halo_top = d > 0; P.halo_bot = d < nb - 1;
      if (deep) {
        const int e0 = std::max(0, r0 - niter), e1 = std::min(hb, r1 + niter);
        P.own0 = r0 - e0; P.own_hb = r1 - r0;
        P.r0 = e0; P.hb = e1 - e0;
        P.halo_top = P.halo_bot = false;                     // the cut is treated as an image edge (replicated)
      }
      bands.b[d].planes.push_back(P);
    }
  }
  for (Band& B : bands.b) {
    HIP_TRY(hipSetDevice(B.dev));
    if (int r = stage_band(B, job, flags)) return r;
  }
  const double t_up = wall_ms();

  const int diag = (flags & QS_DIAGONALS) != 0;
  if (trace_on()) for (Band& B : bands.b) { HIP_TRY(hipSetDevice(B.dev)); HIP_TRY(hipEventRecord(B.evT0, B.s)); }
  // progress callback installed: one event per band and iteration, in enqueue order (iteration-major)
  struct ItEv { int dev; hipEvent_t e; };
  std::vector<ItEv> it_ev;
  struct ItEvFree { std::vector<ItEv>& v; ~ItEvFree() { for (ItEv& x : v) { (void)hipSetDevice(x.dev); (void)hipEventDestroy(x.e); } } } it_ev_free{it_ev};
  // Progress (reference :2656-2664): the bands advance in lock step, so "iteration it of every component" is what
  // completes; the calls whose share of the work that covers are made, in the reference's sequence, when the LAST band
  // has finished the iteration.  With a callback installed the enqueue below stays at most ONE iteration ahead of what has
  // been reported, so that a cancel stops further launches instead of letting the whole job run on every GPU first
  // (ADVICE round 5).  A cancel ends like a tripped range check: the host input is untouched, the caller re-runs the job
  // in the reference's order with the recorded answers (ProgressPlan::replay).
  long long per_iter = 0;
  for (int ci = 0; ci < job->ncomp; ++ci) per_iter += (long long)job->hblk[ci] * job->vsamp[ci];
  int reported = 0;                                          // iterations whose progress calls have been made
  bool tripped = false;
  auto report_upto = [&](int upto) -> int {                  // make the calls of iterations [reported, upto)
    const size_t nb = bands.b.size();
    if (reported == 0 && upto > 0) {                         // every band's range check before the first call
      for (Band& B : bands.b) {
        HIP_TRY(hipSetDevice(B.dev));
        HIP_TRY(hipEventSynchronize(B.evS));
        const int32_t* h0 = static_cast<const int32_t*>(B.hstatus0.p);
        for (size_t i = 0; i < B.planes.size(); ++i) tripped |= h0[i] != 0;
      }
      if (tripped) return QS_HIP_OK;                         // no call made: the careful route makes them live
    }
    for (; reported < upto && !plan->cancelled; ++reported) {
      for (size_t d = 0; d < nb; ++d) {
        const ItEv& x = it_ev[(size_t)reported * nb + d];
        HIP_TRY(hipSetDevice(x.dev));
        HIP_TRY(hipEventSynchronize(x.e));
      }
      // the callback runs on the caller's device, whatever band was synchronised last (it may use HIP / torch itself)
      HIP_TRY(hipSetDevice(bands.home));
      plan->advance(per_iter * (reported + 1));
    }
    return QS_HIP_OK;
  };
  auto abandon = [&](const char* why) -> int {              // drain every band; nothing has reached caller memory
    for (Band& B : bands.b) { HIP_TRY(hipSetDevice(B.dev)); HIP_TRY(hipStreamSynchronize(B.s)); }
    if (trace_on()) fprintf(stderr, "qs_hip trace: sharded(set) %s\n", why);
    return JOB_RERUN_CAREFUL;
  };
  // Pass A runs once; every pass B but the last writes the next iteration's pixel planes itself (fused pass A) into
  // the band's second set of planes -- plane (it & 1) is read, plane ((it + 1) & 1) written.
  for (int it = 0; it < niter; ++it) {
    const int cur = it & 1;
    if (plan && it >= 2) {                                   // iteration it - 2 must have been reported before it is queued
      if (int r = report_upto(it - 1)) return r;
      if (tripped) return abandon("range check tripped");
      if (plan->cancelled) return abandon("cancelled by the progress callback");
    }
    if (it == 0)
      for (size_t d = 0; d < bands.b.size(); ++d) {
        Band& B = bands.b[d];
        HIP_TRY(hipSetDevice(B.dev));
        if (int r = before_overwrite(bands, d)) return r;
        qs_launch_idct_set(B.set, 1, B.s);
        if (plan) {
          // the range check is pass A's: its flags travel now, so that no progress call is made for a job the reference
          // would have left before its first call (quantsmooth.h:2610)
          const size_t np = std::max<size_t>(1, B.planes.size());
          if (!B.hstatus0.alloc(np * sizeof(int32_t))) return qs_fail(QS_HIP_ENOMEM, "out of pinned host memory");
          HIP_TRY(hipMemcpyAsync(B.hstatus0.p, B.status.p, np * sizeof(int32_t), hipMemcpyDeviceToHost, B.s));
          HIP_TRY(hipEventCreateWithFlags(&B.evS, hipEventDisableTiming));
          HIP_TRY(hipEventRecord(B.evS, B.s));
        }
      }
    // one pixel row per component and band edge, of the planes this iteration's pass B reads
    if (!deep)
    if (int r = exchange(bands, job->ncomp, [&](int d, int ci, uint8_t** p, int* wb, int* hb) {
          const BandPlane* P = find_plane(bands.b[d], ci);
          if (!P) return false;
          *p = bands.b[d].px.as<uint8_t>() + P->px_off + (cur ? plane_stride(P->wb, P->hb) : 0); *wb = P->wb; *hb = P->hb;
          return true;
        })) return r;
    for (size_t d = 0; d < bands.b.size(); ++d) {
      Band& B = bands.b[d];
      HIP_TRY(hipSetDevice(B.dev));
      for (size_t i = 0; i < B.planes.size(); ++i) {
        uint8_t* a = B.px.as<uint8_t>() + B.planes[i].px_off;
        uint8_t* b = a + plane_stride(B.planes[i].wb, B.planes[i].hb);
        B.set.ref[i].plane = cur ? b : a;
        B.set.ref[i].plane_next = it == niter - 1 ? nullptr : cur ? a : b;
      }
      // the planes about to be written are the ones the neighbours pulled their halo rows from one iteration ago
      if (it < niter - 1) if (int r = before_overwrite(bands, d, /*lag=*/1)) return r;
      qs_launch_smooth_set(B.set, diag, it == niter - 1, B.s);
      if (plan) {                                              // progress: "iteration `it` is done on this band"
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        it_ev.push_back({B.dev, e});
        HIP_TRY(hipEventRecord(e, B.s));
      }
    }
  }
  if (trace_on()) for (Band& B : bands.b) { HIP_TRY(hipSetDevice(B.dev)); HIP_TRY(hipEventRecord(B.evT1, B.s)); }
  for (Band& B : bands.b) {
    HIP_TRY(hipSetDevice(B.dev));
    HIP_TRY(hipGetLastError());
    const size_t np = std::max<size_t>(1, B.planes.size());
    if (!B.hstatus.alloc(np * sizeof(int32_t))) return qs_fail(QS_HIP_ENOMEM, "out of pinned host memory");
    HIP_TRY(hipMemcpyAsync(B.hstatus.p, B.status.p, np * sizeof(int32_t), hipMemcpyDeviceToHost, B.s));
    size_t coef_bytes = 0;
    for (const BandPlane& P : B.planes) coef_bytes += P.cbytes;
    HIP_TRY(B.down.issue(B.coef.p, coef_bytes, B.s, rows_active()));
  }
  const double t_enq = wall_ms();

  if (plan) {
    if (int r = report_upto(niter)) return r;
    if (tripped) return abandon("range check tripped");
    if (plan->cancelled) return abandon("cancelled by the progress callback");
  }
  bool bad = false;
  if (int r = read_flags(bands, bad)) return r;
  if (bad) return JOB_RERUN_CAREFUL;                          // host input is still untouched
  if (int r = land_all_if_no_copy(bands, false)) return r;
  for (Band& B : bands.b) {
    HIP_TRY_RESTORE(hipSetDevice(B.dev));
    std::vector<Piece> back;
    for (const BandPlane& P : B.planes)
      host_pieces(job, P.ci, P.r0 + P.own0, P.own_rows(), P.own_off(), back);
    HIP_TRY_RESTORE(B.down.finish(B.coef.p, back, B.s, B.stage.p != nullptr));
  }
  if (trace_on()) {
    // device time of the iterations per band (kernels + halo pulls + waiting for the neighbours), by HIP events
    float worst = 0;
    std::string per;
    for (Band& B : bands.b) {
      float ms = 0;
      (void)hipSetDevice(B.dev);
      if (hipEventSynchronize(B.evT1) == hipSuccess && hipEventElapsedTime(&ms, B.evT0, B.evT1) == hipSuccess) {
        worst = std::max(worst, ms);
        char buf[32]; snprintf(buf, sizeof buf, " %.2f", ms); per += buf;
      }
    }
    fprintf(stderr, "qs_hip trace: sharded(set) %d band(s)  upload+stage %.2f ms  enqueue %.2f ms  drain+scatter %.2f ms  "
                    "iterations on device: max %.2f ms (per band:%s)%s%s\n",
            n, t_up - t_start, t_enq - t_up, wall_ms() - t_enq, worst, per.c_str(), plan ? "  progress: callback served from this route" : "",
            deep ? "  schedule: deep halo, no exchange" : "  schedule: one halo row per iteration");
  }
  for (int ci = 0; ci < job->ncomp; ++ci)                    // reference :2851-2859
    if (job->has_quant[ci]) for (int i = 0; i < 64; ++i) job->quant[ci][i] = 1;
  return 0;
}

// ---------------------------------------------------------------------------
// coupled YCbCr job (JOINT_YUV and/or UPSAMPLE_UV, optionally LOW_QUALITY): the order of
// qs_job.cpp's general route, band by band.  aux[0] = L (luma at chroma resolution; the luma
// plane itself when luma is 1x1), aux[1..2] = upsampled pixel buffers, aux[3..4] = upsampled
// coefficient arrays.
static int run_sharded_colour(qs_hip_job* job, int flags, int niter, const std::vector<int>& devices) {
  const double t_start = wall_ms();
  const int ws = job->hsamp[0], hs = job->vsamp[0];
  const int hby = job->hblk[0], hbc = job->hblk[1];
  // bands are cut on chroma block rows; at least 8 of them per band
  const int n = std::max(1, std::min((int)devices.size(), hbc / 8));
  std::vector<int> devs(devices.begin(), devices.begin() + n);
  Bands bands((size_t)n);
  if (int r = open_bands(bands, devs)) return r;

  const bool lowq = (flags & QS_LOW_QUALITY) != 0;
  const bool sub = !(ws == 1 && hs == 1);                    // L is a separate, downsampled plane
  const bool upsample = (flags & QS_UPSAMPLE_UV) && sub;     // reference :2805 (image1 only when subsampled)
  const bool joint = (flags & QS_JOINT_YUV) != 0;
  const int plane_flags = flags & (QS_DIAGONALS | QS_NO_REBALANCE | QS_NO_REBALANCE_UV);
  const int diag = (plane_flags & QS_DIAGONALS) != 0;

  for (int d = 0; d < n; ++d) {
    Band& B = bands.b[d];
    int c0, c1, y0, y1;
    if (int r = qs_hip_colour_band_rows(hby, hbc, hs, n, d, &y0, &y1, &c0, &c1)) return r;
    for (int ci = 0; ci < 3; ++ci) {
      BandPlane P;
      P.ci = ci; P.wb = job->wblk[ci];
      P.r0 = ci ? c0 : y0; P.hb = ci ? c1 - c0 : y1 - y0;
      P.halo_top = d > 0; P.halo_bot = d < n - 1;
      B.planes.push_back(P);
    }
    HIP_TRY(hipSetDevice(B.dev));
    if (int r = stage_band(B, job, flags)) return r;
    if (sub) HIP_TRY(B.aux[0].alloc(qs_hip_plane_bytes(job->wblk[1], c1 - c0)));
  }
  const double t_up = wall_ms();

  // ref(B, ci).plane is the component's CURRENT pixel plane everywhere below; with the fused schedule ref.plane_next is
  // its second plane (stage_band allocates two per component) and the two swap after every pass B that wrote it
  auto plane_of = [&](int ci) {
    return [&bands, ci](int d, int, uint8_t** p, int* wb, int* hb) {
      const BandPlane& P = bands.b[d].planes[ci];
      *p = bands.b[d].set.ref[ci].plane; *wb = P.wb; *hb = P.hb;
      return true;
    };
  };
  auto ref = [](Band& B, int ci) -> QsPlaneRef& { return B.set.ref[ci]; };
  auto lowres = [&](Band& B) -> uint8_t* { return sub ? B.aux[0].as<uint8_t>() : ref(B, 0).plane; };
  // Fused schedule (recovery route, niter >= 1; the twin of bands.py: _run_colour_bands_fused): pass A once per
  // component; every pass B writes the plane the next stage reads -- the next iteration's, or the refresh the chroma
  // stages / the upsampling read -- and the +-1023 clamp rides on the last one (the fused IDCT sees the unclamped
  // coefficients, reference :2668-2689).  LOW_QUALITY and niter = 0 keep the unfused order below.
  const bool fuse = !lowq && niter > 0;
  for (Band& B : bands.b)
    for (int ci = 0; ci < 3; ++ci) ref(B, ci).plane_next = ref(B, ci).plane + plane_stride(B.planes[ci].wb, B.planes[ci].hb);
  // component ci of every band: one fused pass B (JOINT_YUV step first); write_next: it also writes the second plane
  auto fused_pass_b = [&](int ci, int final_clamp, bool write_next) -> int {
    for (size_t d = 0; d < bands.b.size(); ++d) {
      Band& B = bands.b[d];
      HIP_TRY(hipSetDevice(B.dev));
      const BandPlane& P = B.planes[ci];
      // (the plane about to be written is the one the neighbours pulled their halo rows from one pass earlier)
      if (write_next) if (int r = before_overwrite(bands, d, /*lag=*/1)) return r;
      if (ci && joint) qs_launch_joint(ref(B, ci).cst, ref(B, ci).coef, ref(B, ci).plane, lowres(B), P.wb, P.hb, 0, 0, B.s);
      qs_launch_smooth_plane(ref(B, ci).cst, ref(B, ci).coef, ref(B, ci).plane, write_next ? ref(B, ci).plane_next : nullptr,
                             !P.halo_top, !P.halo_bot, P.wb, P.hb, diag, comp_rebalance(job, ci, flags), final_clamp,
                             0, P.wb * P.hb, B.s);
      if (write_next) std::swap(ref(B, ci).plane, ref(B, ci).plane_next);
    }
    return QS_HIP_OK;
  };
  auto first_pass_a = [&](int ci) -> int {
    for (size_t d = 0; d < bands.b.size(); ++d) {
      Band& B = bands.b[d];
      HIP_TRY(hipSetDevice(B.dev));
      if (int r = before_overwrite(bands, d)) return r;
      const BandPlane& P = B.planes[ci];
      qs_launch_idct_plane(ref(B, ci).cst, ref(B, ci).coef, ref(B, ci).plane, P.wb, P.hb, 1,
                           !P.halo_top, !P.halo_bot, ref(B, ci).status, B.s);
    }
    return QS_HIP_OK;
  };

  // ---- luma: niter iterations + the refresh pass that feeds the chroma stages
  if (fuse) {
    if (int r = first_pass_a(0)) return r;
    for (int it = 0; it < niter; ++it) {
      if (int r = exchange(bands, 1, plane_of(0))) return r;
      if (int r = fused_pass_b(0, it == niter - 1, true)) return r;
    }
    // the refresh of a subsampled luma only feeds the downsample, which stays inside the band
    if (!sub) if (int r = exchange(bands, 1, plane_of(0))) return r;
  }
  for (int it = 0; !fuse && it <= niter; ++it) {
    for (size_t d = 0; d < bands.b.size(); ++d) {
      Band& B = bands.b[d];
      HIP_TRY(hipSetDevice(B.dev));
      if (int r = before_overwrite(bands, d)) return r;
      const BandPlane& P = B.planes[0];
      qs_launch_idct_plane(ref(B, 0).cst, ref(B, 0).coef, ref(B, 0).plane, P.wb, P.hb, it == 0,
                           !P.halo_top, !P.halo_bot, ref(B, 0).status, B.s);
    }
    const bool refresh = it == niter;
    // the refresh pass of a subsampled luma only feeds the downsample, which stays inside the band
    if (!refresh || !sub)
      if (int r = exchange(bands, 1, plane_of(0))) return r;
    if (refresh) break;
    for (Band& B : bands.b) {
      HIP_TRY(hipSetDevice(B.dev));
      const BandPlane& P = B.planes[0];
      if (lowq)
        qs_launch_lowq(ref(B, 0).cst, ref(B, 0).coef, ref(B, 0).plane, P.wb, P.hb, comp_rebalance(job, 0, flags), 0,
                       2.0f * sqrtf(0.5f), B.s);
      else
        qs_launch_smooth_plane(ref(B, 0).cst, ref(B, 0).coef, ref(B, 0).plane, nullptr, 1, 1, P.wb, P.hb, diag,
                               comp_rebalance(job, 0, flags), 0, 0, P.wb * P.hb, B.s);
    }
  }
  // the +-1023 clamp comes after the refresh pass (reference :2668-2689 sits after the loop)
  for (Band& B : bands.b) {
    HIP_TRY(hipSetDevice(B.dev));
    const BandPlane& P = B.planes[0];
    if (!fuse) qs_launch_clamp(ref(B, 0).coef, (size_t)P.wb * P.hb, B.s);
    if (sub)
      qs_launch_downsample(ref(B, 0).plane, P.wb, P.hb, B.aux[0].as<uint8_t>(), B.planes[1].wb, B.planes[1].hb, ws, hs, B.s);
  }
  if (sub)
    if (int r = exchange(bands, 1, [&](int d, int, uint8_t** p, int* wb, int* hb) {
          *p = bands.b[d].aux[0].as<uint8_t>(); *wb = bands.b[d].planes[1].wb; *hb = bands.b[d].planes[1].hb;
          return true;
        })) return r;

  // ---- chroma
  const int w1 = (job->image_width + ws - 1) / ws, h1_img = (job->image_height + hs - 1) / hs;
  const size_t up_pitch = qs_hip_upsample_pitch(job->image_width, ws);
  for (int ci = 1; ci <= 2; ++ci) {
    const int extra = upsample ? 1 : 0;
    if (fuse) {
      if (int r = first_pass_a(ci)) return r;
      for (int it = 0; it < niter; ++it) {
        if (int r = exchange(bands, 1, plane_of(ci))) return r;
        if (int r = fused_pass_b(ci, it == niter - 1, it < niter - 1 || extra)) return r;
      }
      if (extra) if (int r = exchange(bands, 1, plane_of(ci))) return r;   // the refreshed plane's halo: the upsampling's 3x3 windows
    }
    for (int it = 0; !fuse && it < niter + extra; ++it) {
      for (size_t d = 0; d < bands.b.size(); ++d) {
        Band& B = bands.b[d];
        HIP_TRY(hipSetDevice(B.dev));
        if (int r = before_overwrite(bands, d)) return r;
        const BandPlane& P = B.planes[ci];
        qs_launch_idct_plane(ref(B, ci).cst, ref(B, ci).coef, ref(B, ci).plane, P.wb, P.hb, it == 0,
                             !P.halo_top, !P.halo_bot, ref(B, ci).status, B.s);
      }
      if (int r = exchange(bands, 1, plane_of(ci))) return r;
      if (it == niter) break;
      const int last = (it == niter - 1) && !extra;
      for (Band& B : bands.b) {
        HIP_TRY(hipSetDevice(B.dev));
        const BandPlane& P = B.planes[ci];
        const int reb = comp_rebalance(job, ci, flags);
        if (lowq) {
          if (joint) qs_launch_joint(ref(B, ci).cst, ref(B, ci).coef, ref(B, ci).plane, lowres(B), P.wb, P.hb, reb, last, B.s);
          else qs_launch_lowq(ref(B, ci).cst, ref(B, ci).coef, ref(B, ci).plane, P.wb, P.hb, reb, last, 2.0f * sqrtf(0.5f), B.s);
        } else {
          if (joint) qs_launch_joint(ref(B, ci).cst, ref(B, ci).coef, ref(B, ci).plane, lowres(B), P.wb, P.hb, 0, 0, B.s);
          qs_launch_smooth_plane(ref(B, ci).cst, ref(B, ci).coef, ref(B, ci).plane, nullptr, 1, 1, P.wb, P.hb, diag, reb, last,
                                 0, P.wb * P.hb, B.s);
        }
      }
    }
    if (extra && !fuse)                                      // as for luma: clamp after the refresh
      for (Band& B : bands.b) {
        HIP_TRY(hipSetDevice(B.dev));
        qs_launch_clamp(ref(B, ci).coef, (size_t)B.planes[ci].wb * B.planes[ci].hb, B.s);
      }
    if (upsample)
      for (Band& B : bands.b) {
        HIP_TRY(hipSetDevice(B.dev));
        const BandPlane &Y = B.planes[0], &C = B.planes[ci];
        const int px0 = C.r0 * 8;                               // first low-res pixel row of the band (image coordinates)
        const int h1 = std::max(0, std::min(h1_img - px0, C.hb * 8));
        const int first_rows = std::max(0, std::min(8 - px0, h1));
        HIP_TRY(B.aux[ci].alloc(up_pitch * ((size_t)Y.hb * 8 + 8 * hs) + 64));
        HIP_TRY(B.aux[2 + ci].alloc((size_t)Y.wb * Y.hb * 128));
        HIP_TRY(hipMemsetAsync(B.aux[ci].p, 0, up_pitch * ((size_t)Y.hb * 8 + 8 * hs) + 64, B.s));
        qs_launch_upsample(ref(B, ci).plane, lowres(B), C.wb, ref(B, 0).plane, Y.wb, B.aux[ci].as<uint8_t>(), (int)up_pitch,
                           Y.wb * 8, Y.hb * 8, w1, h1, first_rows, ws, hs, B.s);
        qs_launch_fdct_plane(B.aux[ci].as<uint8_t>(), (int)up_pitch, B.aux[2 + ci].as<int16_t>(), Y.wb, Y.hb, B.s);
      }
  }

  // ---- results into pinned memory behind the kernels
  const size_t up_row = (size_t)job->wblk[0] * 128;
  for (Band& B : bands.b) {
    HIP_TRY(hipSetDevice(B.dev));
    HIP_TRY(hipGetLastError());
    if (!B.hstatus.alloc(3 * sizeof(int32_t))) return qs_fail(QS_HIP_ENOMEM, "out of pinned host memory");
    HIP_TRY(hipMemcpyAsync(B.hstatus.p, B.status.p, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, B.s));
    size_t coef_bytes = 0;
    for (const BandPlane& P : B.planes) coef_bytes += P.cbytes;
    HIP_TRY(B.down.issue(B.coef.p, coef_bytes, B.s, rows_active()));
    if (upsample)
      for (int j = 0; j < 2; ++j)
        HIP_TRY(B.down_up[j].issue(B.aux[3 + j].p, (size_t)B.planes[0].hb * up_row, B.s));
  }
  const double t_enq = wall_ms();

  bool bad = false;
  if (int r = read_flags(bands, bad)) return r;
  if (bad) return JOB_RERUN_CAREFUL;
  int16_t* up_host[2] = {nullptr, nullptr};
  if (upsample)
    for (int j = 0; j < 2; ++j) {
      up_host[j] = static_cast<int16_t*>(malloc((size_t)hby * up_row));
      if (!up_host[j]) { free(up_host[0]); return qs_fail(QS_HIP_ENOMEM, "out of host memory"); }
    }
  struct UpFree { int16_t** p; bool keep; ~UpFree() { if (!keep) { free(p[0]); free(p[1]); } } } up_free{up_host, false};
  if (int r = land_all_if_no_copy(bands, upsample)) return r;
  for (Band& B : bands.b) {
    HIP_TRY_RESTORE(hipSetDevice(B.dev));
    std::vector<Piece> back;
    for (const BandPlane& P : B.planes)
      host_pieces(job, P.ci, P.r0, P.hb, P.coef_off, back);
    HIP_TRY_RESTORE(B.down.finish(B.coef.p, back, B.s, B.stage.p != nullptr));
    if (upsample)
      for (int j = 0; j < 2; ++j) {
        const BandPlane& Y = B.planes[0];
        HIP_TRY_RESTORE(B.down_up[j].finish(B.aux[3 + j].p,
                                    std::vector<Piece>{{up_host[j] + (size_t)Y.r0 * Y.wb * 64, 0, (size_t)Y.hb * up_row}}, B.s));
      }
  }
  if (trace_on())
    fprintf(stderr, "qs_hip trace: sharded(colour) %d band(s)  upload+stage %.2f ms  enqueue %.2f ms  drain+scatter %.2f ms\n",
            n, t_up - t_start, t_enq - t_up, wall_ms() - t_enq);
  if (upsample) {                                            // reference :2836-2849
    job->coef_up[0] = up_host[0]; job->coef_up[1] = up_host[1]; up_free.keep = true;
    job->up_wblk = job->wblk[0]; job->up_hblk = job->hblk[0];
    job->out_hsamp0 = job->out_vsamp0 = 1;
  }
  for (int ci = 0; ci < job->ncomp; ++ci)                    // reference :2851-2859
    if (job->has_quant[ci]) for (int i = 0; i < 64; ++i) job->quant[ci][i] = 1;
  return 0;
}

// ---- configuration ---------------------------------------------------------
std::mutex g_cfg_mu;
bool g_cfg_set = false;                // qs_hip_set_devices was called
std::vector<int> g_cfg_devices;

static std::vector<int> parse_devices(const char* v) {
  std::vector<int> out;
  if (!v || !*v) return out;
  const int visible = qs_hip_device_count();
  if (!strcmp(v, "all")) { for (int i = 0; i < visible; ++i) out.push_back(i); return out; }
  for (const char* p = v; *p;) {
    char* end = nullptr;
    const long d = strtol(p, &end, 10);
    if (end == p) break;
    out.push_back((int)d);
    p = *end == ',' ? end + 1 : end;
    if (*end && *end != ',') break;
  }
  return out;
}

// the coupled route handles exactly the YCbCr layouts the general route couples
static bool colour_shardable(const qs_hip_job* job, int flags, int niter) {
  // (niter 0 = "upsample only": left to the general route, which knows the reference's corner cases)
  if (niter < 1 || !job_needs_lowres(job, flags)) return false;
  for (int ci = 0; ci < 3; ++ci) {
    if (!job->has_quant[ci]) return false;
    int acc = 0;
    for (int i = 0; i < 64; ++i) acc |= job->quant[ci][i];
    if (acc <= 1 || acc >= 0x800) return false;            // iterations skipped / stop: the general path knows how
  }
  // chroma planes must cover the luma band rows exactly: hblk[0] <= hblk[1] * v_samp (always true for
  // libjpeg's geometry) and both chroma components alike
  return job->wblk[1] == job->wblk[2] && job->hblk[1] == job->hblk[2] &&
         job->hblk[0] <= job->hblk[1] * job->vsamp[0] && job->hblk[0] > (job->hblk[1] - 1) * job->vsamp[0];
}

}  // namespace

std::vector<int> qsj::configured_devices() {
  std::vector<int> devs;
  std::lock_guard<std::mutex> lk(g_cfg_mu);
  if (g_cfg_set) return g_cfg_devices;
  // QS_HIP_DEVICES is read once per process (getenv per call would race with a host application's setenv)
  static const std::vector<int> from_env = [] { const char* v = getenv("QS_HIP_DEVICES"); return v ? parse_devices(v) : std::vector<int>(); }();
  if (!from_env.empty()) return from_env;
  // Default: the caller's current device only.  A process (or thread) per GPU is the common deployment;
  // spreading over every visible GPU is something the caller opts into (qs_hip_set_devices, or
  // QS_HIP_DEVICES=all / a list).
  devs.push_back(current_device());
  return devs;
}

std::vector<int> qsj::shard_devices_for(const qs_hip_job* job, int flags, int niter) {
  std::vector<int> devs = configured_devices();
  if (devs.size() < 2) return {};
  const size_t min_blocks = env_size("QS_HIP_SHARD_MIN_BLOCKS", (size_t)512 << 10);   // (read per call: tests lower it)
  size_t blocks = 0;
  for (int ci = 0; ci < job->ncomp; ++ci) blocks += (size_t)job->wblk[ci] * job->hblk[ci];
  if (blocks < min_blocks) return {};
  if (!(job_fusable(job, flags) || colour_shardable(job, flags, niter))) return {};
  return devs;
}

// ---------------------------------------------------------------------------
// One band per PROCESS: the halo rows travel through RCCL ("ncclGroupStart; ncclSend / ncclRecv
// x <= 2; ncclGroupEnd" between the iterations).  The library does not link librccl: the caller, who created the
// communicator, has it loaded; its five entry points are looked up in that copy (dlopen with RTLD_NOLOAD first).
namespace {
struct Rccl {
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl x;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);              // the copy the caller's communicator lives in
      if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) return x;
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(h, "ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    x.Send = reinterpret_cast<decltype(x.Send)>(dlsym(h, "ncclSend"));
    x.Recv = reinterpret_cast<decltype(x.Recv)>(dlsym(h, "ncclRecv"));
    x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(h, "ncclAllReduce"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    x.ok = x.GroupStart && x.GroupEnd && x.Send && x.Recv && x.AllReduce;
    return x;
  }();
  return r;
}
#define RCCL_TRY(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) \
  return qs_fail(QS_HIP_ENODEV, "%s failed: %s", #expr, R.GetErrorString ? R.GetErrorString(r_) : "RCCL error"); } while (0)
}  // namespace

extern "C" int qs_hip_do_quantsmooth_band(qs_hip_job* job, int flags, int niter, int rank, int nranks, void* nccl_comm) {
  if (!job || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !nccl_comm))
    return qs_fail(QS_HIP_EINVAL, "qs_hip_do_quantsmooth_band: bad rank / communicator");
  RowScope rows(nullptr);
  const int todo = prepare_job(job, flags, &niter);
  if (todo <= 0) return todo;
  if (!job_fusable(job, flags))
    return qs_fail(QS_HIP_ENOTSUP, "qs_hip_do_quantsmooth_band: independent components only (no JOINT_YUV / UPSAMPLE_UV / LOW_QUALITY, ordinary tables)");
  warm_wait();
  if (qs_hip_device_count() <= 0) return qs_fail(QS_HIP_ENODEV, "no HIP device available (this library has no CPU fallback)");
  const Rccl& R = rccl();
  if (nranks > 1 && !R.ok) return qs_fail(QS_HIP_ENODEV, "qs_hip_do_quantsmooth_band: librccl.so.1 is not loadable");
  const ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);

  struct Restore { int dev; ~Restore() { (void)hipSetDevice(dev); } } restore{current_device()};
  Bands bands(1);
  if (int r = open_bands(bands, std::vector<int>{current_device()})) return r;
  Band& B = bands.b[0];
  for (int ci = 0; ci < job->ncomp; ++ci) {                  // the job IS the band: every component's rows of this rank
    BandPlane P;
    P.ci = ci; P.wb = job->wblk[ci]; P.hb = job->hblk[ci]; P.r0 = 0;
    P.halo_top = rank > 0; P.halo_bot = rank < nranks - 1;
    B.planes.push_back(P);
  }
  if (int r = stage_band(B, job, flags)) return r;
  const int diag = (flags & QS_DIAGONALS) != 0;
  const int np = (int)B.planes.size();
  for (int it = 0; it < niter; ++it) {
    const int cur = it & 1;
    if (it == 0) qs_launch_idct_set(B.set, 1, B.s);
    for (int i = 0; i < np; ++i) {
      uint8_t* a = B.px.as<uint8_t>() + B.planes[i].px_off;
      uint8_t* b = a + plane_stride(B.planes[i].wb, B.planes[i].hb);
      B.set.ref[i].plane = cur ? b : a;
      B.set.ref[i].plane_next = it == niter - 1 ? nullptr : cur ? a : b;
    }
    if (nranks > 1) {
      // the exchange of reference-equivalent data: one pixel row per component and band edge, of the planes the coming
      // pass B reads; all sends and receives of the iteration in ONE group on the band's stream
      RCCL_TRY(R.GroupStart());
      for (int i = 0; i < np; ++i) {
        size_t send_top, send_bot, recv_top, recv_bot, pitch;
        (void)qs_hip_band_halo_rows(B.planes[i].wb, B.planes[i].hb, &send_top, &send_bot, &recv_top, &recv_bot, &pitch);
        uint8_t* p = B.set.ref[i].plane;
        if (rank > 0) {
          RCCL_TRY(R.Send(p + send_top, pitch, ncclUint8, rank - 1, comm, B.s));
          RCCL_TRY(R.Recv(p + recv_top, pitch, ncclUint8, rank - 1, comm, B.s));
        }
        if (rank < nranks - 1) {
          RCCL_TRY(R.Send(p + send_bot, pitch, ncclUint8, rank + 1, comm, B.s));
          RCCL_TRY(R.Recv(p + recv_bot, pitch, ncclUint8, rank + 1, comm, B.s));
        }
      }
      RCCL_TRY(R.GroupEnd());
    }
    qs_launch_smooth_set(B.set, diag, it == niter - 1, B.s);
  }
  HIP_TRY(hipGetLastError());
  // the range check is the JOB's: any rank's tripped flag is everybody's (reference :2599-2610)
  if (nranks > 1) RCCL_TRY(R.AllReduce(B.status.p, B.status.p, (size_t)np, ncclInt32, ncclMax, comm, B.s));
  if (!B.hstatus.alloc((size_t)np * sizeof(int32_t))) return qs_fail(QS_HIP_ENOMEM, "out of pinned host memory");
  HIP_TRY(hipMemcpyAsync(B.hstatus.p, B.status.p, (size_t)np * sizeof(int32_t), hipMemcpyDeviceToHost, B.s));
  size_t coef_bytes = 0;
  for (const BandPlane& P : B.planes) coef_bytes += P.cbytes;
  HIP_TRY(B.down.issue(B.coef.p, coef_bytes, B.s, false));
  bool bad = false;
  if (int r = read_flags(bands, bad)) return r;
  if (bad) {                                                 // nothing was written on any rank
    HIP_TRY(hipStreamSynchronize(B.s));
    return QS_HIP_BAND_RANGE_CHECK;
  }
  if (int r = land_all_if_no_copy(bands, false)) return r;
  {
    std::vector<Piece> back;
    for (const BandPlane& P : B.planes) host_pieces(job, P.ci, 0, P.hb, P.coef_off, back);
    HIP_TRY_RESTORE(B.down.finish(B.coef.p, back, B.s, B.stage.p != nullptr));
  }
  for (int ci = 0; ci < job->ncomp; ++ci)                    // reference :2851-2859
    if (job->has_quant[ci]) for (int i = 0; i < 64; ++i) job->quant[ci][i] = 1;
  return 0;
}

int qsj::run_sharded(qs_hip_job* job, int flags, int niter, const std::vector<int>& devices, ProgressPlan* plan) {
  if (devices.empty()) return qs_fail(QS_HIP_EINVAL, "sharded job: empty device list");
  if (devices.size() > 64) return qs_fail(QS_HIP_EINVAL, "sharded job: more than 64 bands");
  // the caller's current device is put back on every path
  struct Restore { int dev; ~Restore() { (void)hipSetDevice(dev); } } restore{current_device()};
  if (job_fusable(job, flags)) return run_sharded_set(job, flags, niter, devices, plan);
  if (colour_shardable(job, flags, niter)) return run_sharded_colour(job, flags, niter, devices);
  return qs_fail(QS_HIP_ENOTSUP, "sharded job: this flag / table combination runs on one device");
}

// schedule of the independent-component route: -1 = not set by the API (the environment decides)
static std::atomic<int> g_shard_schedule{-1};
bool qsj::shard_schedule_deep() {
  const int v = g_shard_schedule.load();
  if (v >= 0) return v == 1;
  const char* e = getenv("QS_HIP_SHARD_SCHEDULE");
  return e && (!strcmp(e, "deep") || !strcmp(e, "1"));
}
extern "C" int qs_hip_set_shard_schedule(int schedule) {
  if (schedule < -1 || schedule > 1) return qs_fail(QS_HIP_EINVAL, "qs_hip_set_shard_schedule: 0 = halo row per iteration, 1 = deep halo, -1 = default");
  g_shard_schedule.store(schedule);
  return QS_HIP_OK;
}

extern "C" int qs_hip_set_devices(const int* devices, int n) {
  if (n < 0 || (n > 0 && !devices)) return qs_fail(QS_HIP_EINVAL, "qs_hip_set_devices: bad argument");
  const int visible = qs_hip_device_count();
  for (int i = 0; i < n; ++i)
    if (devices[i] < 0 || devices[i] >= visible) return qs_fail(QS_HIP_EINVAL, "qs_hip_set_devices: no HIP device %d", devices[i]);
  std::lock_guard<std::mutex> lk(g_cfg_mu);
  g_cfg_set = n > 0;
  g_cfg_devices.assign(devices, devices + n);
  return QS_HIP_OK;
}
