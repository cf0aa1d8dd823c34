// qs_jobint.h -- what the translation units of the job layer share.  Not part of the ABI.
//   qs_job.cpp    entry points, validation / early-outs, the general per-component route (run_job), prewarm
//   qs_fused.cpp  the plane-set route (independent components; very large planes as pipelined bands)
//   qs_shard.cpp  one job over several devices (bands + halo rows pulled with peer copies)
//   qs_batch.cpp  qs_hip_do_quantsmooth_batch (many jobs per call; coupled groups)
#pragma once
#include <vector>

#include "qs_xfer.h"

namespace qsj {

enum { JOB_RERUN_CAREFUL = -1000 };   // internal result: run the job again in the reference's sequential order

double wall_ms();
void warm_wait();                      // block while a qs_hip_prewarm() thread is still at work
bool trace_on();                       // QS_HIP_TRACE=1: phase times on stderr
bool shard_schedule_deep();            // qs_hip_set_shard_schedule / QS_HIP_SHARD_SCHEDULE: the communication-avoiding band schedule
size_t env_size(const char* name, size_t dflt);

// reference quantsmooth.h:2639 + :1567-1568: does component ci get the rebalance step
int comp_rebalance(const qs_hip_job* job, int ci, int flags);
// reference :2447-2453: JOINT_YUV / UPSAMPLE_UV couple chroma to luma for this job
bool job_needs_lowres(const qs_hip_job* job, int flags);
// components independent of each other and ordinary quant tables: the plane-set routes apply
bool job_fusable(const qs_hip_job* job, int flags);

// Host side of a job's blocks.  Normally job->coef[ci] is one contiguous array; through
// qs_hip_do_quantsmooth_rows the caller hands over one pointer per block row instead (libjpeg's
// JBLOCKROWs), which the transfer helpers gather from / scatter to directly.  host_pieces appends
// the pieces for block rows [row0, row0 + nrows) of component ci, placed at `arena_off` of a device
// arena (adjacent rows are merged into one piece).  The row table is per calling thread and set
// for the duration of one entry-point call (RowScope).
void host_pieces(const qs_hip_job* job, int ci, int row0, int nrows, size_t arena_off, std::vector<qsx::Piece>& out);
bool rows_active();    // the current call came through qs_hip_do_quantsmooth_rows
struct RowScope {
  explicit RowScope(int16_t* const* const* rows);
  ~RowScope();
  RowScope(const RowScope&) = delete;
  RowScope& operator=(const RowScope&) = delete;
};

// general route on the current device (qs_job.cpp)
int run_job(qs_hip_job* job, int flags, int niter, int progprec,
            qs_hip_progress_fn progress, void* userdata, bool eager);

// The reference's progress protocol (quantsmooth.h:2474-2482, 2656-2664) on the pipelined routes.  Which calls the
// reference makes, and with which `cur`, depends on the geometry alone: after every pass B of every component, in
// component-major order, `prog_cur` grows by hblk * v_samp and a call is made when it crosses the running threshold.
// init() replays that arithmetic without calling anybody and keeps the list {work units done when the call is due,
// value reported}.  A pipelined route does the same work in another order (all components per launch, band after
// band); it reports the units it has COMPLETED (advance()), and every call whose share of the work is done is made,
// in the reference's sequence -- the callback sees the same calls with the same arguments, each when at least that
// fraction of the job is really finished.  A non-zero return cancels: the route drops its device state (nothing has
// reached the caller's arrays that cannot be put back) and the job is re-run from the untouched input in the
// reference's own order with replay() standing in for the callback: it answers the calls already made from the record
// (0, ..., 0, stop) without reaching the user a second time -- so the cancelled result is the reference's, bit for bit --
// and hands later calls through (the re-run after a tripped range check, which continues live).
struct ProgressPlan {
  struct Call { long long units; int cur; };
  std::vector<Call> calls;
  qs_hip_progress_fn fn = nullptr;
  void* userdata = nullptr;
  int progprec = 0;            // as passed by the caller (run_job normalises it itself)
  int progprec_eff = 0;        // the `max` argument of every call
  size_t made = 0;             // calls made so far
  bool cancelled = false;      // call number made - 1 returned non-zero
  size_t replayed = 0;         // replay(): calls answered from the record so far
  void init(const qs_hip_job* job, int niter, int progprec_arg, qs_hip_progress_fn f, void* ud);
  bool advance(long long units_done);          // true once cancelled
  static int replay(void* self, int cur, int max);
};

// plane-set route (qs_fused.cpp): jobs[which[*]] are fusable (job_fusable); results[ji] = what
// qs_hip_do_quantsmooth would have returned for that job.  Returns 0, or < 0 when the route failed (a job whose
// rows had already been written is then either complete, results[ji] == 0, or restored to its input).
int run_fused(qs_hip_job* const* jobs, const std::vector<int>& which, int flags, int niter, int* results,
              ProgressPlan* plan = nullptr);   // plan: single-job calls with a progress callback
// (prewarm) the buffer sizes run_fused will ask for, one entry per group, for this one job
void fused_stage_sizes(const qs_hip_job* job, int niter, std::vector<size_t>& coef_bytes, std::vector<size_t>& px_bytes);
// validation and the reference's early-outs (1: work to do, 0: finished with result 0, < 0: bad job), and
// the single-job dispatcher behind qs_hip_do_quantsmooth (qs_job.cpp)
int prepare_job(qs_hip_job* job, int flags, int* niter);
int do_quantsmooth_impl(qs_hip_job* job, int flags, int niter, int progprec, qs_hip_progress_fn progress, void* userdata);

// multi-device route (qs_shard.cpp).  `devices`: HIP ordinals, one entry per band; an ordinal
// may repeat (several logical devices on one GPU: how the route is tested on a one-GPU box).
// Returns the job result, JOB_RERUN_CAREFUL when the range check tripped (host input untouched),
// or < 0.
int run_sharded(qs_hip_job* job, int flags, int niter, const std::vector<int>& devices, ProgressPlan* plan = nullptr);
// the configured device list (qs_hip_set_devices / QS_HIP_DEVICES / all visible devices) when
// this job should be sharded, empty otherwise
std::vector<int> shard_devices_for(const qs_hip_job* job, int flags, int niter);
// the configured device list itself (qs_hip_set_devices, else QS_HIP_DEVICES, else all visible devices)
std::vector<int> configured_devices();

}  // namespace qsj
