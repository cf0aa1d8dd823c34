// qs_jobint.h -- what the translation units of the job layer share (qs_job.cpp: single-device
// routes and the C entry points; qs_shard.cpp: the multi-device route).  Not part of the ABI.
#pragma once
#include <vector>

#include "qs_xfer.h"

namespace qsj {

enum { JOB_RERUN_CAREFUL = -1000 };   // internal result: run the job again in the reference's sequential order

double wall_ms();
bool trace_on();                       // QS_HIP_TRACE=1: phase times on stderr
size_t env_size(const char* name, size_t dflt);

// reference quantsmooth.h:2639 + :1567-1568: does component ci get the rebalance step
int comp_rebalance(const qs_hip_job* job, int ci, int flags);
// reference :2447-2453: JOINT_YUV / UPSAMPLE_UV couple chroma to luma for this job
bool job_needs_lowres(const qs_hip_job* job, int flags);
// components independent of each other and ordinary quant tables: the plane-set routes apply
bool job_fusable(const qs_hip_job* job, int flags);

// general route on the current device (qs_job.cpp)
int run_job(qs_hip_job* job, int flags, int niter, int progprec,
            qs_hip_progress_fn progress, void* userdata, bool eager);

// multi-device route (qs_shard.cpp).  `devices`: HIP ordinals, one entry per band; an ordinal
// may repeat (several logical devices on one GPU: how the route is tested on a one-GPU box).
// Returns the job result, JOB_RERUN_CAREFUL when the range check tripped (host input untouched),
// or < 0.
int run_sharded(qs_hip_job* job, int flags, int niter, const std::vector<int>& devices);
// the configured device list (qs_hip_set_devices / QS_HIP_DEVICES / all visible devices) when
// this job should be sharded, empty otherwise
std::vector<int> shard_devices_for(const qs_hip_job* job, int flags, int niter);

}  // namespace qsj
