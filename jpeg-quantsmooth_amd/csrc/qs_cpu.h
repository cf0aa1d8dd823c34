/*
 * qs_cpu.h -- the CPU back end behind include/libjpegqs.h ("CPU fallback when no
 * HIP device").  Internal to libjpegqs.so (csrc/jpegqs_shim.c is the only caller); libjpegqs_hip.so -- the C ABI
 * of the GPU hot path -- has no CPU route and keeps failing with QS_HIP_ENODEV.
 *
 * Same job contract as qs_hip_do_quantsmooth() / qs_hip_do_quantsmooth_rows() of include/jpegqs_hip.h (the whole
 * of reference quantsmooth.h:2404-2878 on caller-owned host arrays): blocks rewritten in place, quant tables set
 * to 1, the UPSAMPLE_UV replacement arrays in job->coef_up[] (plain malloc here: release with qs_cpu_free()),
 * return value = the reference's `stop`, or a negative QS_HIP_E* code when the job is malformed.
 */
#ifndef QS_CPU_H
#define QS_CPU_H

#include "../../include/jpegqs_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* rows == NULL: job->coef[ci] are flat arrays; otherwise rows[ci][y] points at block row y (what libjpeg's
 * access_virt_barray hands out) and job->coef[ci] is ignored.  threads: jpegqs_control_t.threads
 * (reference quantsmooth.h:2467-2472: < 0 leave OpenMP alone, 0 one per processor, n exactly n). */
int qs_cpu_do_quantsmooth(qs_hip_job *job, int16_t *const *const *rows, int flags, int niter, int threads,
		int progprec, qs_hip_progress_fn progress, void *userdata);
void qs_cpu_free(void *p);
/* "avx512f" / "avx2" / "generic": which clone of the lane kernels this processor runs (for the --info 16 line) */
const char *qs_cpu_isa(void);
/* blocks processed side by side in one vector (the compile-time lane count) */
int qs_cpu_lanes(void);

#ifdef __cplusplus
}
#endif
#endif
