/*
 * jpegqs_hip.h -- flat C ABI of the MI355X (gfx950) implementation of
 * jpeg-quantsmooth's coefficient-recovery path.
 *
 * Two layers, both plain C (pointers + sizes, no libjpeg / torch types):
 *
 *  1. JOB LAYER  qs_hip_do_quantsmooth(): the whole of the reference's
 *     do_quantsmooth() (reference quantsmooth.h:2404-2878) on caller-owned HOST
 *     arrays -- same inputs (JCOEF blocks as jpeg_read_coefficients() returns
 *     them, quant tables, sampling factors, flags/niter/progress), same outputs
 *     (blocks rewritten in place, quant tables set to 1, return value = the
 *     reference's `stop`).  The libjpeg-facing drop-in (include/libjpegqs.h,
 *     csrc/jpegqs_shim.c) gathers JBLOCKROWs into these arrays and calls this.
 *
 *  2. PLANE LAYER  qs_hip_*_plane(): the individual GPU passes on DEVICE
 *     pointers and an explicit HIP stream, for callers that keep data resident
 *     in HBM (bench.py, the multi-GPU band driver, pipelines that decode on the
 *     GPU).  One call = one kernel launch, asynchronous on `stream`.
 *
 * All functions return 0 on success and a negative QS_HIP_E* code on failure;
 * qs_hip_last_error() gives the message.  There is no CPU fallback: without a
 * usable HIP device every compute entry point fails with QS_HIP_ENODEV.
 */
#ifndef JPEGQS_HIP_H
#define JPEGQS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QS_HIP_MAXC 4

enum {
	QS_HIP_OK = 0,
	QS_HIP_ENODEV = -1,   /* no HIP device / runtime error */
	QS_HIP_EINVAL = -2,   /* bad argument */
	QS_HIP_ENOMEM = -3,   /* host or device allocation failed */
	QS_HIP_ENOTSUP = -4   /* flags this entry point does not implement (use the one that does) */
};

/* ---- job layer ---------------------------------------------------------- */

/* One image, as do_quantsmooth() sees it (reference quantsmooth.h:2423-2427,
 * 2447-2451, 2488-2493).  Layout is shared with the test oracles. */
typedef struct {
	int32_t ncomp;                     /* cinfo->num_components, 1..4 */
	int32_t colorspace;                /* cinfo->jpeg_color_space (1 gray, 3 YCbCr, ...) */
	int32_t image_width, image_height; /* cinfo->image_width/height */
	int32_t wblk[QS_HIP_MAXC];         /* comp_info[ci].width_in_blocks */
	int32_t hblk[QS_HIP_MAXC];         /* comp_info[ci].height_in_blocks */
	int32_t hsamp[QS_HIP_MAXC];        /* comp_info[ci].h_samp_factor */
	int32_t vsamp[QS_HIP_MAXC];        /* comp_info[ci].v_samp_factor */
	int32_t has_quant[QS_HIP_MAXC];    /* comp_info[ci].quant_table != NULL */
	uint16_t quant[QS_HIP_MAXC][64];   /* quantval[], natural order; set to 1 on return */
	int16_t *coef[QS_HIP_MAXC];        /* hblk*wblk blocks of 64 JCOEF, in/out */
	/* UPSAMPLE_UV only: replacement chroma arrays at luma resolution
	 * (reference :2696-2703, 2836-2849); owned by the caller, release with qs_hip_free()
	 * (NOT free(): large arrays are pinned buffers of the library's pool) */
	int16_t *coef_up[2];
	int32_t up_wblk, up_hblk;          /* 0 when chroma was not replaced */
	int32_t out_hsamp0, out_vsamp0;    /* component 0 sampling factors on return */
} qs_hip_job;

/* opts->progress of reference libjpegqs.h:41-45 */
typedef int (*qs_hip_progress_fn)(void *userdata, int cur, int max);

/* flags: JPEGQS_* algorithm bits (reference libjpegqs.h:16-23); niter, progprec,
 * progress, userdata: the jpegqs_control_t fields of the same names.
 * Returns the reference's `stop` (0 done, 1 cancelled/rejected input) or <0. */
int qs_hip_do_quantsmooth(qs_hip_job *job, int flags, int niter, int progprec,
		qs_hip_progress_fn progress, void *userdata);
/* The same job with every component's blocks given as separate ROWS: rows[ci][y] points at
 * wblk[ci] consecutive blocks of block row y -- what libjpeg's access_virt_barray hands out
 * (reference quantsmooth.h:2557-2560) -- and job->coef[ci] is ignored.  The rows are read and,
 * on success, rewritten in place; the library's helper threads gather them into / scatter them
 * from its pinned staging memory, so a libjpeg-facing caller needs no intermediate copy
 * (csrc/jpegqs_shim.c).  Rows must not overlap. */
int qs_hip_do_quantsmooth_rows(qs_hip_job *job, int16_t *const *const *rows, int flags, int niter,
		int progprec, qs_hip_progress_fn progress, void *userdata);
/* The same for many jobs in one call (one flags/niter setting for all).  Jobs whose
 * components are independent of each other (no JOINT_YUV / UPSAMPLE_UV coupling, no
 * LOW_QUALITY: CLI --quality 3 and 4) are processed TOGETHER: one pass-A and one
 * pass-B launch per iteration over all their planes, so small images fill the 256 CUs
 * as a group; YCbCr jobs coupled by JOINT_YUV / UPSAMPLE_UV (--quality 5 and 6) advance in groups
 * of about 200k blocks (all luma planes as one launch per pass, then all chroma planes), the rest
 * goes through qs_hip_do_quantsmooth -- groups and single jobs up to four at a time from helper
 * threads (their error text is not kept: results[i] carries the code).  results[i] =
 * what qs_hip_do_quantsmooth would have returned for jobs[i].  Returns 0, or < 0 when
 * the batch as a whole could not run (bad arguments, no device).  With several devices
 * configured (qs_hip_set_devices / QS_HIP_DEVICES) the jobs of a batch are spread
 * over them as whole jobs, balanced by block count: independent objects, no exchange.  Not part
 * of the reference API: an addition for callers that serve many images. */
int qs_hip_do_quantsmooth_batch(qs_hip_job *const *jobs, int njobs, int flags, int niter, int *results);

/* ---- several GPUs, one host process (the reference's counterpart is the
 * OpenMP row split INSIDE do_quantsmooth, reference quantsmooth.h:2587-2640) ----
 * A job of at least 512k blocks (QS_HIP_SHARD_MIN_BLOCKS) without a progress callback is cut
 * into one block-row band per device; each band stays on its GPU for the whole job and pulls one
 * pixel row per component from each neighbouring band after every pass A (hipMemcpyPeerAsync over
 * xGMI).  Bit-exact with the one-device result.  Both the independent-component flags (CLI
 * --quality 3/4) and the coupled YCbCr flags (--quality 5/6) are covered; anything else runs on
 * the current device.  OPT-IN: by default everything runs on the caller's current HIP device (a process
 * or thread per GPU is the usual deployment and must not find its jobs on other workers' GPUs).  The device
 * list is given by qs_hip_set_devices(), else by the environment variable QS_HIP_DEVICES ("all", or ordinals
 * such as "0,1,2,3"; read once per process); fewer than two entries = no sharding.  An ordinal may repeat (several bands on one GPU: how the route is
 * tested on a one-GPU box).  n = 0 returns to the default. */
int qs_hip_set_devices(const int *devices, int n);
/* How the bands of an independent-component job (CLI --quality 3/4) keep each other exact:
 *   0 (default)  one pixel row per band edge is exchanged after every pass A (niter small, latency-bound transfers);
 *   1            COMMUNICATION-AVOIDING: every cut side of a band carries `niter` block rows of its neighbour and
 *                all iterations run without any exchange -- the error made by treating the cut as an image edge
 *                moves one block row per iteration and never reaches the rows the band owns.  Costs 2 * niter block
 *                rows of extra work per inner band (+4.7 % at 8192 x 8192 over 8 devices, niter 3; +2.3 % at 16384^2).
 *  -1            back to the default (the environment variable QS_HIP_SHARD_SCHEDULE=deep also selects 1).
 * Both give the one-device result bit for bit.  Coupled YCbCr jobs (--quality 5/6) always use 0.
 * The reference's counterpart: the OpenMP row split of quantsmooth.h:2587-2640, which shares one pixel plane. */
int qs_hip_set_shard_schedule(int schedule);
/* the same job, cut over exactly these devices whatever its size (QS_HIP_ENOTSUP when the
 * flag / table combination has no sharded route; a progress callback is not available here) */
int qs_hip_do_quantsmooth_sharded(qs_hip_job *job, int flags, int niter, const int *devices, int ndev);

/* ---- several GPUs, ONE PROCESS PER GPU: the halo rows travel through RCCL ----
 * `job` is THIS rank's band: for every component the block rows the rank owns (hblk[ci] = rows of the band, cut with
 * qs_hip_band_rows so that the ranks' bands tile the image from top = rank 0 to bottom = rank nranks - 1), quant tables
 * and geometry as for qs_hip_do_quantsmooth.  After pass A and between the iterations the band sends its first / last
 * pixel row per component to rank - 1 / rank + 1 and receives theirs into its apron rows -- one ncclGroupStart /
 * ncclSend / ncclRecv (<= 4 per component) / ncclGroupEnd per iteration on the band's stream, no host synchronisation.
 * nccl_comm: the caller's ncclComm_t (RCCL), ranks numbered in band order; may be NULL when nranks == 1.  The library
 * does not link librccl: it uses the copy already loaded in the caller's process.
 * Independent components only (CLI --quality 3/4; QS_HIP_ENOTSUP otherwise), no progress callback.
 * Returns 0, a negative QS_HIP_E* code, or QS_HIP_BAND_RANGE_CHECK when a coefficient failed the reference's range check
 * on SOME rank (the flag is all-reduced): then no rank has written anything, and the image has to go through
 * qs_hip_do_quantsmooth as a whole, which applies the reference's stop semantics (quantsmooth.h:2599-2610).
 * The reference's counterpart is the OpenMP row split of quantsmooth.h:2587-2640, whose threads share the pixel plane.
 * (The communication-avoiding alternative needs no entry point: hand qs_hip_do_quantsmooth a band extended by niter
 *  block rows per cut side and keep the owned rows -- jpeg-quantsmooth_amd/bands.py: deep_band_rows.) */
#define QS_HIP_BAND_RANGE_CHECK 2
int qs_hip_do_quantsmooth_band(qs_hip_job *job, int flags, int niter, int rank, int nranks, void *nccl_comm);

/* The band arithmetic itself -- ONE definition, used by the in-process route above (csrc/qs_shard.cpp: halo rows
 * pulled with hipMemcpyPeerAsync) and by the one-process-per-GPU driver (bands.py: the same rows sent with RCCL):
 * block rows [*row0, *row1) of band `band` of `nbands` (edges on multiples of `align` block rows);
 * the luma / chroma row ranges of a band of a coupled YCbCr job (cut on chroma block rows, luma = the v_samp
 * times taller range of the same image rows; the last band takes the remaining luma rows);
 * and the byte offsets, inside a band's pixel plane, of the two rows it sends (first / last pixel row) and of
 * the two apron rows it receives into, each *nbytes long (reference quantsmooth.h:1396-1401 is what reads them). */
int qs_hip_band_rows(int hblk, int nbands, int band, int align, int *row0, int *row1);
int qs_hip_colour_band_rows(int hblk_luma, int hblk_chroma, int v_samp, int nbands, int band,
		int *y0, int *y1, int *c0, int *c1);
int qs_hip_band_halo_rows(int wblk, int hblk, size_t *send_top, size_t *send_bot,
		size_t *recv_top, size_t *recv_bot, size_t *nbytes);

/* The progress calls a job will make, as a function of its geometry (no device needed): the reference calls
 * progress(userdata, cur, max) after a pass B whenever its running block-row count crosses a threshold
 * (quantsmooth.h:2474-2482, 2656-2664); the pipelined routes of this library make exactly these calls, in this order,
 * each when at least that share of the work has completed.  Fills cur_out[0 .. min(n, max_calls)) and *max_out (the
 * `max` argument of every call) for a job whose components all run `niter` iterations (ordinary quant tables, no
 * cross-component flags); returns the number of calls n. */
int qs_hip_progress_calls(const qs_hip_job *geometry, int niter, int progprec, int *cur_out, int max_calls, int *max_out);

/* Optional, returns at once: bring the GPU side up IN THE BACKGROUND while the caller is still busy with something
 * else -- typically libjpeg's entropy decoding between jpeg_read_header() and jpeg_read_coefficients().  A fresh
 * process otherwise pays for the HIP runtime, the device context, the code object and the pinned staging buffers
 * inside its first do_quantsmooth (about 0.1 s for a full-HD image, 0.2-0.3 s for 8192 x 8192).  geometry: a job
 * whose ncomp / colorspace / wblk / hblk / hsamp / vsamp / has_quant / quant fields describe the coming call (coef
 * pointers are ignored; quant tables only decide the route), or NULL to start the runtime only.  The first job-layer
 * call waits for a prewarm still in flight.  Not in the reference API. */
int qs_hip_prewarm(const qs_hip_job *geometry, int flags, int niter);

void qs_hip_free(void *p);
/* the job layer keeps freed device buffers (up to 6 GiB per device), pinned staging buffers (up
 * to 2 GiB) and HIP streams in process-wide caches, each entry tied to the device it was created
 * on and handed out only to callers whose current HIP device is that one (a host thread may
 * hipSetDevice() to any GPU before calling the job layer); this returns them to the driver.
 * qs_hip_do_quantsmooth_batch keeps at most QS_HIP_GROUP_WINDOW (6) groups of about 200k blocks
 * in flight, so its memory does not grow with the size of the batch.  (It also
 * starts eight helper threads on first use, for the host side of large transfers; they
 * sleep between jobs and live until the process ends.) */
void qs_hip_release_cache(void);

/* ---- plane layer (device pointers, async on `stream`) --------------------- */

int qs_hip_device_count(void);
const char *qs_hip_last_error(void);
/* Version of this interface: bumped whenever a struct layout or the meaning of an argument changes (5: round 5 --
 * qs_hip_plane_ref back to its 48-byte form, second planes through qs_hip_smooth_planes_next; 6: round 6 --
 * additions only: qs_hip_set_shard_schedule, the RCCL band entry points).  A caller built against
 * this header can compare QS_HIP_ABI_VERSION with what the loaded library reports. */
#define QS_HIP_ABI_VERSION 6
int qs_hip_abi_version(void);

/* bytes of the per-component constant block; pixel-plane pitch and size */
size_t qs_hip_consts_bytes(void);
size_t qs_hip_plane_pitch(int wblk);
size_t qs_hip_plane_bytes(int wblk, int hblk);
/* byte offset of pixel (x = -QS apron .. , y) helpers for halo exchange:
 * row y (y = -1 .. hblk*8) of the plane starts at qs_hip_plane_row_offset(wblk, y)
 * and is qs_hip_plane_pitch(wblk) bytes long (apron columns included). */
size_t qs_hip_plane_row_offset(int wblk, int y);

/* Build the constant block for one component on the host (quant-derived values,
 * reference :2497-2540, and the weight tables, reference :251-301) into
 * `host_out` (qs_hip_consts_bytes() bytes); the caller copies it to the device. */
int qs_hip_consts_build(void *host_out, const uint16_t quant[64], int flags);

/* pass A: [first: dequantise + range check ->*d_status |= 1] IDCT into the plane
 * (reference :2589-2620).  rep_top/rep_bot: fill the y=-1 / y=h apron rows by
 * replication (0 when they are halo rows owned by a neighbouring band). */
int qs_hip_idct_plane(const void *d_consts, int16_t *d_coef, uint8_t *d_plane,
		int wblk, int hblk, int first, int rep_top, int rep_bot,
		int32_t *d_status, void *stream);

/* pass B: per-block recovery loop + rebalance (+ final clamp)
 * (reference :2627-2640 -> quantsmooth_block :564-1849; clamp :2668-2689).
 * Supported here: flags & (DIAGONALS | NO_REBALANCE | NO_REBALANCE_UV). */
int qs_hip_smooth_plane(const void *d_consts, int16_t *d_coef, const uint8_t *d_plane,
		int wblk, int hblk, int flags, int luma, int final_clamp, void *stream);
/* pass B that ALSO writes the pixel plane of the next iteration (pass A of iteration n + 1 fused into pass B of
 * iteration n: the kernel holds the block's final coefficients anyway).  d_plane_next must be a second plane of the
 * same geometry -- the other blocks of the launch still read d_plane -- and receives exactly what
 * qs_hip_idct_plane(first = 0, rep_top, rep_bot) would write after this call.  Reference :2589-2620 + :2627-2640. */
int qs_hip_smooth_plane_next(const void *d_consts, int16_t *d_coef, const uint8_t *d_plane, uint8_t *d_plane_next,
		int wblk, int hblk, int flags, int luma, int final_clamp, int rep_top, int rep_bot, void *stream);
/* the same for block rows [row0, row1) only: a band runs its interior rows while
 * the halo rows are still in flight and its first/last row afterwards */
int qs_hip_smooth_rows(const void *d_consts, int16_t *d_coef, const uint8_t *d_plane,
		int wblk, int hblk, int row0, int row1, int flags, int luma, int final_clamp, void *stream);

/* pass A / pass B over a SET of whole planes in one launch (any mix of sizes and quant
 * tables: the components of a job, or of many small jobs).  luma: as in
 * qs_hip_smooth_plane; d_status: the plane's range-check flag (pass A, first iteration). */
#define QS_HIP_MAX_PLANES 56
typedef struct {
	const void *d_consts;
	int16_t *d_coef;
	uint8_t *d_plane;
	int32_t *d_status;
	int32_t wblk, hblk, luma;
	int32_t band;  /* 0: a whole plane.  Bit 0 / bit 1: the plane is a band of block rows whose top /
	                * bottom apron row is a halo row received from the neighbouring band (pass A leaves it alone) */
} qs_hip_plane_ref;   /* 48 bytes, unchanged since the plane-set calls appeared: every field must be set by the caller */
int qs_hip_idct_planes(const qs_hip_plane_ref *refs, int n, int first, void *stream);
int qs_hip_smooth_planes(const qs_hip_plane_ref *refs, int n, int flags, int final_clamp, void *stream);
/* qs_hip_smooth_planes that ALSO writes the next iteration's pixel planes (see qs_hip_smooth_plane_next):
 * d_plane_next[i] is the second plane of refs[i] -- same geometry, a different buffer -- or NULL for a plane that gets
 * none; d_plane_next == NULL is qs_hip_smooth_planes.  The second planes travel in a PARALLEL array on purpose: the
 * struct above keeps its size and stride, so callers compiled against an earlier header (who neither zero nor know a
 * trailing field) stay correct. */
int qs_hip_smooth_planes_next(const qs_hip_plane_ref *refs, uint8_t *const *d_plane_next, int n, int flags,
		int final_clamp, void *stream);

/* JOINT_YUV chroma predictor + fdct_clamp for one chroma plane (reference :577-579,
 * 893-921, 343-347, 551-561); d_luma_lowres = luma at this plane's resolution and
 * geometry.  rebalance/final_clamp: run them here (LOW_QUALITY chroma, which skips
 * the recovery loop) instead of in qs_hip_smooth_plane. */
int qs_hip_joint_plane(const void *d_consts, int16_t *d_coef, const uint8_t *d_plane,
		const uint8_t *d_luma_lowres, int wblk, int hblk,
		int rebalance, int final_clamp, void *stream);
/* LOW_QUALITY pass B: range filter + fdct_clamp + rebalance (reference :924-938, 1161-1178) */
int qs_hip_lowq_plane(const void *d_consts, int16_t *d_coef, const uint8_t *d_plane,
		int wblk, int hblk, int rebalance, int final_clamp, void *stream);
/* box-downsample the luma plane by ws x hs into a plane of lwblk x lhblk blocks,
 * replicated out to its edges and apron (reference :2753-2815) */
int qs_hip_downsample_plane(const uint8_t *d_luma, int ywblk, int yhblk, uint8_t *d_lowres,
		int lwblk, int lhblk, int ws, int hs, void *stream);
/* UPSAMPLE_UV: chroma plane -> full-resolution pixel buffer (pitch/size from the two
 * helpers), guided by luma (reference :1851-2393, 2714-2730); then qs_hip_fdct_plane
 * re-encodes it into ywblk x yhblk blocks (reference :2735-2750) */
size_t qs_hip_upsample_pitch(int image_width, int ws);
size_t qs_hip_upsample_bytes(int image_width, int image_height, int ws, int hs);
int qs_hip_upsample_plane(const uint8_t *d_chroma, const uint8_t *d_luma_lowres, int cwblk,
		const uint8_t *d_luma, int ywblk, int yhblk, uint8_t *d_pixels,
		int image_width, int image_height, int ws, int hs, void *stream);
/* the same with explicit geometry, for one band of a sharded image: w1 = chroma
 * width in pixels (image level), h1 = valid low-res rows in this band, first_rows =
 * how many of its leading rows belong to the image's first 8-row strip (they get
 * the reference's right-edge replicate, :2390-2393), pitch = row pitch of d_pixels */
int qs_hip_upsample_rows(const uint8_t *d_chroma, const uint8_t *d_luma_lowres, int cwblk,
		const uint8_t *d_luma, int ywblk, int yhblk, uint8_t *d_pixels, size_t pitch,
		int w1, int h1, int first_rows, int ws, int hs, void *stream);
int qs_hip_fdct_plane(const uint8_t *d_pixels, size_t pitch, int16_t *d_coef, int wblk, int hblk, void *stream);

/* final +-1023 clamp alone (reference :2668-2689) */
int qs_hip_clamp_plane(int16_t *d_coef, int wblk, int hblk, void *stream);
/* dequantise only (reference :2551-2566) */
int qs_hip_dequant_plane(const void *d_consts, int16_t *d_coef, int wblk, int hblk, void *stream);

#ifdef __cplusplus
}
#endif
#endif
