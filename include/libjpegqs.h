/*
 * JPEG Quant Smooth API definitions
 * Copyright (C) 2020-2026 Ilya Kurdyukov  (the API below -- names, enum values,
 * struct layout, prototypes -- is that of the reference's libjpegqs.h, LGPL-2.1;
 * the attribution stays with it)
 *
 * libjpegqs.h -- the JPEG Quant Smooth library API, as implemented by the
 * MI355X (gfx950) build in this repository.
 *
 * This is the drop-in surface: the names, enum values, struct layout and
 * function signatures are those of the reference's public header
 * (reference libjpegqs.h:14-56) so that a program written against the
 * reference -- its `jpegqs` CLI (reference quantsmooth.c:550), example.c:96,
 * the IrfanView plugin -- links against libjpegqs_hip_shim.so unchanged.
 * Include <jpeglib.h> (any libjpeg 6b..9, or libjpeg-turbo) before this file.
 *
 * Implementation: csrc/jpegqs_shim.c hands the JBLOCKROWs of each component to
 * the GPU library (include/jpegqs_hip.h), which runs the whole
 * coefficient-recovery path on the MI355X.  On a machine with NO visible HIP
 * device the same job runs on the library's CPU back end (csrc/qs_cpu.c, same
 * bit-exact results; announced on stderr) -- like the reference, the call then
 * still delivers a smoothed image.  JPEGQS_BACKEND=hip in the environment
 * forbids that route, JPEGQS_BACKEND=cpu forces it.  A GPU that is present but
 * fails (out of memory, launch error) is not papered over: do_quantsmooth()
 * reports the error on stderr, counts a libjpeg warning
 * (srcinfo->err->num_warnings), leaves the coefficients and quantisation tables
 * untouched (the file stays a valid JPEG) and returns non-zero.
 */
#ifndef JPEGQS_H
#define JPEGQS_H

#ifdef __cplusplus
extern "C" {
#endif

/* jpegqs_control_t.flags: algorithm bits 0-6, CPU cap bits 12-15 (accepted and
 * ignored here: the reference's SIMD tiers do not exist), info/log bits 16-20. */
enum {
	JPEGQS_ITER_MAX        = 100,  /* niter is clamped to [0, 100] */
	JPEGQS_DIAGONALS       = 1,    /* --quality >= 4: diagonal neighbour terms */
	JPEGQS_JOINT_YUV       = 2,    /* --quality >= 5: chroma predicted from luma */
	JPEGQS_UPSAMPLE_UV     = 4,    /* --quality >= 6: chroma upsampled to luma size */
	JPEGQS_LOW_QUALITY     = 8,    /* --quality 0..2: fast range filter */
	JPEGQS_NO_REBALANCE    = 16,
	JPEGQS_NO_REBALANCE_UV = 32,
	JPEGQS_TRANSCODE       = 64,   /* caller writes coefficients (no decode re-init) */
	JPEGQS_FLAGS_MASK      = 0x7f,
	JPEGQS_CPU_SHIFT       = 12,
	JPEGQS_CPU_MASK        = 15,
	JPEGQS_INFO_SHIFT      = 16,
	JPEGQS_INFO_COMP1      = 1 << JPEGQS_INFO_SHIFT,   /* component table / sampling */
	JPEGQS_INFO_QUANT      = 2 << JPEGQS_INFO_SHIFT,   /* dump quantisation tables */
	JPEGQS_INFO_COMP2      = 4 << JPEGQS_INFO_SHIFT,   /* component sizes in blocks */
	JPEGQS_INFO_TIME       = 8 << JPEGQS_INFO_SHIFT,   /* "quantsmooth: X ms" */
	JPEGQS_INFO_CPU        = 16 << JPEGQS_INFO_SHIFT   /* which back end runs */
};

#ifndef JPEGQS_ATTR
#define JPEGQS_ATTR
#endif

#define JPEGQS_VERSION "1.20230818-hip"
#define JPEGQS_COPYRIGHT "MI355X implementation of JPEG Quant Smooth (algorithm (C) 2016-2026 Ilya Kurdyukov)"

typedef struct {
	int flags;      /* JPEGQS_* bits */
	int niter;      /* iterations, default 3 in the CLI */
	int threads;    /* CPU back end only (as in the reference); the GPU path ignores it */
	int progprec;   /* progress granularity: 0 -> 20 steps, < 0 -> finest */
	void *userdata; /* passed back to progress() */
	int (*progress)(void *data, int cur, int max); /* non-zero return cancels */
} jpegqs_control_t;

/* Smooth the coefficient arrays of `srcinfo` in place (after
 * jpeg_read_coefficients()).  Returns 0 when complete, non-zero when
 * cancelled, rejected (damaged tables / coefficients) or when the GPU failed. */
JPEGQS_ATTR
int do_quantsmooth(j_decompress_ptr srcinfo, jvirt_barray_ptr *coef_arrays, jpegqs_control_t *opts);

/* Not in the reference: why the calling thread's last do_quantsmooth() returned non-zero.
 * 0 = for the reference's own reasons (cancelled by progress(), rejected tables/coefficients: the
 * image is still decodable); < 0 = the back end failed (out of memory, launch error, no HIP device
 * while JPEGQS_BACKEND=hip -- a QS_HIP_E* code of jpegqs_hip.h) and the image was left untouched. */
JPEGQS_ATTR
int jpegqs_hip_backend_status(void);
/* Not in the reference: which back end the calling thread's last do_quantsmooth() ran on --
 * "hip", "cpu" (no HIP device was visible, or JPEGQS_BACKEND=cpu), "none" (nothing to do / failed). */
JPEGQS_ATTR
const char *jpegqs_hip_backend_name(void);

/* Not in the reference: bring the GPU side up in the background while libjpeg is still decoding the file.
 * jpegqs_hip_prewarm(NULL, NULL) first thing in main() starts the HIP runtime; called again after
 * jpeg_read_header() with the options of the coming do_quantsmooth() it also sizes the transfer buffers.
 * Optional -- do_quantsmooth() works without it, the first call in a process is just slower by 0.1-0.3 s.
 * jpegqs_start_decompress() calls it by itself. */
JPEGQS_ATTR
void jpegqs_hip_prewarm(j_decompress_ptr cinfo, jpegqs_control_t *opts);

#ifndef TRANSCODE_ONLY
/* Decode-mode wrappers: replace jpeg_start_decompress()/jpeg_finish_decompress()
 * so that jpeg_read_scanlines() delivers the smoothed image. */
JPEGQS_ATTR
boolean jpegqs_start_decompress(j_decompress_ptr cinfo, jpegqs_control_t *opts);

JPEGQS_ATTR
boolean jpegqs_finish_decompress(j_decompress_ptr cinfo);
#endif

#ifdef __cplusplus
}
#endif
#endif
