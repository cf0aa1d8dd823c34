/*
 * jpegqs_shim.c -- libjpeg-facing drop-in for the reference's library API
 * (include/libjpegqs.h  <->  reference libjpegqs.h:47-56), host side in C.
 *
 * do_quantsmooth() here does what a maintainer of the reference would keep on
 * the host: talk to libjpeg (virtual block arrays, quantisation tables,
 * component geometry, logging, decode-mode re-initialisation) and hand the
 * coefficient-recovery path itself to the GPU library through the flat C ABI
 * (include/jpegqs_hip.h).  Behaviour mirrored from reference
 * quantsmooth.h:2404-2905; citations below.
 *
 * Back ends (the reference's counterpart is its run-time dispatcher, libjpegqs.c:80-156, which always
 * ends in a function that smooths the image): the GPU library whenever a HIP device is visible; the CPU
 * back end of csrc/qs_cpu.c -- same job contract, same bit-exact results -- when NONE is (a box without a
 * GPU still gets a smoothed file), announced on stderr.  JPEGQS_BACKEND=hip forbids the CPU route (the
 * call then fails as before), JPEGQS_BACKEND=cpu (or QS_HIP_FORCE_CPU=1) forces it.  A GPU that is
 * present but FAILS (out of memory, launch error) is never papered over with the CPU: the call reports
 * the failure, counts a libjpeg warning and leaves the image untouched.
 *
 * Build: cc -shared -fPIC jpegqs_shim.c qs_cpu.c -fopenmp -I<libjpeg include> -ldl -lpthread   (libjpegqs_hip.so is
 *        loaded at first use, see hip_lib() below)
 *        (no libjpeg symbols are needed unless TRANSCODE_ONLY is left undefined,
 *        in which case the four jinit_* entry points below come from libjpeg).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE   /* dladdr */
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>

#include "jpeglib.h"
#include "../../include/libjpegqs.h"
#include "../../include/jpegqs_hip.h"
#include "qs_cpu.h"

#define logfmt(...) fprintf(stderr, __VA_ARGS__)

/* ---- the GPU library is loaded at first use, not linked ------------------------------------------------------------------
 * libjpegqs_hip.so needs the HIP runtime (libamdhip64); a program linked against THIS library must still start -- and
 * get its images smoothed by the CPU back end -- on a machine where that runtime is not installed at all.  So the six
 * entry points used here are looked up with dlopen / dlsym: first the file named by QS_HIP_LIB (A/B builds), then
 * libjpegqs_hip.so next to this library (found through dladdr), then by name through the loader's search path. */
#include <dlfcn.h>
#include <pthread.h>
static struct {
	int ok;
	char why[256];
	int (*do_rows)(qs_hip_job *, int16_t *const *const *, int, int, int, qs_hip_progress_fn, void *);
	int (*do_flat)(qs_hip_job *, int, int, int, qs_hip_progress_fn, void *);
	int (*prewarm)(const qs_hip_job *, int, int);
	int (*device_count)(void);
	const char *(*last_error)(void);
	void (*free_)(void *);
} H;
static pthread_once_t hip_once = PTHREAD_ONCE_INIT;
static void hip_load_once(void) {
	void *h = NULL;
	const char *over = getenv("QS_HIP_LIB");
	Dl_info me;
	char path[4096];
	if (over && *over) h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
	if (!h && dladdr((void *)&hip_load_once, &me) && me.dli_fname) {
		const char *slash = strrchr(me.dli_fname, '/');
		size_t n = slash ? (size_t)(slash - me.dli_fname) + 1 : 0;
		if (n + sizeof("libjpegqs_hip.so") < sizeof(path)) {
			memcpy(path, me.dli_fname, n);
			strcpy(path + n, "libjpegqs_hip.so");
			h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
		}
	}
	if (!h) h = dlopen("libjpegqs_hip.so", RTLD_NOW | RTLD_LOCAL);
	if (!h) {
		const char *e = dlerror();
		snprintf(H.why, sizeof(H.why), "the GPU library could not be loaded (%s)", e ? e : "dlopen failed");
		return;
	}
	*(void **)&H.do_rows = dlsym(h, "qs_hip_do_quantsmooth_rows");
	*(void **)&H.do_flat = dlsym(h, "qs_hip_do_quantsmooth");
	*(void **)&H.prewarm = dlsym(h, "qs_hip_prewarm");
	*(void **)&H.device_count = dlsym(h, "qs_hip_device_count");
	*(void **)&H.last_error = dlsym(h, "qs_hip_last_error");
	*(void **)&H.free_ = dlsym(h, "qs_hip_free");
	H.ok = H.do_rows && H.do_flat && H.prewarm && H.device_count && H.last_error && H.free_;
	if (!H.ok) snprintf(H.why, sizeof(H.why), "the GPU library lacks an entry point of include/jpegqs_hip.h");
}
static int hip_lib(void) { pthread_once(&hip_once, hip_load_once); return H.ok; }
static int hip_device_count(void) { return hip_lib() ? H.device_count() : 0; }

#ifndef TRANSCODE_ONLY
/* libjpeg-internal entry points used to re-arm the decompressor after the
 * coefficients / sampling factors changed (reference quantsmooth.h:33-61) */
#define QS_DSTATE_SCANNING 205
#define QS_DSTATE_RAW_OK 206
EXTERN(void) jinit_d_main_controller JPP((j_decompress_ptr, boolean));
EXTERN(void) jinit_inverse_dct JPP((j_decompress_ptr));
EXTERN(void) jinit_upsampler JPP((j_decompress_ptr));
EXTERN(void) jinit_color_deconverter JPP((j_decompress_ptr));
#include "qs_turbo_master.h"   /* libjpeg-turbo only: the private master record patched after UPSAMPLE_UV */
#endif

/* Why the last do_quantsmooth() of this thread returned non-zero.  The reference's return value
 * only knows "stop" (cancelled / rejected input, the image stays decodable); a GPU back end can
 * also FAIL (no device, out of memory, launch error), in which case nothing was processed.  The
 * reference API has no room for that distinction, so it is an extra query (not in the
 * reference): 0 = the call did what the reference would have done, otherwise a negative
 * QS_HIP_E* code.  The jpegqs CLI uses it to exit non-zero instead of writing an unprocessed file. */
static __thread int qs_backend_status = 0;
int jpegqs_hip_backend_status(void) { return qs_backend_status; }

/* Which back end the calling thread's last do_quantsmooth() ran on: "hip", "cpu", or "none" (early-out /
 * nothing ran).  Not in the reference. */
static __thread const char *qs_backend_name = "none";
const char *jpegqs_hip_backend_name(void) { return qs_backend_name; }

/* 0 = GPU, 1 = CPU back end (csrc/qs_cpu.c).  The CPU is chosen only when NO HIP device is visible, or on
 * request; *why receives the reason for the stderr notice. */
static int use_cpu_backend(const char **why) {
	const char *e = getenv("JPEGQS_BACKEND"), *f = getenv("QS_HIP_FORCE_CPU");
	if (e && !strcmp(e, "hip")) return 0;
	if (e && !strcmp(e, "cpu")) { *why = "JPEGQS_BACKEND=cpu"; return 1; }
	if (f && *f && strcmp(f, "0")) { *why = "QS_HIP_FORCE_CPU is set"; return 1; }
	if (!hip_lib()) { *why = H.why; return 1; }
	if (hip_device_count() <= 0) { *why = "no HIP device visible"; return 1; }
	return 0;
}

/* The only memory this file holds across libjpeg calls that is not in libjpeg's own pool: the
 * two replacement chroma arrays the GPU library malloc'ed (UPSAMPLE_UV), between its return and
 * their copy into new virtual arrays.  Should a libjpeg error exit (longjmp) fall into that
 * window, the pointers stay parked here and the next call on this thread releases them. */
static __thread void *qs_parked[2] = { NULL, NULL };
static __thread int qs_parked_cpu = 0;          /* they came from the CPU back end (plain malloc) */
/* ... and the component copies of the backing-store path (malloc'ed here) */
static __thread int16_t *qs_copy[QS_HIP_MAXC] = { NULL, NULL, NULL, NULL };
static void release_parked(void) {
	int j;
	for (j = 0; j < 2; j++) if (qs_parked[j]) {
		if (qs_parked_cpu) qs_cpu_free(qs_parked[j]); else if (hip_lib()) H.free_(qs_parked[j]);
		qs_parked[j] = NULL;
	}
	for (j = 0; j < QS_HIP_MAXC; j++) if (qs_copy[j]) { free(qs_copy[j]); qs_copy[j] = NULL; }
}

/* Not in the reference: start the GPU side in the background while libjpeg is still busy with the file
 * (qs_hip_prewarm, include/jpegqs_hip.h).  cinfo == NULL: the runtime only (call it first thing in main());
 * after jpeg_read_header(): also the transfer buffers for this image.  Returns at once; never fails loudly. */
void jpegqs_hip_prewarm(j_decompress_ptr cinfo, jpegqs_control_t *opts) {
	qs_hip_job g;
	int ci, i;
	if (!hip_lib()) return;
	if (!cinfo || !opts) { (void)H.prewarm(NULL, 0, 0); return; }
	if (opts->niter <= 0 && !(opts->flags & JPEGQS_UPSAMPLE_UV)) return;   /* the early-out of reference :2458 needs no device */
	if (cinfo->num_components < 1 || cinfo->num_components > QS_HIP_MAXC) return;
	memset(&g, 0, sizeof(g));
	g.ncomp = cinfo->num_components;
	g.colorspace = (int)cinfo->jpeg_color_space;
	g.image_width = (int)cinfo->image_width;
	g.image_height = (int)cinfo->image_height;
	for (ci = 0; ci < g.ncomp; ci++) {
		jpeg_component_info *comp = cinfo->comp_info + ci;
		JQUANT_TBL *tbl = comp->quant_table;
		if (!tbl && comp->quant_tbl_no >= 0 && comp->quant_tbl_no < NUM_QUANT_TBLS) tbl = cinfo->quant_tbl_ptrs[comp->quant_tbl_no];
		g.wblk[ci] = (int)comp->width_in_blocks; g.hblk[ci] = (int)comp->height_in_blocks;
		g.hsamp[ci] = comp->h_samp_factor; g.vsamp[ci] = comp->v_samp_factor;
		g.has_quant[ci] = tbl != NULL;
		if (tbl) for (i = 0; i < DCTSIZE2; i++) g.quant[ci][i] = tbl->quantval[i];
	}
	(void)H.prewarm(&g, opts->flags & JPEGQS_FLAGS_MASK, opts->niter);
}

static double now_ms(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static int cmp_ptr(const void *a, const void *b) {
	const char *x = *(char *const *)a, *y = *(char *const *)b;
	return x < y ? -1 : x > y;
}

/* all row pointers distinct and non-overlapping <=> every row has its own memory, i.e. the
 * virtual array is memory-resident and the pointers stay valid across access calls */
static int rows_are_resident(j_decompress_ptr srcinfo, int16_t **rows, JDIMENSION n, size_t rowbytes) {
	JDIMENSION y;
	char **tmp = (char**)(*srcinfo->mem->alloc_small)((j_common_ptr)srcinfo, JPOOL_IMAGE, (size_t)n * sizeof(char*));
	memcpy(tmp, rows, (size_t)n * sizeof(char*));
	qsort(tmp, n, sizeof(char*), cmp_ptr);
	for (y = 1; y < n; y++)
		if ((size_t)(tmp[y] - tmp[y - 1]) < rowbytes) return 0;
	return 1;
}

int do_quantsmooth(j_decompress_ptr srcinfo, jvirt_barray_ptr *coef_arrays, jpegqs_control_t *opts) {
	qs_hip_job job;
	int16_t **rows[QS_HIP_MAXC] = { NULL, NULL, NULL, NULL };
	int ci, i, ret, flags = opts->flags;
	int upsampled = 0, in_place = 1, on_cpu = 0, have_work;
	const char *why_cpu = "";
	double t0 = 0;
	jpeg_component_info *comp;
	JDIMENSION blk_y;

	/* --info output, reference quantsmooth.h:2422-2445 */
	if (flags & JPEGQS_INFO_COMP1)
		for (ci = 0; ci < srcinfo->num_components; ci++) {
			comp = srcinfo->comp_info + ci;
			logfmt("component[%i] : table %i, samp %ix%i\n", ci, comp->quant_tbl_no,
					comp->h_samp_factor, comp->v_samp_factor);
		}
	if (flags & JPEGQS_INFO_QUANT)
		for (i = 0; i < NUM_QUANT_TBLS; i++) {
			int x, y;
			JQUANT_TBL *qtbl = srcinfo->quant_tbl_ptrs[i];
			if (!qtbl) continue;
			logfmt("quant[%i]:\n", i);
			for (y = 0; y < DCTSIZE; y++) {
				for (x = 0; x < DCTSIZE; x++) logfmt("%04x ", qtbl->quantval[y * DCTSIZE + x]);
				logfmt("\n");
			}
		}
	/* the reference's early-out (quantsmooth.h:2458) needs no back end at all */
	have_work = opts->niter > 0 || (flags & JPEGQS_UPSAMPLE_UV);
	if (have_work) on_cpu = use_cpu_backend(&why_cpu);
	if (flags & JPEGQS_INFO_CPU) {                  /* the reference's dispatcher prints its choice here (libjpegqs.c:141-145) */
		if (on_cpu) logfmt("SIMD type: cpu back end (%s; %d blocks per vector, %s)\n", why_cpu, qs_cpu_lanes(), qs_cpu_isa());
		else logfmt("SIMD type: hip/gfx950 (%d device(s))\n", hip_device_count());
	}
	if (on_cpu) {
		/* never silent: once per process, whatever the --info bits say */
		static int announced = 0;
		if (!__sync_lock_test_and_set(&announced, 1))
			logfmt("jpegqs: %s -- using the CPU back end (%d blocks per vector, %s)\n", why_cpu, qs_cpu_lanes(), qs_cpu_isa());
	}
	if (flags & JPEGQS_INFO_TIME) t0 = now_ms();

	qs_backend_status = 0;
	qs_backend_name = "none";
	release_parked();
	if (srcinfo->num_components < 1 || srcinfo->num_components > QS_HIP_MAXC) {
		logfmt("jpegqs-hip: unsupported component count %d\n", srcinfo->num_components);
		qs_backend_status = QS_HIP_EINVAL;
		return 1;
	}

	/* ---- the job's view of the coefficients: one pointer per JBLOCKROW, straight into libjpeg's
	 * virtual arrays.  access_virt_barray hands out one row at a time (allocated row width may
	 * exceed width_in_blocks; only the first width_in_blocks blocks count, reference
	 * quantsmooth.h:2557-2560, 2591-2594).  For memory-resident arrays (the normal case) the row
	 * pointers stay valid and are all distinct, and the GPU library gathers from / scatters to them
	 * with its helper threads -- no intermediate copy on this side.  Arrays that libjpeg swaps
	 * through a backing store re-use one window: detected by coinciding pointers, then the rows
	 * are copied (second path below).  All scratch memory here comes from libjpeg's JPOOL_IMAGE
	 * pool, so an error exit (longjmp) out of a libjpeg call cannot leak it. */
	memset(&job, 0, sizeof(job));
	job.ncomp = srcinfo->num_components;
	job.colorspace = (int)srcinfo->jpeg_color_space;
	job.image_width = (int)srcinfo->image_width;
	job.image_height = (int)srcinfo->image_height;
	for (ci = 0; ci < job.ncomp; ci++) {
		comp = srcinfo->comp_info + ci;
		job.wblk[ci] = (int)comp->width_in_blocks;
		job.hblk[ci] = (int)comp->height_in_blocks;
		job.hsamp[ci] = comp->h_samp_factor;
		job.vsamp[ci] = comp->v_samp_factor;
		job.has_quant[ci] = comp->quant_table != NULL;
		if (comp->quant_table)
			for (i = 0; i < DCTSIZE2; i++) job.quant[ci][i] = comp->quant_table->quantval[i];
		if (comp->width_in_blocks == 0 || comp->height_in_blocks == 0) {
			logfmt("jpegqs-hip: empty component %d\n", ci);
			qs_backend_status = QS_HIP_EINVAL;
			return 1;
		}
		rows[ci] = (int16_t**)(*srcinfo->mem->alloc_small)((j_common_ptr)srcinfo, JPOOL_IMAGE,
				(size_t)comp->height_in_blocks * sizeof(int16_t*));
		for (blk_y = 0; blk_y < comp->height_in_blocks; blk_y++) {
			JBLOCKARRAY buf = (*srcinfo->mem->access_virt_barray)
					((j_common_ptr)srcinfo, coef_arrays[ci], blk_y, 1, FALSE);   /* (read access: a swapped array is not marked dirty) */
			rows[ci][blk_y] = (int16_t*)buf[0];
		}
		if (!rows_are_resident(srcinfo, rows[ci], comp->height_in_blocks,
				(size_t)comp->width_in_blocks * sizeof(JBLOCK))) in_place = 0;
	}
	if (!in_place)
		for (ci = 0; ci < job.ncomp; ci++) {              /* copy path: rows are only valid one at a time */
			size_t rowbytes;
			comp = srcinfo->comp_info + ci;
			rowbytes = (size_t)comp->width_in_blocks * sizeof(JBLOCK);
			/* plain malloc, not libjpeg's pool: one alloc_large request is limited to MAX_ALLOC_CHUNK
			 * (about 1e9 bytes) and would stay allocated until the jpeg object is destroyed; the copy is
			 * parked per thread so that a libjpeg error exit (longjmp) cannot leak it */
			job.coef[ci] = qs_copy[ci] = (int16_t*)malloc(rowbytes * comp->height_in_blocks);
			if (!job.coef[ci]) {
				logfmt("jpegqs-hip: out of host memory\n");
				qs_backend_status = QS_HIP_ENOMEM;
				release_parked();
				return 1;
			}
			for (blk_y = 0; blk_y < comp->height_in_blocks; blk_y++) {
				JBLOCKARRAY buf = (*srcinfo->mem->access_virt_barray)
						((j_common_ptr)srcinfo, coef_arrays[ci], blk_y, 1, FALSE);
				memcpy((char*)job.coef[ci] + rowbytes * blk_y, buf[0], rowbytes);
			}
		}

	/* reference :2569-2572 prints this when it starts on a component's plane:
	 * every component with a table whose values are not all <= 1, unless the
	 * iteration count leaves it nothing to do */
	if (flags & JPEGQS_INFO_COMP2)
		for (ci = 0; ci < job.ncomp; ci++) {
			int acc = 0;
			if (!job.has_quant[ci]) continue;
			for (i = 0; i < DCTSIZE2; i++) acc |= job.quant[ci][i];
			if (acc >= 0x800) break;                     /* rejected table: the reference stops here */
			if (acc <= 1 || opts->niter <= 0) continue;
			logfmt("component[%i] : size %ix%i\n", ci, job.wblk[ci], job.hblk[ci]);
		}

	if (on_cpu) {
		ret = qs_cpu_do_quantsmooth(&job, in_place ? (int16_t *const *const *)rows : NULL, flags & JPEGQS_FLAGS_MASK,
				opts->niter, opts->threads, opts->progprec, opts->progress, opts->userdata);
		if (ret < 0) logfmt("jpegqs: the CPU back end rejected the image (code %d)\n", ret);
		else qs_backend_name = "cpu";
	} else {
		if (!hip_lib()) {                            /* (JPEGQS_BACKEND=hip on a machine without the GPU library) */
			ret = QS_HIP_ENODEV;
			logfmt("jpegqs-hip: no HIP device: %s\n", H.why);
		} else {
			if (in_place)
				ret = H.do_rows(&job, (int16_t *const *const *)rows, flags & JPEGQS_FLAGS_MASK,
						opts->niter, opts->progprec, opts->progress, opts->userdata);
			else
				ret = H.do_flat(&job, flags & JPEGQS_FLAGS_MASK, opts->niter, opts->progprec,
						opts->progress, opts->userdata);
			if (ret < 0) logfmt("jpegqs-hip: %s\n", H.last_error());
		}
		if (ret >= 0 && have_work) qs_backend_name = "hip";
	}
	if (ret < 0) {
		/* A back end that FAILED left the image untouched; callers written against the reference only see
		 * "non-zero" (cancelled / rejected: still a smoothed-so-far, decodable image) and its own CLI ignores even
		 * that (reference quantsmooth.c:550).  So the failure is also counted as a libjpeg warning -- what every
		 * libjpeg application checks after a damaged file, and what makes the reference's CLI end with its
		 * documented exit code 2 (quantsmooth.c:626) instead of 0. */
		srcinfo->err->num_warnings++;
		qs_backend_status = ret;
		release_parked();
		return 1;
	}

	/* The job layer rewrites every table it was given to all-ones unless it took
	 * the reference's early-out (niter <= 0 and nothing to upsample, reference
	 * :2458), which touches nothing -- then neither do we. */
	{
		int changed = job.up_wblk > 0;
		for (ci = 0; ci < job.ncomp && !changed; ci++)
			if (job.has_quant[ci])
				for (i = 0; i < DCTSIZE2; i++)
					if (job.quant[ci][i] != srcinfo->comp_info[ci].quant_table->quantval[i]) { changed = 1; break; }
		if (!changed) { release_parked(); return ret; }
	}

	/* ---- copy path only: scatter the processed blocks back (in place there is nothing to do) */
	if (!in_place)
		for (ci = 0; ci < job.ncomp; ci++) {
			size_t rowbytes;
			comp = srcinfo->comp_info + ci;
			rowbytes = (size_t)comp->width_in_blocks * sizeof(JBLOCK);
			for (blk_y = 0; blk_y < comp->height_in_blocks; blk_y++) {
				JBLOCKARRAY buf = (*srcinfo->mem->access_virt_barray)
						((j_common_ptr)srcinfo, coef_arrays[ci], blk_y, 1, TRUE);
				memcpy(buf[0], (char*)job.coef[ci] + rowbytes * blk_y, rowbytes);
			}
			free(qs_copy[ci]); qs_copy[ci] = NULL;
		}

	/* ---- UPSAMPLE_UV replaced the chroma arrays: new virtual arrays at luma
	 * size, sampling factors 1x1 (reference :2696-2703, 2836-2849) */
	if (job.up_wblk > 0 && job.ncomp >= 3) {
		JDIMENSION uw = (JDIMENSION)job.up_wblk, uh = (JDIMENSION)job.up_hblk;
		size_t rowbytes = (size_t)uw * sizeof(JBLOCK);
		jvirt_barray_ptr up[2];
		qs_parked[0] = job.coef_up[0]; qs_parked[1] = job.coef_up[1]; qs_parked_cpu = on_cpu;
		for (ci = 0; ci < 2; ci++)
			up[ci] = (*srcinfo->mem->request_virt_barray)
					((j_common_ptr)srcinfo, JPOOL_IMAGE, FALSE, uw, uh, 1);
		(*srcinfo->mem->realize_virt_arrays)((j_common_ptr)srcinfo);
		for (ci = 0; ci < 2; ci++) {
			for (blk_y = 0; blk_y < uh; blk_y++) {
				JBLOCKARRAY buf = (*srcinfo->mem->access_virt_barray)
						((j_common_ptr)srcinfo, up[ci], blk_y, 1, TRUE);
				memcpy(buf[0], (char*)job.coef_up[ci] + rowbytes * blk_y, rowbytes);
			}
			if (on_cpu) qs_cpu_free(job.coef_up[ci]); else H.free_(job.coef_up[ci]);
			qs_parked[ci] = NULL;
			coef_arrays[ci + 1] = up[ci];
			srcinfo->comp_info[ci + 1].width_in_blocks = uw;
			srcinfo->comp_info[ci + 1].height_in_blocks = uh;
		}
		srcinfo->max_h_samp_factor = 1;
		srcinfo->max_v_samp_factor = 1;
		srcinfo->comp_info[0].h_samp_factor = 1;
		srcinfo->comp_info[0].v_samp_factor = 1;
		upsampled = 1;
	}

	if (!ret && (flags & JPEGQS_INFO_TIME))   /* reference :2820-2825 */
		logfmt("quantsmooth: %.3fms\n", now_ms() - t0);

	/* ---- every quantisation table becomes all-ones (reference :2851-2859) */
	for (ci = 0; ci < NUM_QUANT_TBLS; ci++) {
		JQUANT_TBL *qtbl = srcinfo->quant_tbl_ptrs[ci];
		if (qtbl) for (i = 0; i < DCTSIZE2; i++) qtbl->quantval[i] = 1;
	}
	for (ci = 0; ci < srcinfo->num_components; ci++) {
		JQUANT_TBL *qtbl = srcinfo->comp_info[ci].quant_table;
		if (qtbl) for (i = 0; i < DCTSIZE2; i++) qtbl->quantval[i] = 1;
	}

#ifndef TRANSCODE_ONLY
	/* decode mode: re-arm what depends on the tables / sampling factors so that
	 * jpeg_read_scanlines() works (reference :2861-2876) */
	if (!(flags & JPEGQS_TRANSCODE)) {
		if (upsampled) {
#ifdef LIBJPEG_TURBO_VERSION
			/* libjpeg-turbo keeps per-component MCU column limits in its private master
			 * record; the chroma components now have luma's geometry (reference :2864-2867) */
			struct qs_turbo_master *master = (struct qs_turbo_master *)srcinfo->master;
			master->last_MCU_col[1] = master->last_MCU_col[0];
			master->last_MCU_col[2] = master->last_MCU_col[0];
#endif
			jinit_color_deconverter(srcinfo);
			jinit_upsampler(srcinfo);
			jinit_d_main_controller(srcinfo, FALSE);
			srcinfo->input_iMCU_row = (srcinfo->output_height + DCTSIZE - 1) / DCTSIZE;
		}
		jinit_inverse_dct(srcinfo);
	}
#else
	(void)upsampled;
#endif
	return ret;
}

#ifndef TRANSCODE_ONLY
/* reference quantsmooth.h:2880-2895 */
boolean jpegqs_start_decompress(j_decompress_ptr cinfo, jpegqs_control_t *opts) {
	boolean ret;
	int use_qs = opts->niter > 0 || (opts->flags & JPEGQS_UPSAMPLE_UV);
	if (use_qs) cinfo->buffered_image = TRUE;
	ret = jpeg_start_decompress(cinfo);
	if (use_qs) {
		/* the GPU side comes up while libjpeg reads the scans below (buffered-image mode: jpeg_start_decompress() has
		 * only set the decompressor up -- and has had its chance to refuse the file, e.g. an unsupported colour
		 * conversion, before a device is touched for it) */
		jpegqs_hip_prewarm(cinfo, opts);
		while (!jpeg_input_complete(cinfo)) {
			jpeg_start_output(cinfo, cinfo->input_scan_number);
			jpeg_finish_output(cinfo);
		}
		do_quantsmooth(cinfo, jpeg_read_coefficients(cinfo), opts);
		/* The reference ignores the return value here (cancelled / rejected: the image decodes as it
		 * is).  A back end can also FAIL; the image then decodes unsmoothed, and that does not pass
		 * silently: do_quantsmooth() has counted it as a libjpeg warning (cinfo->err->num_warnings) and
		 * jpegqs_hip_backend_status() keeps the code for callers that know about it. */
		jpeg_start_output(cinfo, cinfo->input_scan_number);
	}
	return ret;
}

/* reference quantsmooth.h:2897-2904 */
boolean jpegqs_finish_decompress(j_decompress_ptr cinfo) {
	if ((cinfo->global_state == QS_DSTATE_SCANNING || cinfo->global_state == QS_DSTATE_RAW_OK)
			&& cinfo->buffered_image)
		jpeg_finish_output(cinfo);
	return jpeg_finish_decompress(cinfo);
}
#endif
