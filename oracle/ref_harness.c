/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Wraps the UNMODIFIED reference sources (compiled from where they lie under
 * /root/reference, see oracle/Makefile) behind a flat, libjpeg-free ABI so
 * that Python tests and bench.py's cpu_baseline leg can drive the reference's
 * own do_quantsmooth()/quantsmooth_block()/idct_islow()/... on plain arrays.
 *
 * How: the reference's plane driver only talks to libjpeg through
 *   srcinfo->mem->{access_virt_barray,request_virt_barray,realize_virt_arrays}
 * and a handful of jpeg_decompress_struct / jpeg_component_info fields
 * (reference quantsmooth.h:2423-2427, 2447-2451, 2488-2493, 2557-2558,
 * 2696-2703).  We hand it a jpeg_decompress_struct whose memory manager is a
 * small fake that serves rows out of caller-owned flat int16 arrays, so no
 * libjpeg code is linked (TRANSCODE_ONLY removes the jinit_* calls).
 *
 * Nothing in this file is product code; no reference source is copied here --
 * the reference header is #included at compile time via -I/root/reference.
 * Output objects go to oracle/_ref/ only.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stddef.h>

#include "jpeglib.h"

#define TRANSCODE_ONLY
#define JPEGQS_ATTR static
#include "quantsmooth.h" /* the reference itself (-I/root/reference) */

#define QSREF_MAXC 4

/* ---- fake virtual block array ------------------------------------------ */
typedef struct {
	JBLOCKROW *rows; /* hblk row pointers */
	JBLOCK *data;    /* hblk * wblk blocks (owned iff owned != 0) */
	JDIMENSION wblk, hblk;
	int owned;
} fake_barray;

typedef struct {
	struct jpeg_memory_mgr pub;
	fake_barray *extra[8];
	int nextra;
} fake_mem;

static fake_barray *fake_barray_new(JBLOCK *data, JDIMENSION wblk, JDIMENSION hblk) {
	fake_barray *a = (fake_barray*)calloc(1, sizeof(*a));
	JDIMENSION y;
	if (!a) return NULL;
	a->wblk = wblk; a->hblk = hblk;
	if (!data) {
		data = (JBLOCK*)calloc((size_t)wblk * hblk + 1, sizeof(JBLOCK));
		a->owned = 1;
	}
	a->data = data;
	a->rows = (JBLOCKROW*)malloc(sizeof(JBLOCKROW) * (hblk + 1));
	for (y = 0; y < hblk; y++) a->rows[y] = data + (size_t)y * wblk;
	return a;
}

static void fake_barray_free(fake_barray *a, int keep_data) {
	if (!a) return;
	if (a->owned && !keep_data) free(a->data);
	free(a->rows); free(a);
}

static JBLOCKARRAY fake_access(j_common_ptr cinfo, jvirt_barray_ptr ptr,
		JDIMENSION start_row, JDIMENSION num_rows, boolean writable) {
	fake_barray *a = (fake_barray*)ptr;
	(void)cinfo; (void)num_rows; (void)writable;
	return a->rows + start_row;
}

static jvirt_barray_ptr fake_request(j_common_ptr cinfo, int pool_id, boolean pre_zero,
		JDIMENSION blocksperrow, JDIMENSION numrows, JDIMENSION maxaccess) {
	fake_mem *m = (fake_mem*)cinfo->mem;
	fake_barray *a = fake_barray_new(NULL, blocksperrow, numrows);
	(void)pool_id; (void)pre_zero; (void)maxaccess;
	if (m->nextra < 8) m->extra[m->nextra++] = a;
	return (jvirt_barray_ptr)a;
}

static void fake_realize(j_common_ptr cinfo) { (void)cinfo; }

/* ---- flat job description (mirrored by tests/qsref.py) ------------------ */
typedef struct {
	int32_t ncomp;            /* 1..4 */
	int32_t colorspace;       /* J_COLOR_SPACE value: 1 gray, 2 RGB, 3 YCbCr */
	int32_t image_width, image_height;
	int32_t wblk[QSREF_MAXC], hblk[QSREF_MAXC];
	int32_t hsamp[QSREF_MAXC], vsamp[QSREF_MAXC];
	int32_t has_quant[QSREF_MAXC];   /* 0 => comp_info[ci].quant_table == NULL */
	uint16_t quant[QSREF_MAXC][64];  /* in: table; out: what the reference left */
	int16_t *coef[QSREF_MAXC];       /* in/out, hblk*wblk*64 int16, natural order */
	/* outputs when UPSAMPLE_UV replaced the chroma arrays: */
	int16_t *coef_up[2];             /* malloc'd here, free with qsref_free() */
	int32_t up_wblk, up_hblk;        /* 0 when not replaced */
	int32_t out_hsamp0, out_vsamp0;  /* comp 0 sampling factors after the call */
} qsref_job;

typedef int (*qsref_progress_fn)(void *, int, int);

int qsref_do_quantsmooth(qsref_job *job, int flags, int niter, int threads,
		int progprec, qsref_progress_fn progress, void *userdata) {
	struct jpeg_decompress_struct cinfo;
	jpeg_component_info comp[QSREF_MAXC];
	JQUANT_TBL qt[QSREF_MAXC];
	fake_mem mem;
	fake_barray *arr[QSREF_MAXC];
	jvirt_barray_ptr coef_arrays[QSREF_MAXC];
	jpegqs_control_t opts;
	int ci, i, ret, maxh = 1, maxv = 1;

	memset(&cinfo, 0, sizeof(cinfo));
	memset(comp, 0, sizeof(comp));
	memset(&mem, 0, sizeof(mem));
	mem.pub.access_virt_barray = fake_access;
	mem.pub.request_virt_barray = fake_request;
	mem.pub.realize_virt_arrays = fake_realize;
	cinfo.mem = &mem.pub;
	cinfo.num_components = job->ncomp;
	cinfo.comp_info = comp;
	cinfo.jpeg_color_space = (J_COLOR_SPACE)job->colorspace;
	cinfo.image_width = job->image_width;
	cinfo.image_height = job->image_height;
	for (ci = 0; ci < job->ncomp; ci++) {
		comp[ci].component_index = ci;
		comp[ci].quant_tbl_no = ci;
		comp[ci].h_samp_factor = job->hsamp[ci];
		comp[ci].v_samp_factor = job->vsamp[ci];
		comp[ci].width_in_blocks = job->wblk[ci];
		comp[ci].height_in_blocks = job->hblk[ci];
		if (job->hsamp[ci] > maxh) maxh = job->hsamp[ci];
		if (job->vsamp[ci] > maxv) maxv = job->vsamp[ci];
		if (job->has_quant[ci]) {
			for (i = 0; i < 64; i++) qt[ci].quantval[i] = job->quant[ci][i];
			qt[ci].sent_table = FALSE;
			comp[ci].quant_table = &qt[ci];
			cinfo.quant_tbl_ptrs[ci] = &qt[ci];
		}
		arr[ci] = fake_barray_new((JBLOCK*)job->coef[ci], job->wblk[ci], job->hblk[ci]);
		coef_arrays[ci] = (jvirt_barray_ptr)arr[ci];
	}
	cinfo.max_h_samp_factor = maxh;
	cinfo.max_v_samp_factor = maxv;

	memset(&opts, 0, sizeof(opts));
	opts.flags = flags; opts.niter = niter; opts.threads = threads;
	opts.progprec = progprec; opts.progress = progress; opts.userdata = userdata;

	ret = do_quantsmooth(&cinfo, coef_arrays, &opts);

	job->up_wblk = job->up_hblk = 0;
	job->coef_up[0] = job->coef_up[1] = NULL;
	for (ci = 1; ci <= 2 && ci < job->ncomp; ci++) {
		if (coef_arrays[ci] != (jvirt_barray_ptr)arr[ci]) {
			fake_barray *u = (fake_barray*)coef_arrays[ci];
			job->coef_up[ci - 1] = (int16_t*)u->data;
			job->up_wblk = u->wblk; job->up_hblk = u->hblk;
		}
	}
	for (i = 0; i < mem.nextra; i++) {
		int used = 0;
		for (ci = 0; ci < 2; ci++)
			if (job->coef_up[ci] == (int16_t*)mem.extra[i]->data) used = 1;
		fake_barray_free(mem.extra[i], used);
	}
	for (ci = 0; ci < job->ncomp; ci++) {
		if (job->has_quant[ci])
			for (i = 0; i < 64; i++) job->quant[ci][i] = qt[ci].quantval[i];
		fake_barray_free(arr[ci], 1);
	}
	job->out_hsamp0 = comp[0].h_samp_factor;
	job->out_vsamp0 = comp[0].v_samp_factor;
	return ret;
}

void qsref_free(void *p) { free(p); }

/* ---- direct access to the reference's static block functions (KATs) ----- */

void qsref_idct_islow(const int16_t *coef, uint8_t *out, int stride) {
	JCOEF tmp[64]; memcpy(tmp, coef, sizeof(tmp));
	idct_islow(tmp, out, stride);
}

void qsref_idct_float(const float *in, float *out) {
	float tmp[64]; memcpy(tmp, in, sizeof(tmp));
	idct_float(tmp, out);
}

void qsref_fdct_float(const float *in, float *out) {
	float tmp[64]; memcpy(tmp, in, sizeof(tmp));
	fdct_float(tmp, out);
}

/* copy out the 64 per-coefficient weight tables, indexed by NATURAL index */
int qsref_tables(int flags, float *out) {
	int size = flags & JPEGQS_DIAGONALS ? 64 * 4 + 8 * 2 : 64 * 2 + 8 * 4, i;
	float **t = quantsmooth_init(flags);
	if (!t) return -1;
	for (i = 0; i < 64; i++) memcpy(out + (size_t)i * size, t[i], size * sizeof(float));
	free(t);
	return size;
}

/* the quant prep of reference quantsmooth.h:2497-2540 is inline in the plane
 * driver; it is exercised through qsref_do_quantsmooth.  For block-level KATs
 * the caller passes a ready 192-entry quantval[] (our oracle's quant_prep is
 * itself validated through the whole-plane comparison). */
void qsref_block(int16_t *coef, const uint16_t *quantval192,
		uint8_t *image, uint8_t *image2, int stride, int flags, int luma) {
	float **t = NULL;
	UINT16 qv[192];
	memcpy(qv, quantval192, sizeof(qv));
	if (!(flags & JPEGQS_LOW_QUALITY)) t = quantsmooth_init(flags);
	quantsmooth_block(coef, qv, image, image2, stride, flags, t, luma);
	if (t) free(t);
}

void qsref_fdct_clamp(float *buf, int16_t *coef, const uint16_t *quantval192) {
	UINT16 qv[192];
	memcpy(qv, quantval192, sizeof(qv));
	fdct_clamp(buf, coef, qv);
}

void qsref_upsample_row(int w1, int y0, int y1, uint8_t *image, uint8_t *image2, int stride,
		uint8_t *image1, int stride1, uint8_t *mem, int st, int ww, int ws, int hs) {
	upsample_row(w1, y0, y1, image, image2, stride, image1, stride1, mem, st, ww, ws, hs);
}

/* which compile-time variant is this? */
const char *qsref_variant(void) {
#if defined(NO_SIMD)
	return "none";
#elif defined(USE_AVX512)
	return "avx512";
#elif defined(USE_AVX2)
	return "avx2";
#elif defined(USE_SSE2)
	return "sse2";
#else
	return "generic";
#endif
}

int qsref_openmp(void) {
#ifdef _OPENMP
	return 1;
#else
	return 0;
#endif
}
