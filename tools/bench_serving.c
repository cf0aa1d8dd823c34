/*
 * bench_serving.c -- many independent jobs submitted from host threads straight
 * through the C ABI (qs_hip_do_quantsmooth is thread-safe: every call leases its
 * own streams and pooled buffers).  PCIe-inclusive, host arrays in and out.
 *
 *   bench_serving <job.bin> <flags> <niter> <threads> <jobs per thread> [batch]
 *
 * batch > 1: every thread hands its jobs over `batch` at a time through
 * qs_hip_do_quantsmooth_batch (independent jobs share one launch per pass).
 *
 * job.bin (written by tools/bench_serving.py): int32 ncomp, colorspace, width,
 * height; per component int32 wblk, hblk, hsamp, vsamp; uint16 quant[64];
 * then the coefficient arrays.  Every result is compared with the first one
 * (after the timed region).
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "jpegqs_hip.h"

typedef struct {
	qs_hip_job proto;
	size_t bytes[QS_HIP_MAXC];
	int16_t *pristine[QS_HIP_MAXC];
	int16_t *expect[QS_HIP_MAXC];
	int flags, niter, per, batch;
} shared_t;

typedef struct { shared_t *sh; int16_t **copies; int bad; } worker_t;
static pthread_barrier_t g_start, g_end;
static int g_warm;   /* untimed jobs per thread before the start barrier (pools, pinned memory, clocks) */

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

static int run_one(shared_t *sh, int16_t **bufs) {
	qs_hip_job job = sh->proto;
	for (int c = 0; c < job.ncomp; c++) job.coef[c] = bufs[c];
	int r = qs_hip_do_quantsmooth(&job, sh->flags, sh->niter, 0, NULL, NULL);
	for (int c = 0; c < 2; c++) if (job.coef_up[c]) qs_hip_free(job.coef_up[c]);
	return r;
}

static void run_batched(worker_t *w, int first, int last) {
	shared_t *sh = w->sh;
	int nc = sh->proto.ncomp, B = sh->batch;
	qs_hip_job *jobs = calloc(B, sizeof *jobs);
	qs_hip_job **ptrs = calloc(B, sizeof *ptrs);
	int *res = calloc(B, sizeof *res);
	for (int i0 = first; i0 < last; i0 += B) {
		int n = last - i0 < B ? last - i0 : B;
		for (int b = 0; b < n; b++) {
			jobs[b] = sh->proto; ptrs[b] = &jobs[b];
			for (int c = 0; c < nc; c++) jobs[b].coef[c] = w->copies[(size_t)(i0 + b) * nc + c];
		}
		if (qs_hip_do_quantsmooth_batch(ptrs, n, sh->flags, sh->niter, res) != 0) { w->bad += n; continue; }
		for (int b = 0; b < n; b++) {
			for (int c = 0; c < 2; c++) if (jobs[b].coef_up[c]) qs_hip_free(jobs[b].coef_up[c]);
			if (res[b] != 0) w->bad++;
		}
	}
	free(jobs); free(ptrs); free(res);
}

static void run_range(worker_t *w, int first, int last) {
	shared_t *sh = w->sh;
	int nc = sh->proto.ncomp;
	if (sh->batch > 1) { run_batched(w, first, last); return; }
	for (int i = first; i < last; i++)
		if (run_one(sh, w->copies + (size_t)i * nc) != 0) w->bad++;
}

static void *worker(void *p) {
	worker_t *w = p;
	run_range(w, 0, g_warm);
	pthread_barrier_wait(&g_start);
	run_range(w, g_warm, g_warm + w->sh->per);
	pthread_barrier_wait(&g_end);
	return NULL;
}

int main(int argc, char **argv) {
	if (argc < 6) { fprintf(stderr, "usage: %s job.bin flags niter threads jobs_per_thread\n", argv[0]); return 2; }
	shared_t sh; memset(&sh, 0, sizeof sh);
	sh.flags = atoi(argv[2]); sh.niter = atoi(argv[3]);
	int nthreads = atoi(argv[4]); sh.per = atoi(argv[5]);
	sh.batch = argc > 6 ? atoi(argv[6]) : 1;
	FILE *f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	int32_t hdr[4];
	if (fread(hdr, 4, 4, f) != 4) return 2;
	qs_hip_job *j = &sh.proto;
	j->ncomp = hdr[0]; j->colorspace = hdr[1]; j->image_width = hdr[2]; j->image_height = hdr[3];
	size_t nblk = 0;
	for (int c = 0; c < j->ncomp; c++) {
		int32_t g[4];
		if (fread(g, 4, 4, f) != 4) return 2;
		j->wblk[c] = g[0]; j->hblk[c] = g[1]; j->hsamp[c] = g[2]; j->vsamp[c] = g[3]; j->has_quant[c] = 1;
		if (fread(j->quant[c], 2, 64, f) != 64) return 2;
		sh.bytes[c] = (size_t)g[0] * g[1] * 128; nblk += (size_t)g[0] * g[1];
	}
	for (int c = 0; c < j->ncomp; c++) {
		sh.pristine[c] = malloc(sh.bytes[c]); sh.expect[c] = malloc(sh.bytes[c]);
		if (fread(sh.pristine[c], 1, sh.bytes[c], f) != sh.bytes[c]) return 2;
		memcpy(sh.expect[c], sh.pristine[c], sh.bytes[c]);
	}
	fclose(f);
	if (qs_hip_device_count() <= 0) { fprintf(stderr, "no HIP device: %s\n", qs_hip_last_error()); return 1; }
	if (run_one(&sh, sh.expect) != 0) { fprintf(stderr, "job failed: %s\n", qs_hip_last_error()); return 1; }

	g_warm = 2 * (sh.batch > 1 ? sh.batch : 4);
	int total = g_warm + sh.per;
	worker_t *ws = calloc(nthreads, sizeof *ws);
	for (int t = 0; t < nthreads; t++) {
		ws[t].sh = &sh; ws[t].copies = malloc(sizeof(int16_t *) * total * j->ncomp);
		for (int i = 0; i < total; i++) for (int c = 0; c < j->ncomp; c++) {
			int16_t *b = malloc(sh.bytes[c]); memcpy(b, sh.pristine[c], sh.bytes[c]);
			ws[t].copies[(size_t)i * j->ncomp + c] = b;
		}
	}
	{	/* warm the pools with one concurrent round */
		int16_t **tmp = malloc(sizeof(int16_t *) * j->ncomp);
		for (int c = 0; c < j->ncomp; c++) { tmp[c] = malloc(sh.bytes[c]); memcpy(tmp[c], sh.pristine[c], sh.bytes[c]); }
		run_one(&sh, tmp);
	}
	pthread_t *th = malloc(sizeof(pthread_t) * nthreads);
	pthread_barrier_init(&g_start, NULL, nthreads + 1);
	pthread_barrier_init(&g_end, NULL, nthreads + 1);
	for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, worker, &ws[t]);
	pthread_barrier_wait(&g_start);               /* every thread has done its warm-up jobs */
	double t0 = now();
	pthread_barrier_wait(&g_end);
	double dt = now() - t0;
	int bad = 0;
	for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); bad += ws[t].bad; }
	double n = (double)nthreads * sh.per;
	for (int t = 0; t < nthreads; t++)            /* every result against the first one, outside the timed region */
		for (int i = 0; i < total; i++) for (int c = 0; c < j->ncomp; c++)
			if (memcmp(ws[t].copies[(size_t)i * j->ncomp + c], sh.expect[c], sh.bytes[c])) { bad++; break; }
	printf("{\"threads\": %d, \"batch\": %d, \"jobs\": %.0f, \"images_per_s\": %.1f, \"mblocks_per_s\": %.2f, \"ms_per_job_per_thread\": %.3f, \"mismatches\": %d}\n",
	       nthreads, sh.batch, n, n / dt, n * nblk / dt / 1e6, dt / sh.per * 1e3, bad);
	return bad != 0;
}
