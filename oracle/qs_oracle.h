/*
 * qs_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the jpeg-quantsmooth coefficient-recovery path
 * (scalar / NO_SIMD arithmetic, "oracle A" of SURVEY.md section 8c).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (jpeg-quantsmooth_amd/) never links or
 * calls it.
 *
 * Parity pin: validated bit-for-bit against the compiled, unmodified
 * reference (oracle/_ref/libqsref_none.so, built by oracle/Makefile from
 * /root/reference) by tests/test_oracle_vs_ref.py, and against the golden
 * vectors under tests/golden/ that were generated from that same build.
 */
#ifndef QS_ORACLE_H
#define QS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QSO_MAXC 4

/* algorithm flag bits, same values as reference libjpegqs.h:16-23 */
enum {
	QSO_DIAGONALS = 1, QSO_JOINT_YUV = 2, QSO_UPSAMPLE_UV = 4, QSO_LOW_QUALITY = 8,
	QSO_NO_REBALANCE = 16, QSO_NO_REBALANCE_UV = 32
};

/* flat job: identical layout to qsref_job in ref_harness.c */
typedef struct {
	int32_t ncomp;
	int32_t colorspace;           /* 1 gray, 2 RGB, 3 YCbCr (J_COLOR_SPACE) */
	int32_t image_width, image_height;
	int32_t wblk[QSO_MAXC], hblk[QSO_MAXC];
	int32_t hsamp[QSO_MAXC], vsamp[QSO_MAXC];
	int32_t has_quant[QSO_MAXC];
	uint16_t quant[QSO_MAXC][64];
	int16_t *coef[QSO_MAXC];
	int16_t *coef_up[2];
	int32_t up_wblk, up_hblk;
	int32_t out_hsamp0, out_vsamp0;
} qso_job;

typedef int (*qso_progress_fn)(void *userdata, int cur, int max);

int qso_do_quantsmooth(qso_job *job, int flags, int niter, int threads,
		int progprec, qso_progress_fn progress, void *userdata);
void qso_free(void *p);

/* building blocks, exposed for known-answer tests */
void qso_quant_prep(const uint16_t q[64], uint16_t eff[64], int *all_le1, int *any_big);
void qso_idct_islow(const int16_t coef[64], uint8_t *out, int stride);
void qso_idct_float(const float in[64], float out[64]);
void qso_fdct_float(const float in[64], float out[64]);
int  qso_table_size(int flags);
int  qso_tables(int flags, float *out); /* 64 * size floats, natural index */
void qso_interval(int coef, int div, int *orig, int *lo, int *hi);
int  qso_interval_recip(int coef, int div, int *orig);
void qso_block(int16_t *coef, const uint16_t eff_quant[64],
		const uint8_t *image, const uint8_t *image2, int stride,
		int flags, int luma);
void qso_fdct_clamp(float *buf, int16_t *coef, const uint16_t eff_quant[64]);
/* band passes in the product's plane layout (CPU stand-in for the kernels in
 * the multi-process band tests) */
void qso_band_idct(int16_t *coef, int wblk, int hblk, const uint16_t rawq[64], int first,
		uint8_t *plane, int pitch, int apron_x, int rep_top, int rep_bot, int *bad);
void qso_band_smooth(int16_t *coef, int wblk, int hblk, const uint16_t rawq[64],
		const uint8_t *plane, int pitch, int apron_x, int flags, int luma, int final_clamp);
void qso_band_smooth_rows(int16_t *coef, int wblk, int hblk, const uint16_t rawq[64],
		const uint8_t *plane, int pitch, int apron_x, int flags, int luma, int final_clamp, int row0, int row1);

#ifdef __cplusplus
}
#endif
#endif
