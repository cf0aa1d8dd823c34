/*
 * qs_cpu.c -- CPU back end of libjpegqs.so: what do_quantsmooth() runs when no HIP device is visible
 * (the reference always produces a smoothed image, reference
 * quantsmooth.h:2404-2878 behind the dispatcher libjpegqs.c:80-156).  See qs_cpu.h for the contract.
 *
 * This is product code, written for this library: it shares nothing with oracle/ (the test checker) and is
 * compared, like the GPU path, against the compiled reference (tests/test_cpu_backend.py).
 *
 * Shape: the GPU kernels' own -- ONE 8x8 BLOCK PER LANE.  The reference's sums a2 += x*y, a3 += y*y are
 * order-sensitive FP32 chains, so a bit-exact implementation cannot spread one block's
 * terms over SIMD lanes the way the reference's AVX paths do (those differ from its scalar path).  Instead QS_NL
 * blocks of one block row sit side by side in GCC generic vectors (vector_size = QS_NL * 4 bytes: one zmm, two
 * ymm or four xmm registers, chosen per function clone at load time), every lane running its block's chain in
 * the scalar order.  IEEE binary32, one rounding per operation: build with -ffp-contract=off -fwrapv, never
 * -ffast-math.  The term order, weights and integer semantics are SURVEY.md Appendix A's; each function cites
 * the reference lines whose BEHAVIOUR it reproduces.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "qs_cpu.h"

#ifndef QS_NL
#define QS_NL 16
#endif
typedef float   vf __attribute__((vector_size(QS_NL * 4)));
typedef int32_t vi __attribute__((vector_size(QS_NL * 4)));

#if defined(__x86_64__) && defined(__GNUC__) && !defined(QS_CPU_NO_CLONES)
#define LANE_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define LANE_CLONES
#endif

/* JPEGQS_* algorithm bits (include/libjpegqs.h; reference libjpegqs.h:16-23) */
enum { F_DIAGONALS = 1, F_JOINT_YUV = 2, F_UPSAMPLE_UV = 4, F_LOW_QUALITY = 8, F_NO_REBALANCE = 16, F_NO_REBALANCE_UV = 32 };
#define COLORSPACE_YCBCR 3   /* JCS_YCbCr */

/* zigzag position -> natural index (ITU T.81 figure 5; the walk of reference quantsmooth.h:1403-1404) */
static const uint8_t zigzag[64] = {
	 0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
	12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
	35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
	58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};

/* The own-pixel buffer is re-derived from the coefficients when the walk k = 63 .. 1 enters a new anti-diagonal
 * (reference quantsmooth.h:313-322, 1406-1408: k = 63,62,60,57,53,48,42,35,27,20,14,9,5,2). */
static int enters_antidiagonal(int k) {
	int a = zigzag[k], b;
	if (k == 63) return 1;
	b = zigzag[k + 1];
	return (a >> 3) + (a & 7) != (b >> 3) + (b & 7);
}

/* float -> int32 the way the reference's `int r = roundf(x)` behaves on x86-64 (cvttss2si): NaN and anything
 * outside int32 give INT_MIN (the a3 == 0 blocks depend on it) */
static inline int32_t to_int_x86(float v) {
	if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
	return (int32_t)v;
}
/* to_int_x86(roundf(v)) without the libm call: truncate, then step away from zero when the (exactly
 * representable) remainder reaches one half */
static inline int32_t round_to_int_x86(float v) {
	int32_t t;
	float rem;
	if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
	t = (int32_t)v;
	rem = v - (float)t;
	if (rem >= 0.5f) t++; else if (rem <= -0.5f) t--;
	return t;
}

/* nearest multiple of `div` (ties away from zero) and the interval of integers that quantise to it: the exact
 * division form, exhaustively equal to the reference's reciprocal tables (reference quantsmooth.h:324-341,
 * 1552-1557) */
static inline void quant_interval(int coef, int div, int *mid, int *lo, int *hi) {
	const int up = div >> 1, dn = (div - 1) >> 1;
	const int m = (coef + (coef < 0 ? -up : up)) / div * div;
	*mid = m;
	*hi = m + (m < 0 ? up : dn);
	*lo = m - (m > 0 ? up : dn);
}

/* ------------------------------------------------------------------------------------------------------------
 * DCTs
 * ---------------------------------------------------------------------------------------------------------- */

/* LL&M 13-bit fixed-point inverse butterfly on 8 int32 values (wrapping arithmetic), reference idct.h:41-52,
 * 57-89; generated for scalars and for lane vectors.  Integer ring arithmetic: only the two shift points of the
 * callers matter, and the reference's zero-AC shortcuts (:487-499, 519-533) are special cases of it. */
#define DEF_IDCT8(NAME, T) \
static inline __attribute__((always_inline)) void NAME(const T *in, T *out) { \
	T s = (in[2] + in[6]) * 4433, e2 = s - in[6] * 15137, e3 = s + in[2] * 6270; \
	T e0 = (in[0] + in[4]) * 8192, e1 = (in[0] - in[4]) * 8192; \
	T b0 = e0 + e3, b3 = e0 - e3, b1 = e1 + e2, b2 = e1 - e2; \
	T t0 = in[7], t1 = in[5], t2 = in[3], t3 = in[1]; \
	T z1 = t0 + t3, z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3, z5 = (z3 + z4) * 9633; \
	t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299; \
	z1 *= 7373; z2 *= 20995; z3 = z5 - z3 * 16069; z4 = z5 - z4 * 3196; \
	t0 += z3 - z1; t1 += z4 - z2; t2 += z3 - z2; t3 += z4 - z1; \
	out[0] = b0 + t3; out[7] = b0 - t3; out[1] = b1 + t2; out[6] = b1 - t2; \
	out[2] = b2 + t1; out[5] = b2 - t1; out[3] = b3 + t0; out[4] = b3 - t0; \
}
DEF_IDCT8(idct8_scalar, int32_t)
DEF_IDCT8(idct8_lanes, vi)

/* one block: int16 coefficients -> 8 rows of u8 at `pitch` (reference idct.h:468-539: pass 1 down the columns
 * with (x + 1024) >> 11, pass 2 along the rows with (x + (257 << 17)) >> 18 clamped to 0..255) */
static void idct_block(const int16_t *coef, uint8_t *out, size_t pitch) {
	int32_t ws[64], in[8], o[8];
	int i, j;
	for (i = 0; i < 8; i++) {
		for (j = 0; j < 8; j++) in[j] = coef[8 * j + i];
		idct8_scalar(in, o);
		for (j = 0; j < 8; j++) ws[8 * j + i] = (o[j] + 1024) >> 11;
	}
	for (i = 0; i < 8; i++) {
		idct8_scalar(ws + 8 * i, o);
		for (j = 0; j < 8; j++) {
			int32_t z = (o[j] + (257 << 17)) >> 18;
			out[i * pitch + j] = (uint8_t)(z < 0 ? 0 : z > 255 ? 255 : z);
		}
	}
}

/* the float LL&M pair, operation by operation as SURVEY.md Appendix A.3 / A.8 state them (float results depend
 * on the order): inverse reference idct.h:565-604, forward idct.h:606-628, 895-916 */
static void idct8_float(const float *in, int is, float *out, int os, float scale) {
	float z2 = in[2 * is], z3 = in[6 * is], z1 = (z2 + z3) * 0.541196100f;
	float t2 = z1 - z3 * 1.847759065f, t3 = z1 + z2 * 0.765366865f;
	float t0, t1, t4, t5, t6, t7, z4, z5;
	z2 = in[0]; z3 = in[4 * is];
	t0 = z2 + z3; t1 = z2 - z3;
	t4 = t0 + t3; t7 = t0 - t3; t5 = t1 + t2; t6 = t1 - t2;
	t0 = in[7 * is]; t1 = in[5 * is]; t2 = in[3 * is]; t3 = in[is];
	z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; z4 = t1 + t3;
	z5 = (z3 + z4) * 1.175875602f;
	t0 *= 0.298631336f; t1 *= 2.053119869f; t2 *= 3.072711026f; t3 *= 1.501321110f;
	z1 *= 0.899976223f; z2 *= 2.562915447f; z3 *= 1.961570560f; z4 *= 0.390180644f;
	z3 -= z5; t0 -= z1 + z3; t2 -= z2 + z3;
	z4 -= z5; t1 -= z2 + z4; t3 -= z1 + z4;
	out[0] = (t4 + t3) * scale; out[7 * os] = (t4 - t3) * scale;
	out[os] = (t5 + t2) * scale; out[6 * os] = (t5 - t2) * scale;
	out[2 * os] = (t6 + t1) * scale; out[5 * os] = (t6 - t1) * scale;
	out[3 * os] = (t7 + t0) * scale; out[4 * os] = (t7 - t0) * scale;
}
static void fdct8_float(const float *in, int is, float *out, int os, float scale) {
	float t0 = in[0] + in[7 * is], t7 = in[0] - in[7 * is];
	float t1 = in[is] + in[6 * is], t6 = in[is] - in[6 * is];
	float t2 = in[2 * is] + in[5 * is], t5 = in[2 * is] - in[5 * is];
	float t3 = in[3 * is] + in[4 * is], t4 = in[3 * is] - in[4 * is];
	float z1 = t0 + t3, z4 = t0 - t3, z2 = t1 + t2, z3 = t1 - t2, z5;
	out[0] = (z1 + z2) * scale; out[4 * os] = (z1 - z2) * scale;
	z1 = (z3 + z4) * 0.541196100f;
	out[2 * os] = (z1 + z4 * 0.765366865f) * scale;
	out[6 * os] = (z1 - z3 * 1.847759065f) * scale;
	z1 = t4 + t7; z2 = t5 + t6; z3 = t4 + t6; z4 = t5 + t7;
	z5 = (z3 + z4) * 1.175875602f;
	t4 *= 0.298631336f; t5 *= 2.053119869f; t6 *= 3.072711026f; t7 *= 1.501321110f;
	z1 *= 0.899976223f; z2 *= 2.562915447f;
	z3 = z3 * 1.961570560f - z5; z4 = z4 * 0.390180644f - z5;
	out[7 * os] = (t4 - (z1 + z3)) * scale; out[5 * os] = (t5 - (z2 + z4)) * scale;
	out[3 * os] = (t6 - (z2 + z3)) * scale; out[os] = (t7 - (z1 + z4)) * scale;
}
/* the first pass carries no scale in the reference (a multiplication by 1.0f is exact, so `scale` may be 1) */
static void idct_float_block(float *blk) {
	float ws[64]; int i;
	for (i = 0; i < 8; i++) idct8_float(blk + i, 8, ws + i, 8, 1.0f);
	for (i = 0; i < 8; i++) idct8_float(ws + 8 * i, 1, blk + 8 * i, 1, 0.125f);
}
static void fdct_float_block(float *blk) {
	float ws[64]; int i;
	for (i = 0; i < 8; i++) fdct8_float(blk + i, 8, ws + i, 8, 1.0f);
	for (i = 0; i < 8; i++) fdct8_float(ws + 8 * i, 1, blk + 8 * i, 1, 0.125f);
}

/* fdct_clamp: forward DCT of the predicted pixels, each coefficient rounded and kept inside the interval of its
 * CURRENT value (reference quantsmooth.h:343-347, 551-561) */
static void fdct_clamp(float *fbuf, int16_t *coef, const uint16_t *q) {
	int i, mid, lo, hi;
	fdct_float_block(fbuf);
	for (i = 0; i < 64; i++) {
		int32_t add = to_int_x86(roundf(fbuf[i]));
		quant_interval(coef[i], q[i], &mid, &lo, &hi);
		if (add > hi) add = hi;
		if (add < lo) add = lo;
		coef[i] = (int16_t)add;
	}
}

/* ------------------------------------------------------------------------------------------------------------
 * pixel planes: (w x h) u8 with a one-pixel apron that holds clamp-to-edge copies -- what the reference's
 * border columns / rows amount to (reference quantsmooth.h:2612-2620)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct { uint8_t *mem; int w, h; size_t pitch; } plane;
static inline uint8_t *at(const plane *p, int x, int y) { return p->mem + (size_t)(y + 1) * p->pitch + (x + 1); }
static int plane_new(plane *p, int w, int h) {
	p->w = w; p->h = h; p->pitch = ((size_t)w + 2 + 15) & ~(size_t)15;
	p->mem = (uint8_t*)malloc(p->pitch * ((size_t)h + 2));
	return p->mem != NULL;
}
static void plane_drop(plane *p) { free(p->mem); p->mem = NULL; }
/* rows 0 .. h_valid-1 hold w_valid pixels: spread the last column to the right edge and apron, the last row to
 * the bottom edge and apron, first column / row into the left / top apron */
static void plane_extend(plane *p, int w_valid, int h_valid) {
	int x, y;
	for (y = 0; y < h_valid; y++) {
		uint8_t *r = at(p, 0, y), e = r[w_valid - 1];
		r[-1] = r[0];
		for (x = w_valid; x <= p->w; x++) r[x] = e;
	}
	memcpy(at(p, -1, -1), at(p, -1, 0), p->pitch);
	for (y = h_valid; y <= p->h; y++) memcpy(at(p, -1, y), at(p, -1, h_valid - 1), p->pitch);
}

/* ------------------------------------------------------------------------------------------------------------
 * the recovery loop (reference quantsmooth_block, quantsmooth.h:1396-1565, + rebalance :1566-1568, 1823-1848)
 * ---------------------------------------------------------------------------------------------------------- */

/* Per coefficient: the list of its terms in the reference's scalar order -- (own pixel, other pixel, weight).
 * Pixel slots: 0..63 the block's own pixels, 64..71 the row above, 72..79 the row below, 80..87 the column to
 * the left, 88..95 the column to the right (the neighbours' edge pixels, frozen for the iteration; reference
 * :1396-1401).  Weights: differences of the coefficient's basis image T = idct_float(e_i), or b*T on the block
 * edge with b = 2 (4 with DIAGONALS) (reference :251-301). */
#define MAX_TERMS 242
typedef struct {
	int n[64];
	uint8_t own[64][MAX_TERMS], other[64][MAX_TERMS];
	float w[64][MAX_TERMS];
} term_table;

static term_table *terms_build(int flags) {
	term_table *tt = (term_table*)malloc(sizeof(term_table));
	const float b = (flags & F_DIAGONALS) ? 4.0f : 2.0f;
	int i, x, y;
	if (!tt) return NULL;
	for (i = 0; i < 64; i++) {
		float T[64];
		int n = 0;
#define TERM(A, B, W) do { tt->own[i][n] = (uint8_t)(A); tt->other[i][n] = (uint8_t)(B); tt->w[i][n] = (W); n++; } while (0)
		memset(T, 0, sizeof(T)); T[i] = 1.0f;
		idct_float_block(T);
		if (i & 7)                                       /* horizontal neighbours inside the block */
			for (y = 0; y < 8; y++) for (x = 0; x < 7; x++) TERM(8 * y + x, 8 * y + x + 1, T[8 * y + x] - T[8 * y + x + 1]);
		for (x = 0; x < 8; x++) TERM(x, 64 + x, T[x] * b);                   /* across the four block edges */
		for (x = 0; x < 8; x++) TERM(56 + x, 72 + x, T[56 + x] * b);
		for (y = 0; y < 8; y++) TERM(8 * y, 80 + y, T[8 * y] * b);
		for (y = 0; y < 8; y++) TERM(8 * y + 7, 88 + y, T[8 * y + 7] * b);
		if (i > 7)                                       /* vertical neighbours inside the block */
			for (y = 0; y < 7; y++) for (x = 0; x < 8; x++) TERM(8 * y + x, 8 * y + x + 8, T[8 * y + x] - T[8 * y + x + 8]);
		if (flags & F_DIAGONALS)
			for (y = 0; y < 7; y++) for (x = 0; x < 7; x++) {
				int p = 8 * y + x;
				TERM(p, p + 9, T[p] - T[p + 9]);
				TERM(p + 1, p + 8, T[p + 1] - T[p + 8]);
			}
#undef TERM
		tt->n[i] = n;
	}
	return tt;
}

static void rebalance_block(int16_t *coef, const uint16_t *q);

/* own pixels of QS_NL blocks from their coefficients, as floats (u8 values: every later difference is exact);
 * inlined into each clone of recover_lanes */
static inline __attribute__((always_inline)) void idct_lanes(const vi *c, vf *pix) {
	vi ws[64], in[8], o[8];
	int i, j;
	for (i = 0; i < 8; i++) {
		for (j = 0; j < 8; j++) in[j] = c[8 * j + i];
		idct8_lanes(in, o);
		for (j = 0; j < 8; j++) ws[8 * j + i] = (o[j] + 1024) >> 11;
	}
	for (i = 0; i < 8; i++) {
		idct8_lanes(ws + 8 * i, o);
		for (j = 0; j < 8; j++) {
			vi z = (o[j] + (257 << 17)) >> 18;
			z &= ~(z >> 31);                             /* < 0 -> 0 */
			z |= (255 - z) >> 31;                        /* > 255 -> all ones */
			z &= 255;
			pix[8 * i + j] = __builtin_convertvector(z, vf);
		}
	}
}

/* QS_NL blocks, lane l = block cf[l] whose top-left pixel is px[l] in a plane of row pitch `pitch`.  Lanes beyond
 * the last real block repeat it (their results are not stored). */
LANE_CLONES
static void recover_lanes(const term_table *tt, const uint16_t *q, int16_t *const *cf, const uint8_t *const *px,
		size_t pitch, int rebalance, int nvalid) {
	vi c[64];
	vf pix[96];
	int i, k, l, z, stale = 1;
	for (l = 0; l < QS_NL; l++) {
		const int16_t *src = cf[l];
		const uint8_t *p = px[l];
		for (i = 0; i < 64; i++) c[i][l] = src[i];
		for (z = 0; z < 8; z++) {
			pix[64 + z][l] = p[z - (ptrdiff_t)pitch];
			pix[72 + z][l] = p[z + 8 * pitch];
			pix[80 + z][l] = p[z * pitch - 1];
			pix[88 + z][l] = p[z * pitch + 8];
		}
	}
	for (k = 63; k > 0; k--) {
		const int n = tt->n[i = zigzag[k]];
		const uint8_t *own = tt->own[i], *other = tt->other[i];
		const float *w = tt->w[i];
		const float range = (float)(2 * q[i]);
		vf a2 = {0}, a3 = {0}, quot;
		int t;
		if (stale && enters_antidiagonal(k)) { idct_lanes(c, pix); stale = 0; }
		/* quantiser 1: the interval below is the single point the coefficient holds, nothing can change (the GPU
		 * kernels skip these too, QS_REC_Q1 in qs_device.h).  After the refresh: that belongs to the anti-diagonal. */
		if (q[i] == 1) continue;
		for (t = 0; t < n; t++) {                        /* reference :1519-1520, one block per lane */
			vf d = pix[own[t]] - pix[other[t]];
			vf m = range - (vf)((vi)d & 0x7fffffff);
			m = (vf)((vi)m & ~((vi)m >> 31));            /* max(m, 0): a negative (or -0) m has its sign bit set */
			m *= m;
			{
				vf xs = d * m, ys = w[t] * m;
				a2 += xs * ys;
				a3 += ys * ys;
			}
		}
		quot = a2 / a3;                                  /* 0/0 -> NaN -> INT_MIN below, as on x86 */
		for (l = 0; l < QS_NL; l++) {                    /* reference :1548-1564 */
			const int32_t r = round_to_int_x86(quot[l]);
			if (r) {
				const int32_t old = c[i][l];
				int32_t add = (int32_t)((uint32_t)old - (uint32_t)r);
				int mid, lo, hi;
				quant_interval(old, q[i], &mid, &lo, &hi);
				if (add > hi) add = hi;
				if (add < lo) add = lo;
				c[i][l] = add;
				stale |= add != old;
			}
		}
	}
	for (l = 0; l < nvalid; l++) {
		int16_t *dst = cf[l];
		for (i = 0; i < 64; i++) dst[i] = (int16_t)c[i][l];
		if (rebalance) rebalance_block(dst, q);
	}
}

/* rebalance, one block (reference :1823-1848: int64 sums over k = 1..63, DC untouched); also what the LOW_QUALITY
 * and predictor-only routes end with (reference :1161-1178 -> `end:` :1566-1568) */
static void rebalance_block(int16_t *coef, const uint16_t *q) {
	int64_t m0 = 0, m1 = 0;
	int mid[64], lo, hi, k;
	for (k = 1; k < 64; k++) {
		quant_interval(coef[k], q[k], &mid[k], &lo, &hi);
		m0 += (int64_t)coef[k] * mid[k];
		m1 += (int64_t)mid[k] * mid[k];
	}
	if (m1 > m0) {
		const int mul = (int)(((m1 << 13) + (m0 >> 1)) / m0);
		for (k = 1; k < 64; k++) {
			const int up = q[k] >> 1, dn = (q[k] - 1) >> 1, m = mid[k];
			int32_t add = (int32_t)((uint32_t)(int32_t)coef[k] * (uint32_t)mul + 0x1000u) >> 13;
			hi = m + (m < 0 ? up : dn); lo = m - (m > 0 ? up : dn);
			if (add > hi) add = hi;
			if (add < lo) add = lo;
			coef[k] = (int16_t)add;
		}
	}
}

/* ------------------------------------------------------------------------------------------------------------
 * chroma from luma: the 3x3 weighted regression of SURVEY.md Appendix A.9 (weights 4 centre, 2 edges, 1
 * corners, int32 sums), used by the JOINT_YUV predictor (reference quantsmooth.h:893-921) and by the guided
 * upsampling (:2133-2158)
 * ---------------------------------------------------------------------------------------------------------- */
static inline float regress(const uint8_t *A, size_t pa, const uint8_t *B, size_t pb, int32_t *sumA_out, int32_t *sumB_out) {
	int32_t sA, sB, sAA, sAB;
	float scale;
	const ptrdiff_t ra = (ptrdiff_t)pa, rb = (ptrdiff_t)pb;
#define TAP(dx, dy) { const int32_t a = A[(dy) * ra + (dx)], b_ = B[(dy) * rb + (dx)]; sA += a; sAA += a * a; sB += b_; sAB += a * b_; }
#define DOUBLE_ALL sA += sA; sB += sB; sAA += sAA; sAB += sAB;
	sA = sB = sAA = sAB = 0;
	TAP(0, 0) DOUBLE_ALL
	TAP(0, -1) TAP(-1, 0) TAP(1, 0) TAP(0, 1) DOUBLE_ALL
	TAP(-1, -1) TAP(1, -1) TAP(-1, 1) TAP(1, 1)
#undef TAP
#undef DOUBLE_ALL
	sAA = sAA * 16 - sA * sA;
	sAB = sAB * 16 - sA * sB;
	scale = (float)sAA;
	if (sAA) scale = (float)sAB / scale;
	scale = scale < -16.0f ? -16.0f : scale;
	scale = scale > 16.0f ? 16.0f : scale;
	*sumA_out = sA; *sumB_out = sB;
	return scale;
}

/* JOINT_YUV: predict the chroma block from the low-resolution luma, then fdct_clamp (reference :893-921) */
static void joint_block(int16_t *coef, const uint16_t *q, const uint8_t *chroma, size_t pc, const uint8_t *luma, size_t pl) {
	float fbuf[64];
	int x, y;
	for (y = 0; y < 8; y++) for (x = 0; x < 8; x++) {
		int32_t sA, sB;
		const uint8_t *A = luma + y * pl + x;
		const float scale = regress(A, pl, chroma + y * pc + x, pc, &sA, &sB);
		float a = ((float)(A[0] * 16 - sA) * scale + (float)sB) * 0.0625f;
		a = (a < 0 ? 0 : a) - 128.0f;
		fbuf[8 * y + x] = a > 128.0f ? 128.0f : a;
	}
	fdct_clamp(fbuf, coef, q);
}

/* LOW_QUALITY: range from the block's own coefficients, 8-neighbour range filter, fdct_clamp
 * (reference quantsmooth.h:924-938, 1161-1178) */
static void lowq_block(int16_t *coef, const uint16_t *q, const uint8_t *px, size_t pitch) {
	float fbuf[64], range = 0;
	const float c0 = 2, c1 = c0 * sqrtf(0.5f);
	const ptrdiff_t p = (ptrdiff_t)pitch;
	int sum = 0, x, y;
	for (x = 1; x < 64; x++) {
		int a = coef[x]; a = a < 0 ? -a : a;
		range += (float)(q[x] * a); sum += a;
	}
	if (sum) range *= 4.0f / (float)sum;
	if (range > 128.0f) range = 128.0f;
	range = roundf(range);
	for (y = 0; y < 8; y++) for (x = 0; x < 8; x++) {
		const uint8_t *c = px + y * p + x;
		int a = c[0];
		float a0 = 0, an = 0;
#define TAP(CW, dx, dy) { const float t0 = (float)(a - c[(dy) * p + (dx)]); float t = range - fabsf(t0), aw; \
			t = t < 0 ? 0 : t; t *= t; aw = (CW) * t; a0 += t0 * t * aw; an += aw * aw; }
		TAP(c1, -1, -1) TAP(c0, 0, -1) TAP(c1, 1, -1)
		TAP(c0, -1, 0)                 TAP(c0, 1, 0)
		TAP(c1, -1, 1)  TAP(c0, 0, 1)  TAP(c1, 1, 1)
#undef TAP
		if (an > 0.0f) a = to_int_x86((float)a - a0 / an);   /* the reference's `int a -= float` truncates (:1173) */
		fbuf[8 * y + x] = (float)(a - 128);
	}
	fdct_clamp(fbuf, coef, q);
}

/* ------------------------------------------------------------------------------------------------------------
 * the job (reference do_quantsmooth, quantsmooth.h:2404-2878, on the flat job of include/jpegqs_hip.h)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
	const qs_hip_job *job;
	int16_t *const *const *rows;
} coef_view;
static inline int16_t *block_at(const coef_view *v, int ci, int by, int bx) {
	if (v->rows) return v->rows[ci][by] + (size_t)bx * 64;
	return v->job->coef[ci] + ((size_t)by * v->job->wblk[ci] + bx) * 64;
}

/* luma box-averaged down to chroma resolution and spread to the plane's edges (reference :2753-2815; the 4:2:0
 * fast path :2774-2785 is the ws = hs = 2 case of the general mean) */
static void downsample_luma(const plane *Y, plane *L, int ws, int hs) {
	const int w1 = (Y->w + ws - 1) / ws, h1 = (Y->h + hs - 1) / hs;
	int y;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
	for (y = 0; y < h1; y++) {
		int x, xx, yy, hh = Y->h - y * hs;
		hh = hh < hs ? hh : hs;
		for (x = 0; x < w1; x++) {
			int wv = Y->w - x * ws, sum = 0, div;
			wv = wv < ws ? wv : ws; div = wv * hh;
			for (yy = 0; yy < hh; yy++) for (xx = 0; xx < wv; xx++) sum += *at(Y, x * ws + xx, y * hs + yy);
			*at(L, x, y) = (uint8_t)((sum + div / 2) / div);
		}
	}
	plane_extend(L, w1, h1);
}

/* UPSAMPLE_UV: chroma plane C -> full-resolution pixels guided by luma (reference :1851-1864, 2133-2158,
 * 2363-2393 and the strip loop :2724-2730), then re-encoded block by block (:2735-2750) into `out`
 * (Y's block geometry).  Only columns < Y->w and rows < Y->h of the pixel buffer are ever read back, so only
 * those are produced.  The reference's right-edge replicate reaches the rows of the FIRST 8-row strip only (its
 * loop bounds are relative to a pointer it has already advanced, :1860-1861, 2390-2393): reproduced. */
static int upsample_chroma(const plane *C, const plane *L, const plane *Y, int16_t *out, int wblk, int hblk,
		int image_width, int image_height, int ws, int hs) {
	const int w1 = (image_width + ws - 1) / ws, h1 = (image_height + hs - 1) / hs;
	const int ww = Y->w, hh = Y->h, wcalc = ((w1 + 7) & ~7);
	const size_t st = (size_t)ww;
	uint8_t *mem = (uint8_t*)malloc(st * (size_t)hh);
	int y, by;
	if (!mem) return 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8)
#endif
	for (y = 0; y < h1; y++) {
		int x, xx, yy;
		if (y * hs >= hh) continue;
		for (x = 0; x < wcalc && x * ws < ww; x++) {
			int32_t sA, sB;
			const float scale = regress(at(L, x, y), L->pitch, at(C, x, y), C->pitch, &sA, &sB);
			const float offset = (float)*at(C, x, y) - (float)*at(L, x, y) * scale + 0.5f;
			for (yy = 0; yy < hs && y * hs + yy < hh; yy++)
				for (xx = 0; xx < ws && x * ws + xx < ww; xx++) {
					const int32_t v = to_int_x86((float)*at(Y, x * ws + xx, y * hs + yy) * scale + offset);
					mem[(size_t)(y * hs + yy) * st + x * ws + xx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
				}
		}
		if (y < 8)
			for (yy = 0; yy < hs && y * hs + yy < hh; yy++) {
				uint8_t *row = mem + (size_t)(y * hs + yy) * st;
				for (x = w1 * ws; x < ww; x++) row[x] = row[w1 * ws - 1];
			}
	}
	for (y = h1 * hs; y < hh; y++) memcpy(mem + (size_t)y * st, mem + (size_t)(h1 * hs - 1) * st, st);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
	for (by = 0; by < hblk; by++) {
		int bx, i, j;
		for (bx = 0; bx < wblk; bx++) {
			float f[64];
			const uint8_t *p = mem + (size_t)by * 8 * st + bx * 8;
			int16_t *dst = out + ((size_t)by * wblk + bx) * 64;
			for (i = 0; i < 8; i++) for (j = 0; j < 8; j++) f[8 * i + j] = (float)(p[i * st + j] - 128);
			fdct_float_block(f);
			for (i = 0; i < 64; i++) dst[i] = (int16_t)to_int_x86(roundf(f[i]));
		}
	}
	free(mem);
	return 1;
}

void qs_cpu_free(void *p) { free(p); }
int qs_cpu_lanes(void) { return QS_NL; }
const char *qs_cpu_isa(void) {
#if defined(__x86_64__) && defined(__GNUC__) && !defined(QS_CPU_NO_CLONES)
	__builtin_cpu_init();
	if (__builtin_cpu_supports("avx512f")) return "avx512f";
	if (__builtin_cpu_supports("avx2")) return "avx2";
#endif
	return "generic";
}

int qs_cpu_do_quantsmooth(qs_hip_job *job, int16_t *const *const *rows, int flags, int niter, int threads,
		int progprec, qs_hip_progress_fn progress, void *userdata) {
	coef_view view;
	term_table *tt = NULL;
	plane Yfull = { NULL, 0, 0, 0 }, Llow = { NULL, 0, 0, 0 };   /* the reference's image1 / image2 */
	int llow_is_luma_plane = 0;
	int16_t *up[2] = { NULL, NULL };
	int ci, i, stop = 0, need_lowres;
	int prog_next = 0, prog_max = 0, prog_thr = 0;
#ifdef _OPENMP
	int old_threads = -1;
#endif

	if (!job || job->ncomp < 1 || job->ncomp > QS_HIP_MAXC) return QS_HIP_EINVAL;
	for (ci = 0; ci < job->ncomp; ci++) {
		if (job->wblk[ci] <= 0 || job->hblk[ci] <= 0) return QS_HIP_EINVAL;
		if (rows ? !rows[ci] : !job->coef[ci]) return QS_HIP_EINVAL;
		if (job->hsamp[ci] < 1 || job->hsamp[ci] > 4 || job->vsamp[ci] < 1 || job->vsamp[ci] > 4) return QS_HIP_EINVAL;
		if ((long long)job->wblk[ci] * job->hblk[ci] > (1ll << 27)) return QS_HIP_EINVAL;
	}
	view.job = job; view.rows = rows;
	job->up_wblk = job->up_hblk = 0; job->coef_up[0] = job->coef_up[1] = NULL;
	job->out_hsamp0 = job->hsamp[0]; job->out_vsamp0 = job->vsamp[0];

	/* reference :2447-2458.  (libjpeg reports JCS_YCbCr with exactly three components; a flat job with four has
	 * no meaning for the two replacement arrays, so the coupling needs ncomp == 3, as in the GPU job layer) */
	need_lowres = (flags & (F_JOINT_YUV | F_UPSAMPLE_UV)) && job->colorspace == COLORSPACE_YCBCR && job->ncomp == 3 &&
			job->hsamp[1] == 1 && job->vsamp[1] == 1 && job->hsamp[2] == 1 && job->vsamp[2] == 1;
	if (niter < 0) niter = 0;
	if (niter > 100) niter = 100;
	if (niter <= 0 && !((flags & F_UPSAMPLE_UV) && need_lowres)) return 0;
	if (need_lowres) {
		/* the chroma planes must cover the box-averaged luma (true for every geometry libjpeg produces) */
		const int ws = job->hsamp[0], hs = job->vsamp[0];
		if (ws == 1 && hs == 1 && (job->wblk[1] != job->wblk[0] || job->hblk[1] != job->hblk[0])) return QS_HIP_EINVAL;
		if ((job->wblk[0] * 8 + ws - 1) / ws > job->wblk[1] * 8 || (job->hblk[0] * 8 + hs - 1) / hs > job->hblk[1] * 8 ||
				job->wblk[2] != job->wblk[1] || job->hblk[2] != job->hblk[1]) return QS_HIP_EINVAL;
		if ((flags & F_UPSAMPLE_UV) && (job->image_width <= 0 || job->image_height <= 0 ||
				(job->image_width + ws - 1) / ws > job->wblk[1] * 8 || (job->image_height + hs - 1) / hs > job->hblk[1] * 8 ||
				job->image_width > job->wblk[0] * 8 || job->image_height > job->hblk[0] * 8)) return QS_HIP_EINVAL;
	}

	if (!(flags & F_LOW_QUALITY)) {
		tt = terms_build(flags);
		if (!tt) return 0;                               /* reference :2463 */
	}
#ifdef _OPENMP
	if (threads >= 0) {                                  /* reference :2467-2472 */
		old_threads = omp_get_max_threads();
		omp_set_num_threads(threads ? threads : omp_get_num_procs());
	}
#else
	(void)threads;
#endif
	if (progress) {                                      /* reference :2474-2482 */
		for (ci = 0; ci < job->ncomp; ci++) prog_max += job->hblk[ci] * job->vsamp[ci] * niter;
		if (progprec == 0) progprec = 20;
		if (progprec < 0) progprec = prog_max;
		prog_thr = (int)((unsigned)(prog_max + progprec - 1) / (unsigned)progprec);
	}

	for (ci = 0; ci < job->ncomp; ci++) {
		const int wb = job->wblk[ci], hb = job->hblk[ci];
		const uint16_t *rawq = job->quant[ci];
		uint16_t q[64];
		const int luma = !ci || job->colorspace != COLORSPACE_YCBCR;   /* reference :2639 */
		const int rebalance = !(flags & F_NO_REBALANCE) && (luma || !(flags & F_NO_REBALANCE_UV));
		int iters = niter, extra = 0, it, acc = 0, by;
		int prog_cur = prog_next;
		const int prog_inc = job->vsamp[ci];
		plane P = { NULL, 0, 0, 0 };
		int keep_plane = 0;

		prog_next += hb * prog_inc * niter;
		if (!job->has_quant[ci]) continue;
		if (Yfull.mem || (!ci && need_lowres)) extra = 1;        /* one more pass A: the planes chroma reads */
		for (i = 0; i < 64; i++) { acc |= rawq[i]; q[i] = rawq[i] ? rawq[i] : 1; }   /* reference :2497-2511 */
		if (acc <= 1) iters = 0;
		if (acc >= 0x800) stop = 1;
		if (iters + extra == 0) continue;

		if (stop || !plane_new(&P, wb * 8, hb * 8)) {    /* dequantise only (reference :2551-2566) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
			for (by = 0; by < hb; by++) {
				int bx, j;
				for (bx = 0; bx < wb; bx++) {
					int16_t *c = block_at(&view, ci, by, bx);
					for (j = 0; j < 64; j++) c[j] = (int16_t)(c[j] * rawq[j]);
				}
			}
			continue;
		}

		for (it = 0; it < iters + extra; it++) {
			int bad = 0;
			/* pass A: (dequantise + range check,) pixels of every block (reference :2589-2609) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(|:bad)
#endif
			for (by = 0; by < hb; by++) {
				int bx, j;
				for (bx = 0; bx < wb; bx++) {
					int16_t *c = block_at(&view, ci, by, bx);
					if (!it) {
						int seen = 0;
						for (j = 0; j < 64; j++) {
							const int v = c[j] * rawq[j];
							c[j] = (int16_t)v; seen |= v + 0x800;
						}
						if (seen >> 12) bad = 1;
					}
					idct_block(c, at(&P, bx * 8, by * 8), P.pitch);
				}
			}
			if (bad) { stop = 1; break; }                /* reference :2610 */
			plane_extend(&P, P.w, P.h);                  /* reference :2612-2620 */
			if (it == iters) break;                      /* the refresh-only pass */

			/* pass B (reference :2627-2640) */
			{
				const uint8_t *lowres = (Llow.mem && (flags & F_JOINT_YUV)) ? at(&Llow, 0, 0) : NULL;
				const size_t lowres_pitch = Llow.pitch;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
				for (by = 0; by < hb; by++) {
					int bx, l;
					if (lowres)
						for (bx = 0; bx < wb; bx++)
							joint_block(block_at(&view, ci, by, bx), q, at(&P, bx * 8, by * 8), P.pitch,
									lowres + (size_t)by * 8 * lowres_pitch + bx * 8, lowres_pitch);
					if (flags & F_LOW_QUALITY) {
						for (bx = 0; bx < wb; bx++) {
							int16_t *c = block_at(&view, ci, by, bx);
							if (!lowres) lowq_block(c, q, at(&P, bx * 8, by * 8), P.pitch);
							if (rebalance) rebalance_block(c, q);
						}
						continue;
					}
					for (bx = 0; bx < wb; bx += QS_NL) {
						int16_t *cf[QS_NL];
						const uint8_t *px[QS_NL];
						const int nvalid = wb - bx < QS_NL ? wb - bx : QS_NL;
						for (l = 0; l < QS_NL; l++) {
							const int b = bx + (l < nvalid ? l : nvalid - 1);
							cf[l] = block_at(&view, ci, by, b);
							px[l] = at(&P, b * 8, by * 8);
						}
						recover_lanes(tt, q, cf, px, P.pitch, rebalance, nvalid);
					}
				}
			}
			if (progress) {                              /* reference :2656-2664 */
				int cur = prog_cur += hb * prog_inc;
				if (cur >= prog_thr) {
					cur = (int)((int64_t)progprec * cur / prog_max);
					prog_thr = (int)(((int64_t)(cur + 1) * prog_max + progprec - 1) / progprec);
					stop = progress(userdata, cur, progprec);
				}
				if (stop) break;
			}
		}

		/* +-1023 (reference :2668-2689) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
		for (by = 0; by < hb; by++) {
			int bx, j;
			for (bx = 0; bx < wb; bx++) {
				int16_t *c = block_at(&view, ci, by, bx);
				for (j = 0; j < 64; j++) c[j] = (int16_t)(c[j] > 1023 ? 1023 : c[j] < -1023 ? -1023 : c[j]);
			}
		}

		if (!stop && Yfull.mem) {                        /* UPSAMPLE_UV (reference :2691-2752) */
			const size_t n = (size_t)job->wblk[0] * job->hblk[0] * 64;
			up[ci - 1] = (int16_t*)malloc(n * sizeof(int16_t));
			if (up[ci - 1] && !upsample_chroma(&P, &Llow, &Yfull, up[ci - 1], job->wblk[0], job->hblk[0],
					job->image_width, job->image_height, job->hsamp[0], job->vsamp[0])) {
				free(up[ci - 1]); up[ci - 1] = NULL;
			}
		} else if (!stop && !ci && need_lowres) {        /* keep luma for the chroma passes (reference :2753-2815) */
			const int ws = job->hsamp[0], hs = job->vsamp[0];
			if (ws == 1 && hs == 1) {
				Llow = P; llow_is_luma_plane = 1; keep_plane = 1;
			} else if (plane_new(&Llow, job->wblk[1] * 8, job->hblk[1] * 8)) {
				downsample_luma(&P, &Llow, ws, hs);
				if (flags & F_UPSAMPLE_UV) { Yfull = P; keep_plane = 1; }
			}
		}
		if (!keep_plane) plane_drop(&P);
	}

#ifdef _OPENMP
	if (old_threads > 0) omp_set_num_threads(old_threads);
#endif
	free(tt);
	(void)llow_is_luma_plane;
	plane_drop(&Llow);
	if (Yfull.mem) {
		plane_drop(&Yfull);
		if (!stop && up[0] && up[1]) {                   /* reference :2836-2849 */
			job->coef_up[0] = up[0]; job->coef_up[1] = up[1]; up[0] = up[1] = NULL;
			job->up_wblk = job->wblk[0]; job->up_hblk = job->hblk[0];
			job->out_hsamp0 = job->out_vsamp0 = 1;
		}
	}
	free(up[0]); free(up[1]);
	for (ci = 0; ci < job->ncomp; ci++)                  /* reference :2851-2859 */
		if (job->has_quant[ci]) for (i = 0; i < 64; i++) job->quant[ci][i] = 1;
	return stop;
}
