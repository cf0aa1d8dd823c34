/*
 * decode_demo.c -- TEST INFRASTRUCTURE: decode a JPEG to raw pixels through the
 * decode-mode API (jpegqs_start_decompress / jpeg_read_scanlines /
 * jpegqs_finish_decompress, reference libjpegqs.h:50-56, usage as in the
 * reference's example.c:96,123) and write "P5/P6"-less raw samples to stdout.
 *
 *   -DUSE_REFERENCE : compile the reference itself in (oracle/_ref/decode_ref_none)
 *   otherwise       : link against the product's libjpegqs.so (tests/_build/decode_hip)
 *
 * usage: decode_demo <quality 0..6> <niter> in.jpg > out.raw
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "jpeglib.h"

#ifdef USE_REFERENCE
#define JPEGQS_ATTR static
#include "quantsmooth.h"   /* the reference (-I/root/reference), NO_SIMD build */
#else
#include "libjpegqs.h"     /* include/libjpegqs.h of this repository */
#endif

int main(int argc, char **argv) {
	struct jpeg_decompress_struct ci;
	struct jpeg_error_mgr err;
	jpegqs_control_t opts;
	FILE *in;
	int q, flags = 0;
	JSAMPROW row;

	if (argc != 4 || !(in = fopen(argv[3], "rb"))) { fprintf(stderr, "usage: decode_demo q niter in.jpg\n"); return 1; }
	q = atoi(argv[1]);
	if (q < 3) { flags |= JPEGQS_LOW_QUALITY; q += 4; }
	if (q >= 4) flags |= JPEGQS_DIAGONALS;
	if (q >= 5) flags |= JPEGQS_JOINT_YUV;
	if (q >= 6) flags |= JPEGQS_UPSAMPLE_UV;
	memset(&opts, 0, sizeof(opts));
	opts.flags = flags; opts.niter = atoi(argv[2]); opts.threads = 1;

	ci.err = jpeg_std_error(&err);
	jpeg_create_decompress(&ci);
	jpeg_stdio_src(&ci, in);
	jpeg_read_header(&ci, TRUE);
	ci.dct_method = JDCT_ISLOW;
	jpegqs_start_decompress(&ci, &opts);
#ifndef USE_REFERENCE
	if (jpegqs_hip_backend_status() < 0 || err.num_warnings) {   /* the GPU back end failed: say so, deliver nothing */
		fprintf(stderr, "decode_demo: back end failed (status %d, %ld libjpeg warning(s))\n",
				jpegqs_hip_backend_status(), err.num_warnings);
		jpeg_destroy_decompress(&ci);
		return 3;
	}
#endif
	row = (JSAMPROW)malloc((size_t)ci.output_width * ci.output_components);
	fprintf(stderr, "%ux%ux%d\n", ci.output_width, ci.output_height, ci.output_components);
	while (ci.output_scanline < ci.output_height) {
		jpeg_read_scanlines(&ci, &row, 1);
		fwrite(row, 1, (size_t)ci.output_width * ci.output_components, stdout);
	}
	jpegqs_finish_decompress(&ci);
	jpeg_destroy_decompress(&ci);
	free(row); fclose(in);
	return 0;
}
