/*
 * jpeg_coefs.c -- MEASUREMENT INPUT HELPER (not part of the product): read a JPEG file with libjpeg
 * (jpeg_read_coefficients, exactly what the jpegqs CLI hands to do_quantsmooth, reference quantsmooth.c:549-550)
 * and dump the quantised DCT coefficient arrays and quantisation tables as one flat binary file, so that bench.py
 * can time the GPU path on libjpeg-encoded input without a Python libjpeg binding.
 *
 *   jpeg_coefs in.jpg out.bin
 *   out.bin: int32 magic 0x51534a43, ncomp, image_width, image_height, colorspace;
 *            per component: int32 wblk, hblk, hsamp, vsamp, has_quant; uint16 quant[64];
 *            then per component hblk * wblk blocks of 64 int16 (natural order, row-major)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "jpeglib.h"

int main(int argc, char **argv) {
	struct jpeg_decompress_struct ci;
	struct jpeg_error_mgr err;
	jvirt_barray_ptr *coefs;
	FILE *in, *out;
	int32_t hdr[5];
	int c;
	if (argc != 3 || !(in = fopen(argv[1], "rb")) || !(out = fopen(argv[2], "wb"))) {
		fprintf(stderr, "usage: jpeg_coefs in.jpg out.bin\n");
		return 1;
	}
	ci.err = jpeg_std_error(&err);
	jpeg_create_decompress(&ci);
	jpeg_stdio_src(&ci, in);
	jpeg_read_header(&ci, TRUE);
	coefs = jpeg_read_coefficients(&ci);
	hdr[0] = 0x51534a43; hdr[1] = ci.num_components; hdr[2] = (int32_t)ci.image_width;
	hdr[3] = (int32_t)ci.image_height; hdr[4] = (int32_t)ci.jpeg_color_space;
	fwrite(hdr, sizeof(hdr), 1, out);
	for (c = 0; c < ci.num_components; c++) {
		jpeg_component_info *comp = ci.comp_info + c;
		int32_t g[5] = { (int32_t)comp->width_in_blocks, (int32_t)comp->height_in_blocks,
				comp->h_samp_factor, comp->v_samp_factor, comp->quant_table != NULL };
		uint16_t q[64];
		int i;
		for (i = 0; i < 64; i++) q[i] = comp->quant_table ? comp->quant_table->quantval[i] : 0;
		fwrite(g, sizeof(g), 1, out);
		fwrite(q, sizeof(q), 1, out);
	}
	for (c = 0; c < ci.num_components; c++) {
		jpeg_component_info *comp = ci.comp_info + c;
		JDIMENSION y;
		for (y = 0; y < comp->height_in_blocks; y++) {
			JBLOCKARRAY row = (*ci.mem->access_virt_barray)((j_common_ptr)&ci, coefs[c], y, 1, FALSE);
			fwrite(row[0], sizeof(JBLOCK), comp->width_in_blocks, out);
		}
	}
	fclose(out);
	jpeg_finish_decompress(&ci);
	jpeg_destroy_decompress(&ci);
	fclose(in);
	return 0;
}
