/*
 * jpegqs_cli.c -- `jpegqs` command-line transcoder on top of the drop-in
 * library (include/libjpegqs.h): JPEG in -> coefficient recovery on the GPU ->
 * JPEG out.  Same options, option syntax, defaults, marker handling, --verbose
 * banner and exit codes as the reference CLI (reference quantsmooth.c:257-259,
 * 288-393, 405-444, 471-489, 541-596, 626) so that scripts and the reference's
 * GUI front-end can call it unchanged; the implementation (table-driven parser,
 * single code path for file/stdio) is ours.  (The reference's own unmodified
 * quantsmooth.c also builds against the library: oracle/Makefile `dropin`.)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jpeglib.h"
#define TRANSCODE_ONLY
#include "../../include/libjpegqs.h"

typedef struct { char shortname; const char *longname; int has_arg; int *dst; } opt_t;

static void usage(const char *prog) {
	fprintf(stderr,
"JPEG Quant Smooth : " JPEGQS_COPYRIGHT " : " JPEGQS_VERSION "\n"
"Back end: AMD MI355X (gfx950) through HIP; CPU back end when no HIP device is visible\n"
"Uses libjpeg, run with \"--verbose 1\" to show its version and copyright\n"
"\n"
"Usage:\n"
"  %s [options] input.jpg output.jpg\n"
"\n"
"Options:\n"
"  -q, --quality n   Quality setting (1-6, default is 3)\n"
"  -n, --niter n     Number of iterations (default is 3)\n"
"  -t, --threads n   Set the number of CPU threads to use (CPU back end only)\n"
"  -o, --optimize    Option for libjpeg to produce smaller output file\n"
"  -v, --verbose n   Print libjpeg debug output\n"
"  -i, --info n      Print quantsmooth debug output (default is 15)\n"
"                      Use the sum of flags: 0 - silent,\n"
"                      1/2/4 - various information,\n"
"                      8 - processing time, 16 - back end.\n"
"  -p, --cpu n       Accepted for compatibility (selects a CPU ISA in the reference)\n"
"\n", prog);
}

static int is_number(const char *s) { return s && s[0] >= '0' && s[0] <= '9'; }

int main(int argc, char **argv) {
	int optimize = 0, verbose = 0, info = 15, cpu = 0, copy = 2;
	int quality = 3, niter = -1, flags_override = -1, threads = 0;
	const opt_t table[] = {
		{ 'o', "--optimize", 0, &optimize }, { 'v', "--verbose", 1, &verbose },
		{ 'i', "--info", 1, &info },         { 'n', "--niter", 1, &niter },
		{ 'q', "--quality", 1, &quality },   { 't', "--threads", 1, &threads },
		{ 'f', "--flags", 1, &flags_override }, { 'p', "--cpu", 1, &cpu },
		{ 'c', "--copy", 1, &copy },
	};
	const int ntable = (int)(sizeof(table) / sizeof(table[0]));
	const char *prog = argv[0];
	int argi = 1, i;
	struct jpeg_decompress_struct src;
	struct jpeg_compress_struct dst;
	struct jpeg_error_mgr src_err, dst_err;
	jvirt_barray_ptr *coefs;
	jpegqs_control_t opts;
	FILE *in = stdin, *out = stdout;

	/* options: "-q 3", "-q3", "--quality 3"; "--" ends them; anything that does
	 * not parse leaves the remaining words as positionals (then the count check
	 * below prints the usage, as the reference does) */
	while (argi < argc) {
		const char *a = argv[argi], *val = NULL;
		const opt_t *o = NULL;
		int consumed = 1;
		if (a[0] != '-' || !a[1]) break;
		if (!strcmp(a, "--")) { argi++; break; }
		for (i = 0; i < ntable; i++) {
			if (a[1] != '-' && a[1] == table[i].shortname) {
				o = &table[i];
				if (a[2]) { if (!o->has_arg) o = NULL; else val = a + 2; }
				break;
			}
			if (a[1] == '-' && !strcmp(a, table[i].longname)) { o = &table[i]; break; }
		}
		if (!o) break;
		if (o->has_arg) {
			if (!val) { val = argi + 1 < argc ? argv[argi + 1] : NULL; consumed = 2; }
			if (!is_number(val)) break;
			*o->dst = atoi(val);
		} else {
			*o->dst = 1;
		}
		argi += consumed;
	}
	src.err = jpeg_std_error(&src_err);
	if (verbose) {
		/* reference quantsmooth.c:405-444: which libjpeg this is -- the compile-time number, then the version and
		 * copyright strings of the library actually loaded, found in its message table (the version string sits
		 * next to the copyright and starts with a digit); the level handed to libjpeg is one less */
		const char *msg = NULL, *ver = NULL;
		int n = src_err.last_jpeg_message;
#ifdef LIBJPEG_TURBO_VERSION
#define QS_STR2(x) #x
#define QS_STR(x) QS_STR2(x)
		fprintf(stderr, "Compiled with libjpeg-turbo version %s\n", QS_STR(LIBJPEG_TURBO_VERSION));
#else
		fprintf(stderr, "Compiled with libjpeg version %d\n", JPEG_LIB_VERSION);
#endif
		for (i = 0; i < n; i++) {
			msg = src_err.jpeg_message_table[i];
			if (msg && !memcmp(msg, "Copyright", 9)) break;
		}
		if (i < n) {
			if (i + 1 < n) ver = src_err.jpeg_message_table[i + 1];
			if (ver && (ver[0] < '0' || ver[0] > '9')) ver = NULL;
			fprintf(stderr, "Version string: %s\n%s\n\n", ver ? ver : "not found", msg);
		} else {
			fprintf(stderr, "Copyright not found\n\n");
		}
		verbose--;
		if (argc - argi == 0) return 1;              /* "jpegqs --verbose 1" alone: the banner was the point */
	}
	if (argc - argi != 2) { usage(prog); return 1; }

	memset(&opts, 0, sizeof(opts));
	{	/* quality -> algorithm flags, reference quantsmooth.c:380-393 */
		int q = quality, fl = 0;
		if (q < 3) { fl |= JPEGQS_LOW_QUALITY; q += 4; }
		if (q >= 4) fl |= JPEGQS_DIAGONALS;
		if (q >= 5) fl |= JPEGQS_JOINT_YUV;
		if (q >= 6) fl |= JPEGQS_UPSAMPLE_UV;
		if (flags_override >= 0) fl = flags_override & JPEGQS_FLAGS_MASK;
		if (cpu > JPEGQS_CPU_MASK) cpu = JPEGQS_CPU_MASK;
		opts.flags = fl | JPEGQS_TRANSCODE | (cpu << JPEGQS_CPU_SHIFT) | (info << JPEGQS_INFO_SHIFT);
		opts.niter = niter >= 0 ? niter : 3;
		opts.threads = threads;
	}

	/* the HIP runtime starts up on a background thread while the file is opened and entropy-decoded
	 * (not for --niter 0 without upsampling: that is a plain transcode and needs no device) */
	if (opts.niter > 0 || (opts.flags & JPEGQS_UPSAMPLE_UV)) jpegqs_hip_prewarm(NULL, NULL);

	jpeg_create_decompress(&src);
	dst.err = jpeg_std_error(&dst_err);
	jpeg_create_compress(&dst);
	src_err.trace_level = dst_err.trace_level = verbose;
	src.mem->max_memory_to_use = dst.mem->max_memory_to_use;

	if (strcmp(argv[argi], "-") && !(in = fopen(argv[argi], "rb"))) {
		fprintf(stderr, "%s: can't open input file \"%s\"\n", prog, argv[argi]);
		return 1;
	}
	jpeg_stdio_src(&src, in);

	if (copy > 0) jpeg_save_markers(&src, JPEG_COM, 0xFFFF);
	if (copy > 1) for (i = 0; i < 16; i++) jpeg_save_markers(&src, JPEG_APP0 + i, 0xFFFF);

	(void)jpeg_read_header(&src, TRUE);
	jpegqs_hip_prewarm(&src, &opts);      /* geometry known: the transfer buffers are set up during the decode */
	coefs = jpeg_read_coefficients(&src);
	/* The reference ignores the return value (its do_quantsmooth cannot fail, and a cancelled or
	 * rejected run still leaves a decodable image, quantsmooth.c:550).  The GPU back end can fail
	 * (a machine WITHOUT a GPU is not a failure: the library's CPU back end runs): then nothing was
	 * processed, and writing the input back out would hand scripts an unsmoothed file.  (The
	 * reference's own CLI on this library gets exit code 2 in that case: the library counts a
	 * libjpeg warning.) */
	if (do_quantsmooth(&src, coefs, &opts) && jpegqs_hip_backend_status() < 0) {
		fprintf(stderr, "%s: GPU back end failed (code %d), no output written\n", prog, jpegqs_hip_backend_status());
		jpeg_destroy_compress(&dst);
		jpeg_destroy_decompress(&src);
		if (in != stdin) fclose(in);
		return 3;
	}

	jpeg_copy_critical_parameters(&src, &dst);
	if (optimize) dst.optimize_coding = TRUE;

	/* the output is opened only now, so it may name the input file */
	if (strcmp(argv[argi + 1], "-") && !(out = fopen(argv[argi + 1], "wb"))) {
		fprintf(stderr, "%s: can't open output file \"%s\"\n", prog, argv[argi + 1]);
		return 1;
	}
	jpeg_stdio_dest(&dst, out);
	jpeg_write_coefficients(&dst, coefs);
	{	/* saved markers, minus the ones libjpeg regenerates itself */
		jpeg_saved_marker_ptr m;
		for (m = src.marker_list; m; m = m->next) {
			int jfif = m->marker == JPEG_APP0 && m->data_length >= 5 && !memcmp(m->data, "JFIF", 5);
			int adobe = m->marker == JPEG_APP0 + 14 && m->data_length >= 5 && !memcmp(m->data, "Adobe", 5);
			if ((dst.write_JFIF_header && jfif) || (dst.write_Adobe_marker && adobe)) continue;
			jpeg_write_marker(&dst, m->marker, m->data, m->data_length);
		}
	}
	jpeg_finish_compress(&dst);
	jpeg_destroy_compress(&dst);
	(void)jpeg_finish_decompress(&src);
	jpeg_destroy_decompress(&src);
	if (in != stdin) fclose(in);
	if (out != stdout) fclose(out);
	return src_err.num_warnings + dst_err.num_warnings ? 2 : 0;
}
