/* TEST: our copy of libjpeg-turbo's private master record (csrc/qs_turbo_master.h) against the one the
 * reference declares (quantsmooth.h:44-60), field by field.  Compiled with -fsyntax-only where
 * /root/reference is mounted:  gcc -fsyntax-only -Itests/stubs/turbo -I<libjpeg> -I/root/reference -DNO_SIMD ... */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stddef.h>
#include "jpeglib.h"
#define JPEGQS_ATTR static
#include "quantsmooth.h"          /* the reference: declares struct jpeg_decomp_master */
#include "qs_turbo_master.h"
#define SAME(f) _Static_assert(offsetof(struct qs_turbo_master, f) == offsetof(struct jpeg_decomp_master, f), #f)
SAME(prepare_for_output_pass); SAME(finish_output_pass); SAME(is_dummy_pass);
#if LIBJPEG_TURBO_VERSION_NUMBER >= 2001090
SAME(lossless);
#endif
SAME(first_iMCU_col); SAME(last_iMCU_col); SAME(first_MCU_col); SAME(last_MCU_col);
int main(void) { return 0; }
