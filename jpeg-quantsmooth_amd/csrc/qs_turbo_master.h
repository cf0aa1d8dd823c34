/* qs_turbo_master.h -- leading part of libjpeg-turbo's PRIVATE struct jpeg_decomp_master (its jpegint.h), as far
 * as the field the decode-mode tail patches after UPSAMPLE_UV: master->last_MCU_col[1..2] (reference
 * quantsmooth.h:2864-2867; the reference carries the same private declaration at quantsmooth.h:44-60, one
 * layout per libjpeg-turbo version).  Only meaningful when the shim is compiled against libjpeg-turbo headers
 * (LIBJPEG_TURBO_VERSION defined).  tests/test_abi.py compiles this against a stub that dresses libjpeg 9d's
 * header up as libjpeg-turbo and, where /root/reference is mounted, checks every field offset against the
 * reference's declaration with _Static_assert. */
#ifndef QS_TURBO_MASTER_H
#define QS_TURBO_MASTER_H
#include <stddef.h>
#ifdef LIBJPEG_TURBO_VERSION
struct qs_turbo_master {
	void (*prepare_for_output_pass) (j_decompress_ptr);
	void (*finish_output_pass) (j_decompress_ptr);
	boolean is_dummy_pass;
#if LIBJPEG_TURBO_VERSION_NUMBER >= 2001090
	boolean lossless;
#define QS_TURBO_HAS_LOSSLESS 1
#else
#define QS_TURBO_HAS_LOSSLESS 0
#endif
	JDIMENSION first_iMCU_col, last_iMCU_col;
	JDIMENSION first_MCU_col[MAX_COMPONENTS];
	JDIMENSION last_MCU_col[MAX_COMPONENTS];
};
/* no padding surprises: two pointers, the booleans, then JDIMENSIONs back to back */
_Static_assert(offsetof(struct qs_turbo_master, is_dummy_pass) == 2 * sizeof(void (*)(void)), "qs_turbo_master layout");
_Static_assert(offsetof(struct qs_turbo_master, first_iMCU_col) ==
               2 * sizeof(void (*)(void)) + (1 + QS_TURBO_HAS_LOSSLESS) * sizeof(boolean), "qs_turbo_master layout");
_Static_assert(offsetof(struct qs_turbo_master, last_MCU_col) ==
               offsetof(struct qs_turbo_master, first_iMCU_col) + (2 + MAX_COMPONENTS) * sizeof(JDIMENSION), "qs_turbo_master layout");
#endif
#endif
