/* TEST STUB: libjpeg 9d's header dressed up as libjpeg-turbo, so that the LIBJPEG_TURBO_VERSION branch of
 * csrc/jpegqs_shim.c (decode-mode re-initialisation after UPSAMPLE_UV, reference quantsmooth.h:44-60,
 * 2864-2867) can at least be COMPILED and its private-struct layout checked in an image that has no
 * libjpeg-turbo headers.  -DQS_STUB_TURBO_NUMBER=... selects the version branch. */
#ifndef QS_STUB_TURBO_NUMBER
#define QS_STUB_TURBO_NUMBER 3000090
#endif
#define LIBJPEG_TURBO_VERSION stub
#define LIBJPEG_TURBO_VERSION_NUMBER QS_STUB_TURBO_NUMBER
#include_next "jpeglib.h"
