/* oracle/dropin/quantsmooth.h -- TEST INFRASTRUCTURE.
 * What a maintainer of the reference does to put its own front-ends (quantsmooth.c, example.c -- both
 * `#include "quantsmooth.h"`, the header-only implementation) on top of the LIBRARY instead: this two-line header
 * takes the implementation's place on the include path and forwards to the library API (include/libjpegqs.h, the
 * same names as reference libjpegqs.h:14-56; example.c:36 says "use libjpegqs.h for linking with library").
 * oracle/Makefile target `dropin` compiles the reference's UNMODIFIED sources from where they lie against it. */
#include <stdint.h>
#include "libjpegqs.h"
