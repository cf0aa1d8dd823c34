// qs_devfn.h -- small device functions shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qs_device.h"

// float -> int32 the way x86-64 cvttss2si does it (the reference's
// `int range = roundf(a2)`): NaN / out of range => INT_MIN.
__device__ __forceinline__ int f2i_x86(float v) {
  return (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000;
}

// C99 roundf: nearest, ties away from zero (exact: x - trunc(x) is exact).
__device__ __forceinline__ float round_half_away(float x) {
  float t = __builtin_truncf(x);
  float fr = __builtin_fabsf(x - t);
  return fr >= 0.5f ? t + __builtin_copysignf(1.0f, x) : t;
}

// nearest multiple of the quantiser (ties away from zero) via the reference's
// reciprocal tables, and the interval that quantises to it.
// reference quantsmooth.h:332-336, 1552-1557.
__device__ __forceinline__ void interval(int c, int div, int x1, int sh, int& orig, int& lo, int& hi) {
  int a = ((x1 * c) >> 16) + c;
  a = (int)(((uint32_t)a << sh) + 0x4000u) >> 15;   // == (-a * x2 + 0x4000) >> 15 with x2 = -(1 << sh), see QsConsts::x2
  a *= div;
  const int d0 = (div - 1) >> 1, d1 = div >> 1;
  orig = a;
  hi = a + (a < 0 ? d1 : d0);
  lo = a - (a > 0 ? d1 : d0);
}


// index of the plane that owns 64-block group `w` of a plane-set launch
// (binary search over the prefix array; everything here is wave-uniform)
__device__ __forceinline__ int qs_set_find(const QsPlaneSet& set, int w) {
  int lo = 0, hi = set.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (set.wave0[mid] <= w) lo = mid; else hi = mid - 1;
  }
  return lo;
}
