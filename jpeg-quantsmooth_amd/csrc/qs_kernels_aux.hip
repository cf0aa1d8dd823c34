// qs_kernels_aux.hip -- the cross-component and low-quality stages of the
// recovery path (gfx950).  Together < 5 % of the time of a --quality 5/6 run,
// so these are written for exactness and coalescing, not for the last cycle:
//
//   qs_joint_kernel      JOINT_YUV chroma predictor + fdct_clamp
//                        (reference quantsmooth.h:577-579, 893-921, 343-347, 551-561)
//                        optionally followed by rebalance + final clamp
//                        (the LOW_QUALITY chroma case, reference :936 `goto end`)
//   qs_lowq_kernel       LOW_QUALITY range filter + fdct_clamp + rebalance
//                        (reference :924-938, 1161-1178)
//   qs_downsample_kernel box-downsampled luma at chroma resolution, replicated
//                        out to the chroma plane + apron (reference :2753-2815)
//   qs_upsample_kernel   UPSAMPLE_UV: per low-res pixel luma->chroma linear model,
//                        applied to full-res luma (reference :1851-1864,
//                        2133-2158, 2363-2393) + the edge replication of
//                        :2390-2393 / :2729-2730 (qs_upsample_edges_kernel)
//   qs_fdct_plane_kernel re-encode the upsampled pixels (reference :2735-2750)
//
// All float arithmetic follows the scalar reference operation by operation
// (compile with -ffp-contract=off); integer sums are exact in any order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qs_device.h"
#include "qs_devfn.h"
#include "qs_launch.h"

// --------------------------------------------------------------------------
// float LL&M forward DCT, one 8-point pass (reference idct.h:608-628)
__device__ __forceinline__ void fdct8(float (&v)[8], bool scale) {
  float t0, t1, t2, t3, t4, t5, t6, t7, z1, z2, z3, z4, z5, r0, r1, r2, r3, r4, r5, r6, r7;
  t0 = v[0] + v[7]; t7 = v[0] - v[7];
  t1 = v[1] + v[6]; t6 = v[1] - v[6];
  t2 = v[2] + v[5]; t5 = v[2] - v[5];
  t3 = v[3] + v[4]; t4 = v[3] - v[4];
  z1 = t0 + t3; z4 = t0 - t3; z2 = t1 + t2; z3 = t1 - t2;
  r0 = z1 + z2; r4 = z1 - z2;
  z1 = (z3 + z4) * 0.541196100f;
  r2 = z1 + z4 * 0.765366865f;
  r6 = z1 - z3 * 1.847759065f;
  z1 = t4 + t7; z2 = t5 + t6; z3 = t4 + t6; z4 = t5 + t7;
  z5 = (z3 + z4) * 1.175875602f;
  t4 = t4 * 0.298631336f; t5 = t5 * 2.053119869f;
  t6 = t6 * 3.072711026f; t7 = t7 * 1.501321110f;
  z1 = z1 * 0.899976223f; z2 = z2 * 2.562915447f;
  z3 = z3 * 1.961570560f - z5;
  z4 = z4 * 0.390180644f - z5;
  r7 = t4 - (z1 + z3); r5 = t5 - (z2 + z4);
  r3 = t6 - (z2 + z3); r1 = t7 - (z1 + z4);
  if (scale) {
    r0 *= 0.125f; r1 *= 0.125f; r2 *= 0.125f; r3 *= 0.125f;
    r4 *= 0.125f; r5 *= 0.125f; r6 *= 0.125f; r7 *= 0.125f;
  }
  v[0] = r0; v[1] = r1; v[2] = r2; v[3] = r3; v[4] = r4; v[5] = r5; v[6] = r6; v[7] = r7;
}

// 2-D: columns first, then rows (x 0.125), reference idct.h:895-916
__device__ __forceinline__ void fdct2d(float (&f)[64]) {
#pragma unroll
  for (int x = 0; x < 8; ++x) {
    float c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = f[j * 8 + x];
    fdct8(c, false);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j * 8 + x] = c[j];
  }
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f[y * 8 + j];
    fdct8(r, true);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[y * 8 + j] = r[j];
  }
}

// load / store one block's 64 coefficients (own 128 B, 8 x dwordx4 per lane)
__device__ __forceinline__ void load_block(const int16_t* coef, size_t blk, int (&c)[64]) {
  const uint4* p = reinterpret_cast<const uint4*>(coef) + blk * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint4 v = p[j];
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c[j * 8 + k * 2] = (int16_t)(d[k] & 0xffff);
      c[j * 8 + k * 2 + 1] = (int32_t)d[k] >> 16;
    }
  }
}
__device__ __forceinline__ void store_block(int16_t* coef, size_t blk, const int (&c)[64]) {
  uint4* p = reinterpret_cast<uint4*>(coef) + blk * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    uint32_t d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      d[k] = ((uint32_t)c[j * 8 + k * 2] & 0xffffu) | ((uint32_t)c[j * 8 + k * 2 + 1] << 16);
    p[j] = make_uint4(d[0], d[1], d[2], d[3]);
  }
}

// FDCT + round + clamp into each coefficient's interval (reference :551-561)
// (QS_OPAQUE_ROW: the quantiser scalars are indexed through an opaque per-row offset, so that their scalar loads stay
//  next to their use, 24 at a time; with compile-time indices hipcc hoists all 192 of a block to the top of the kernel
//  and spills them into VGPR lanes / AGPRs -- 385 spilled SGPRs and one wave per SIMD in the kernels below)
//  TOK: a value the previous row produced -- the row offset is "computed" only after it, which keeps the scheduler
//  from lining all eight offsets, and behind them all 192 invariant loads, up at the top anyway)
#define QS_OPAQUE_ROW(N0, G, TOK) int N0 = (G) * 8; asm volatile("" : "+s"(N0), "+v"(TOK))
template <class CP>
__device__ __forceinline__ void fdct_clamp(float (&f)[64], int (&c)[64], CP cst) {
  fdct2d(f);
  int tok = 0;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    QS_OPAQUE_ROW(n0, g, tok);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int orig, lo, hi;
      interval(c[g * 8 + k], cst->qn[n0 + k], cst->x1n[n0 + k], cst->x2n[n0 + k], orig, lo, hi);
      int v = f2i_x86(round_half_away(f[g * 8 + k]));
      c[g * 8 + k] = min(max(v, lo), hi);
    }
    tok = c[g * 8 + 7];
  }
}

// fdct_clamp for a block whose coefficients are still in memory: they stream through eight at a time (one 16-byte
// line: load, clamp the eight FDCT values into their intervals, store), so the 64 coefficients never occupy registers
// next to the 64 floats -- the predictor kernels then fit 128 VGPRs (four waves per SIMD instead of two)
template <class CP>
__device__ __forceinline__ void fdct_clamp_stream(float (&f)[64], int16_t* __restrict__ coef, size_t blk, CP cst) {
  fdct2d(f);
  uint32_t tok = 0;
  uint4* p = reinterpret_cast<uint4*>(coef) + blk * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    QS_OPAQUE_ROW(n0, j, tok);
    const uint4 v = p[j];
    uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int c2[2] = {(int16_t)(d[k] & 0xffff), (int32_t)d[k] >> 16};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int n = n0 + k * 2 + h;
        int orig, lo, hi;
        interval(c2[h], cst->qn[n], cst->x1n[n], cst->x2n[n], orig, lo, hi);
        const int r = f2i_x86(round_half_away(f[j * 8 + k * 2 + h]));
        c2[h] = min(max(r, lo), hi);
      }
      d[k] = ((uint32_t)c2[0] & 0xffffu) | ((uint32_t)c2[1] << 16);
    }
    p[j] = make_uint4(d[0], d[1], d[2], d[3]);
    tok = d[3];
  }
}

// rebalance on register-resident coefficients (reference :1823-1848)
template <class CP>
__device__ __forceinline__ void rebalance_regs(int (&c)[64], CP cst) {
  long long m0 = 0, m1 = 0;
  int tok = 0;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    QS_OPAQUE_ROW(n0, g, tok);
#pragma unroll
    for (int k = (g == 0 ? 1 : 0); k < 8; ++k) {
      int orig, lo, hi;
      interval(c[g * 8 + k], cst->qn[n0 + k], cst->x1n[n0 + k], cst->x2n[n0 + k], orig, lo, hi);
      m0 += (long long)(c[g * 8 + k] * orig);
      m1 += (long long)(orig * orig);
    }
    tok = (int)m1;
  }
  if (m1 > m0) {
    const int mul = (int)(((m1 << 13) + (m0 >> 1)) / m0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      QS_OPAQUE_ROW(n0, g, tok);
#pragma unroll
      for (int k = (g == 0 ? 1 : 0); k < 8; ++k) {
        int orig, lo, hi;
        interval(c[g * 8 + k], cst->qn[n0 + k], cst->x1n[n0 + k], cst->x2n[n0 + k], orig, lo, hi);
        const int v = (c[g * 8 + k] * mul + 0x1000) >> 13;
        c[g * 8 + k] = min(max(v, lo), hi);
      }
      tok = c[g * 8 + 7];
    }
  }
}

// pixels x-1 .. x+8 of one plane row (x0 = byte offset of pixel x in the row,
// a multiple of 8): three aligned loads
__device__ __forceinline__ void load_row10(const uint8_t* row, int (&p)[10]) {
  const uint32_t a = *reinterpret_cast<const uint32_t*>(row - 4);
  const uint2 b = *reinterpret_cast<const uint2*>(row);
  const uint32_t c = *reinterpret_cast<const uint32_t*>(row + 8);
  p[0] = a >> 24;
#pragma unroll
  for (int k = 0; k < 4; ++k) { p[1 + k] = (b.x >> (8 * k)) & 0xff; p[5 + k] = (b.y >> (8 * k)) & 0xff; }
  p[9] = c & 0xff;
}

// weighted 3x3 regression slope of B on A, weights 4/2/1 (reference :894-913)
__device__ __forceinline__ float regress(const int (&a0)[10], const int (&a1)[10], const int (&a2)[10],
                                         const int (&b0)[10], const int (&b1)[10], const int (&b2)[10],
                                         int x, int& sA, int& sB) {
  int sa = 0, sb = 0, saa = 0, sab = 0;
#define QS_TAP(AR, BR, DX, W) { const int a_ = AR[x + 1 + (DX)], b_ = BR[x + 1 + (DX)]; \
    sa += (W) * a_; sb += (W) * b_; saa += (W) * a_ * a_; sab += (W) * a_ * b_; }
  QS_TAP(a0, b0, -1, 1) QS_TAP(a0, b0, 0, 2) QS_TAP(a0, b0, 1, 1)
  QS_TAP(a1, b1, -1, 2) QS_TAP(a1, b1, 0, 4) QS_TAP(a1, b1, 1, 2)
  QS_TAP(a2, b2, -1, 1) QS_TAP(a2, b2, 0, 2) QS_TAP(a2, b2, 1, 1)
#undef QS_TAP
  saa = saa * 16 - sa * sa;
  sab = sab * 16 - sa * sb;
  float scale = (float)saa;
  if (saa) scale = (float)sab / scale;
  scale = scale < -16.0f ? -16.0f : scale;
  scale = scale > 16.0f ? 16.0f : scale;
  sA = sa; sB = sb;
  return scale;
}

// --------------------------------------------------------------------------
// JOINT_YUV predictor: one chroma block per lane.  planeC = this component's
// plane (pass A of this iteration), planeL = low-res luma; same geometry.
template <class CP>
__device__ __forceinline__ void joint_block(CP cst, int16_t* __restrict__ coef,
                                            const uint8_t* __restrict__ planeC, const uint8_t* __restrict__ planeL,
                                            int wblk, int pitch, int blk, int do_rebalance, int final_clamp) {
  const int by = blk / wblk, bx = blk - by * wblk;
  const size_t org = (size_t)(by * 8 + 1) * pitch + QS_APRON_X + bx * 8;

  float f[64];
  int a0[10], a1[10], a2[10], b0[10], b1[10], b2[10];
  load_row10(planeL + org - pitch, a0); load_row10(planeC + org - pitch, b0);
  load_row10(planeL + org, a1);         load_row10(planeC + org, b1);
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    // one pixel row at a time: left to itself hipcc issues the loads of all ten rows of both planes up front and
    // keeps them (and their unpacked bytes) live -- 384 registers, one wave per SIMD, 105 of them spilled at a
    // 256-register budget.  The barrier keeps a row's loads behind the previous row's arithmetic.
    asm volatile("" ::: "memory");
    load_row10(planeL + org + (size_t)(y + 1) * pitch, a2);
    load_row10(planeC + org + (size_t)(y + 1) * pitch, b2);
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      int sA, sB;
      const float scale = regress(a0, a1, a2, b0, b1, b2, x, sA, sB);
      float a = ((float)(a1[x + 1] * 16 - sA) * scale + (float)sB) * 0.0625f;
      a = (a < 0 ? 0 : a) - 128.0f;
      f[y * 8 + x] = a > 128.0f ? 128.0f : a;
      asm volatile("" : "+v"(f[y * 8 + x]));               // one pixel after the other (the 64 predictions are independent:
                                                           //  interleaved, their temporaries cost hundreds of registers)
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) { a0[k] = a1[k]; a1[k] = a2[k]; b0[k] = b1[k]; b1[k] = b2[k]; }
  }
  if (!do_rebalance && !final_clamp) {                       // (wave-uniform) the JOINT_YUV step in front of the recovery kernel
    fdct_clamp_stream(f, coef, blk, cst);
    return;
  }
  int c[64];                                                 // LOW_QUALITY chroma: the block ends here (reference :936)
  load_block(coef, blk, c);
  fdct_clamp(f, c, cst);
  if (do_rebalance) rebalance_regs(c, cst);
  if (final_clamp) {
#pragma unroll
    for (int n = 0; n < 64; ++n) c[n] = min(max(c[n], -1023), 1023);
  }
  store_block(coef, blk, c);
}

__global__ void __launch_bounds__(256)
qs_joint_kernel(const QsConsts* __restrict__ cst, int16_t* __restrict__ coef,
                const uint8_t* __restrict__ planeC, const uint8_t* __restrict__ planeL,
                int wblk, int hblk, int pitch, int do_rebalance, int final_clamp) {
  const int blk = blockIdx.x * 256 + threadIdx.x;
  if (blk >= wblk * hblk) return;
  joint_block(cst, coef, planeC, planeL, wblk, pitch, blk, do_rebalance, final_clamp);
}

// the same over a set of chroma planes (the coupled jobs of a batch, qs_job.cpp: run_coupled): one
// launch instead of one per plane -- a full-HD chroma plane is 128 waves, alone it runs at the
// latency of a single wave
__global__ void __launch_bounds__(256)
qs_joint_set_kernel(const QsPlaneSet set, const QsPlaneAux lowres, int do_rebalance, int final_clamp) {
  const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (w >= set.wave0[set.n]) return;
  const int i = qs_set_find(set, w);
  const QsPlaneRef& r = set.ref[i];
  const int blk = (w - set.wave0[i]) * 64 + (threadIdx.x & 63);
  if (blk >= r.wblk * r.hblk) return;
  // The constants pointer is wave-uniform (every lane of a wave works on the same plane) but comes out of a dynamically
  // indexed kernarg array: to hipcc it is a generic pointer of unknown uniformity, and the 192 quantiser values of
  // fdct_clamp arrive through 186 per-lane flat_load_dwordx4 into as many VGPRs (505 registers, one wave per SIMD).
  // readfirstlane states the uniformity and the constant address space the invariance: scalar loads again.
  typedef const QsConsts __attribute__((address_space(4)))* QsConstsK;
  const uint64_t cu = reinterpret_cast<uint64_t>(r.cst);
  QsConstsK cst = (QsConstsK)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(cu >> 32)) << 32) |
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cu));
  joint_block(cst, r.coef, r.plane, lowres.p[i], r.wblk, r.pitch, blk,
              do_rebalance && (r.mode & QS_PLANE_REBALANCE), final_clamp);
}

// --------------------------------------------------------------------------
// LOW_QUALITY: one block per lane (reference :924-938, 1161-1178)
__global__ void __launch_bounds__(256)
qs_lowq_kernel(const QsConsts* __restrict__ cst, int16_t* __restrict__ coef,
               const uint8_t* __restrict__ plane, int wblk, int hblk, int pitch,
               int do_rebalance, int final_clamp, float c1) {
  const int nblk = wblk * hblk;
  const int blk = blockIdx.x * 256 + threadIdx.x;
  if (blk >= nblk) return;
  const int by = blk / wblk, bx = blk - by * wblk;
  const size_t org = (size_t)(by * 8 + 1) * pitch + QS_APRON_X + bx * 8;

  int c[64];
  load_block(coef, blk, c);
  float range = 0.0f;
  {
    int sum = 0;
#pragma unroll
    for (int n = 1; n < 64; ++n) {
      const int a = c[n] < 0 ? -c[n] : c[n];
      range = range + (float)(cst->qn[n] * a);
      sum += a;
    }
    if (sum) range = range * (4.0f / (float)sum);
    if (range > 128.0f) range = 128.0f;
    range = round_half_away(range);
  }
  const float c0 = 2.0f;
  float f[64];
  int r0[10], r1[10], r2[10];
  load_row10(plane + org - pitch, r0);
  load_row10(plane + org, r1);
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    load_row10(plane + org + (size_t)(y + 1) * pitch, r2);
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      int a = r1[x + 1];
      float a0 = 0.0f, an = 0.0f;
#define QS_LQ(P, CW) { const float t0 = (float)(a - (P)); float t = range - __builtin_fabsf(t0); \
        t = t < 0 ? 0 : t; t = t * t; const float aw = (CW) * t; a0 = a0 + t0 * t * aw; an = an + aw * aw; }
      QS_LQ(r0[x], c1) QS_LQ(r0[x + 1], c0) QS_LQ(r0[x + 2], c1)
      QS_LQ(r1[x], c0)                      QS_LQ(r1[x + 2], c0)
      QS_LQ(r2[x], c1) QS_LQ(r2[x + 1], c0) QS_LQ(r2[x + 2], c1)
#undef QS_LQ
      if (an > 0.0f) a = f2i_x86((float)a - a0 / an);  // the reference keeps `a` as an int
      f[y * 8 + x] = (float)(a - 128);
      asm volatile("" : "+v"(f[y * 8 + x]));               // one pixel after the other, as in joint_block
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) { r0[k] = r1[k]; r1[k] = r2[k]; }
  }
  fdct_clamp(f, c, cst);
  if (do_rebalance) rebalance_regs(c, cst);
  if (final_clamp) {
#pragma unroll
    for (int n = 0; n < 64; ++n) c[n] = min(max(c[n], -1023), 1023);
  }
  store_block(coef, blk, c);
}

// --------------------------------------------------------------------------
// low-res luma: one thread per pixel of the padded target (x in -1..wc, y in -1..hc)
__device__ __forceinline__ int plane_px(const uint8_t* plane, int pitch, int x, int y) {
  return plane[(size_t)(y + 1) * pitch + QS_APRON_X + x];
}

__global__ void __launch_bounds__(256)
qs_downsample_kernel(const uint8_t* __restrict__ Y, int yw, int yh, int ypitch,
                     uint8_t* __restrict__ L, int lw, int lh, int lpitch, int ws, int hs) {
  const int tx = blockIdx.x * 256 + threadIdx.x - 1, ty = blockIdx.y - 1;
  if (tx > lw) return;
  const int w1 = (yw + ws - 1) / ws, h1 = (yh + hs - 1) / hs;
  const int x = min(max(tx, 0), w1 - 1), y = min(max(ty, 0), h1 - 1);
  const int bw = min(ws, yw - x * ws), bh = min(hs, yh - y * hs), n = bw * bh;
  int sum = 0;
  for (int yy = 0; yy < bh; ++yy)
    for (int xx = 0; xx < bw; ++xx) sum += plane_px(Y, ypitch, x * ws + xx, y * hs + yy);
  L[(size_t)(ty + 1) * lpitch + QS_APRON_X + tx] = (uint8_t)((sum + n / 2) / n);
}

// --------------------------------------------------------------------------
// UPSAMPLE_UV: one thread per low-res pixel, writes ws x hs output pixels.
// C = low-res chroma plane (after its extra refresh), Lp = low-res luma,
// Yp = full-res luma; out = u8 [hh][st]
__global__ void __launch_bounds__(256)
qs_upsample_kernel(const uint8_t* __restrict__ C, const uint8_t* __restrict__ Lp, int cpitch,
                   const uint8_t* __restrict__ Yp, int ypitch,
                   uint8_t* __restrict__ out, int st, int xend, int h1, int ws, int hs) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= xend || y >= h1) return;
  int sa = 0, sb = 0, saa = 0, sab = 0;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int w = (dx ? 1 : 2) * (dy ? 1 : 2);
      const int a = plane_px(Lp, cpitch, x + dx, y + dy), b = plane_px(C, cpitch, x + dx, y + dy);
      sa += w * a; sb += w * b; saa += w * a * a; sab += w * a * b;
    }
  saa = saa * 16 - sa * sa;
  sab = sab * 16 - sa * sb;
  float scale = (float)saa;
  if (saa) scale = (float)sab / scale;
  scale = scale < -16.0f ? -16.0f : scale;
  scale = scale > 16.0f ? 16.0f : scale;
  const float offset = (float)plane_px(C, cpitch, x, y) - (float)plane_px(Lp, cpitch, x, y) * scale + 0.5f;
  for (int yy = 0; yy < hs; ++yy)
    for (int xx = 0; xx < ws; ++xx) {
      int v = f2i_x86((float)plane_px(Yp, ypitch, x * ws + xx, y * hs + yy) * scale + offset);
      out[(size_t)(y * hs + yy) * st + x * ws + xx] = (uint8_t)min(max(v, 0), 255);
    }
}

// edge replication after the upsample (stream-ordered behind it):
//  phase 0: rows of the FIRST low-res 8-row strip only: x in [w1*ws, ww) <- x = w1*ws-1
//           (the reference's loop is empty for later strips, :1860-1861, 2390-2393)
//  phase 1: rows [h1*hs, hh) <- row h1*hs-1 (reference :2729-2730)
__global__ void __launch_bounds__(256)
qs_upsample_edges_kernel(uint8_t* __restrict__ out, int st, int ww, int hh, int w1, int h1, int ws, int hs, int phase) {
  const int x = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (phase == 0) {
    const int rows = h1 * hs;          // h1 carries the number of first-strip rows here
    if (r >= rows || x < w1 * ws || x >= ww) return;
    out[(size_t)r * st + x] = out[(size_t)r * st + w1 * ws - 1];
  } else {
    const int row = h1 * hs + r;
    if (row >= hh || x >= st) return;
    out[(size_t)row * st + x] = out[(size_t)(h1 * hs - 1) * st + x];
  }
}

// re-encode: one 8x8 block of `px` per lane -> coefficients (reference :2735-2750)
__global__ void __launch_bounds__(256)
qs_fdct_plane_kernel(const uint8_t* __restrict__ px, int st, int16_t* __restrict__ coef, int wblk, int hblk) {
  const int nblk = wblk * hblk;
  const int blk = blockIdx.x * 256 + threadIdx.x;
  if (blk >= nblk) return;
  const int by = blk / wblk, bx = blk - by * wblk;
  float f[64];
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    const uint2 v = *reinterpret_cast<const uint2*>(px + (size_t)(by * 8 + y) * st + bx * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[y * 8 + k] = (float)((int)((v.x >> (8 * k)) & 0xff) - 128);
      f[y * 8 + 4 + k] = (float)((int)((v.y >> (8 * k)) & 0xff) - 128);
    }
  }
  fdct2d(f);
  int c[64];
#pragma unroll
  for (int n = 0; n < 64; ++n) c[n] = (int16_t)f2i_x86(round_half_away(f[n]));
  store_block(coef, blk, c);
}

// --------------------------------------------------------------------------
// launchers
void qs_launch_joint(const QsConsts* cst, int16_t* coef, const uint8_t* planeC, const uint8_t* planeL,
                     int wblk, int hblk, int do_rebalance, int final_clamp, hipStream_t s) {
  const int nblk = wblk * hblk;
  hipLaunchKernelGGL(qs_joint_kernel, dim3((nblk + 255) / 256), dim3(256), 0, s,
                     cst, coef, planeC, planeL, wblk, hblk, qs_plane_pitch(wblk), do_rebalance, final_clamp);
}

void qs_launch_joint_set(const QsPlaneSet& set, const QsPlaneAux& lowres, int do_rebalance, int final_clamp, hipStream_t s) {
  const int nw = set.wave0[set.n];
  if (nw <= 0) return;
  hipLaunchKernelGGL(qs_joint_set_kernel, dim3((nw + 3) / 4), dim3(256), 0, s, set, lowres, do_rebalance, final_clamp);
}

void qs_launch_lowq(const QsConsts* cst, int16_t* coef, const uint8_t* plane, int wblk, int hblk,
                    int do_rebalance, int final_clamp, float c1, hipStream_t s) {
  const int nblk = wblk * hblk;
  hipLaunchKernelGGL(qs_lowq_kernel, dim3((nblk + 255) / 256), dim3(256), 0, s,
                     cst, coef, plane, wblk, hblk, qs_plane_pitch(wblk), do_rebalance, final_clamp, c1);
}

void qs_launch_downsample(const uint8_t* Y, int ywblk, int yhblk, uint8_t* L, int lwblk, int lhblk,
                          int ws, int hs, hipStream_t s) {
  const int lw = lwblk * 8, lh = lhblk * 8;
  hipLaunchKernelGGL(qs_downsample_kernel, dim3((lw + 2 + 255) / 256, lh + 2), dim3(256), 0, s,
                     Y, ywblk * 8, yhblk * 8, qs_plane_pitch(ywblk), L, lw, lh, qs_plane_pitch(lwblk), ws, hs);
}

void qs_launch_upsample(const uint8_t* C, const uint8_t* L, int cwblk, const uint8_t* Y, int ywblk,
                        uint8_t* out, int st, int ww, int hh, int w1, int h1, int first_rows, int ws, int hs, hipStream_t s) {
  // first_rows: how many leading low-res rows get the right-edge replicate
  // (min(8, h1) for a whole image or the band that holds image row 0, else 0)
  const int xend = (w1 + 7) & ~7;
  if (h1 > 0)
    hipLaunchKernelGGL(qs_upsample_kernel, dim3((xend + 255) / 256, h1), dim3(256), 0, s,
                       C, L, qs_plane_pitch(cwblk), Y, qs_plane_pitch(ywblk), out, st, xend, h1, ws, hs);
  if (w1 * ws < ww && first_rows > 0)
    hipLaunchKernelGGL(qs_upsample_edges_kernel, dim3((ww + 255) / 256, first_rows * hs), dim3(256), 0, s,
                       out, st, ww, hh, w1, first_rows, ws, hs, 0);
  if (h1 * hs < hh && h1 > 0)
    hipLaunchKernelGGL(qs_upsample_edges_kernel, dim3((st + 255) / 256, hh - h1 * hs), dim3(256), 0, s,
                       out, st, ww, hh, w1, h1, ws, hs, 1);
}

void qs_launch_fdct_plane(const uint8_t* px, int st, int16_t* coef, int wblk, int hblk, hipStream_t s) {
  const int nblk = wblk * hblk;
  hipLaunchKernelGGL(qs_fdct_plane_kernel, dim3((nblk + 255) / 256), dim3(256), 0, s, px, st, coef, wblk, hblk);
}
