// qs_kernels.hip -- gfx950 (CDNA4, wave64) kernels for the jpeg-quantsmooth
// coefficient-recovery path.  Written for MI355X only; build with
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// (-ffp-contract=off is part of the numerical contract: the reference's scalar
// path rounds every multiply and add separately, reference quantsmooth.h:1519).
//
// Work decomposition: ONE 8x8 BLOCK PER LANE, 64 blocks per wavefront.
// Why not "one block per wavefront + cross-lane reductions": the reference's
// scalar path accumulates each coefficient's 144/242 float terms strictly in
// sequence (reference quantsmooth.h:1517-1545), and float addition does not
// reassociate -- a DPP/LDS tree over 64 lanes cannot be bit-exact.  With a
// block per lane every lane runs that exact chain privately; all lanes of a
// wave work on the same coefficient index at the same time, so weights,
// quantiser data and control flow are wave-uniform (scalar loads + uniform
// branches) and nothing diverges.
//
// Per-lane state: the block's 64 pixels (scaled by 2^-12, see QS_TERM_D) and the
// 32 edge differences live in VGPRs as exact small floats, so an interior pixel
// difference is one v_sub_f32; the 32 neighbour-edge pixels stay packed four to
// a VGPR; the 64 int16 coefficients live in LDS, one dword column per lane
// (stride 65 dwords => conflict-free both for the per-lane column accesses and
// for the coalesced-load transpose).  128 VGPRs => 4 waves per SIMD.
//
// Build-time switches (all default to the measured-best setting; the others are
// kept as checked-in experiments): QS_SMOOTH_MIN_WAVES,
// QS_PIN_DIFFS, QS_PIN_EDGE, QS_SKIP_ZERO_WEIGHTS, QS_SMEM_PIPELINE,
// QS_IDCT_DOT2, QS_ABLATE_*.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qs_device.h"

#define QS_LDS_PITCH 65 /* dwords per coefficient-pair row, 64 lanes + 1 pad */

// --------------------------------------------------------------------------
// small helpers

__device__ __forceinline__ void wave_lds_sync() {
  // LDS traffic of one wave is processed in order; this only stops the
  // compiler from moving LDS accesses across the point where lanes exchange
  // data through LDS.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t mulc(uint32_t a, int c) {
  // 24-bit multiply: exact low 32 bits whenever |a| < 2^23, which holds for
  // every operand of the two IDCT passes (int16 inputs; pass-2 inputs are an
  // int32 shifted right by 11, sums of at most four of them).
  return (uint32_t)__mul24((int)a, c);
}

// 1-D LL&M inverse DCT butterfly, 13-bit constants, wrapping int32 arithmetic.
// Behaviour of reference idct.h:57-89.
__device__ __forceinline__ void idct8(uint32_t (&v)[8]) {
  uint32_t z1, z2, z3, z4, z5, t0, t1, t2, t3, e0, e1, e2, e3;
  z2 = v[2]; z3 = v[6];
  z1 = mulc(z2 + z3, 4433);
  t2 = z1 - mulc(z3, 15137);
  t3 = z1 + mulc(z2, 6270);
  t0 = (v[0] + v[4]) << 13;
  t1 = (v[0] - v[4]) << 13;
  e0 = t0 + t3; e3 = t0 - t3; e1 = t1 + t2; e2 = t1 - t2;
  t0 = v[7]; t1 = v[5]; t2 = v[3]; t3 = v[1];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; z4 = t1 + t3;
  z5 = mulc(z3 + z4, 9633);
  t0 = mulc(t0, 2446);  t1 = mulc(t1, 16819);
  t2 = mulc(t2, 25172); t3 = mulc(t3, 12299);
  z1 = mulc(z1, 7373);  z2 = mulc(z2, 20995);
  z3 = mulc(z3, 16069); z4 = mulc(z4, 3196);
  z3 = z5 - z3; z4 = z5 - z4;
  t0 += z3 - z1; t1 += z4 - z2; t2 += z3 - z2; t3 += z4 - z1;
  v[0] = e0 + t3; v[7] = e0 - t3;
  v[1] = e1 + t2; v[6] = e1 - t2;
  v[2] = e2 + t1; v[5] = e2 - t1;
  v[3] = e3 + t0; v[4] = e3 - t0;
}

// pass 1 (columns) of the 2-D IDCT on 64 register-resident values; keeps two
// fractional bits (reference idct.h:481-503).
__device__ __forceinline__ void idct_pass1(uint32_t (&ws)[64]) {
#pragma unroll
  for (int x = 0; x < 8; ++x) {
    uint32_t col[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) col[j] = ws[j * 8 + x];
    idct8(col);
#pragma unroll
    for (int j = 0; j < 8; ++j) ws[j * 8 + x] = (uint32_t)((int32_t)(col[j] + 1024u) >> 11);
  }
}

// pass 2 for one row; folds +128 and rounding, clamps to 0..255
// (reference idct.h:509-538).
__device__ __forceinline__ void idct_pass2_row(uint32_t (&row)[8], int (&out)[8]) {
  idct8(row);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // clamp BEFORE the shift (same result as clamping (x >> 18) to 0..255).
    // Toolchain hazard, ROCm 7.2 / gfx950: hipcc turns "shift, clamp to
    // 0..255, pack two bytes" into v_ashr_pk_u8_i32 and then ORs further
    // bytes into bits 31:16 of its result as if they were zero -- they are
    // not on MI355X (observed: corrupted pixels 6/7 of every row).  Clamping
    // first keeps that instruction out; csrc/Makefile greps the ISA for it.
    int z = (int32_t)(row[j] + (257u << 17));
    z = min(max(z, 0), (256 << 18) - 1);
    out[j] = z >> 18;
  }
}

// --------------------------------------------------------------------------
// QS_IDCT_DOT2: column pass of the refresh IDCT on packed int16 pairs.
// In the recovery kernel the 64 coefficients of a block sit in LDS as 32
// dwords.  With this option dword m = 4*x + t of a lane's column holds, for
// block column x, the pair  t=0: (c0,c4)  t=1: (c2,c6)  t=2: (c7,c5)  t=3: (c3,c1)
// (first = low half; cN = coefficient in row N), which is what the LL&M
// butterfly consumes together: every partial sum of the column pass is a
// 2-term dot product with constant int16 weights -> v_dot2c_i32_i16, one
// instruction for two multiplies and two adds (the odd part is expanded into
// its 4x4 integer matrix).  Integer ring arithmetic, so the regrouping is exact
// (no overflow either: |coef| <= 3071 inside the recovery loop).
// Measured on MI355X (A/B in one run, 4096^2): bit-exact, but not faster --
// q3 0.541 vs 0.537 ms, q4 0.818 vs 0.793 ms: v_dot2c_i32_i16 issues at the
// same half rate as v_mul_i32_i24 (tools/ubench_valu.hip) and the accumulator
// v_movs plus the extra register pressure eat the saved adds.  Off by default;
// kept as a checked-in negative result.
#ifndef QS_IDCT_DOT2
#define QS_IDCT_DOT2 0
#endif
typedef short qs_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2(uint32_t pair, int klo, int khi, int acc) {
  const qs_s2 k = {(short)klo, (short)khi};
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(qs_s2, pair), k, acc, false);
}
// row of a coefficient -> (pair slot t, half h) of the layout above
__device__ __forceinline__ constexpr int pair_slot(int r) { return r == 0 || r == 4 ? 0 : r == 2 || r == 6 ? 1 : r == 7 || r == 5 ? 2 : 3; }
__device__ __forceinline__ constexpr int pair_half(int r) { return (r == 4 || r == 6 || r == 5 || r == 1) ? 1 : 0; }

// one column: four packed pairs in, eight workspace values out (descaled by 11
// with the rounding bias folded into the accumulators)
__device__ __forceinline__ void idct_col_dot2(uint32_t p04, uint32_t p26, uint32_t p75, uint32_t p31, uint32_t (&o)[8]) {
  const int t0 = dot2(p04, 8192, 8192, 1024);      // (c0 + c4) << 13, + rounding
  const int t1 = dot2(p04, 8192, -8192, 1024);     // (c0 - c4) << 13, + rounding
  const int t2 = dot2(p26, 4433, -10704, 0);       // z1 - c6 * 15137
  const int t3 = dot2(p26, 10703, 4433, 0);        // z1 + c2 * 6270
  const int e0 = t0 + t3, e3 = t0 - t3, e1 = t1 + t2, e2 = t1 - t2;
  // odd part as a 4x4 integer matrix on (c7, c5, c3, c1)
  const int q0 = dot2(p31, -6436, 2260, dot2(p75, -11363, 9633, 0));
  const int q1 = dot2(p31, -11362, 6437, dot2(p75, 9633, 2261, 0));
  const int q2 = dot2(p31, -2259, 9633, dot2(p75, -6436, -11362, 0));
  const int q3 = dot2(p31, 9633, 11363, dot2(p75, 2260, 6437, 0));
  o[0] = (uint32_t)((e0 + q3) >> 11); o[7] = (uint32_t)((e0 - q3) >> 11);
  o[1] = (uint32_t)((e1 + q2) >> 11); o[6] = (uint32_t)((e1 - q2) >> 11);
  o[2] = (uint32_t)((e2 + q1) >> 11); o[5] = (uint32_t)((e2 - q1) >> 11);
  o[3] = (uint32_t)((e3 + q0) >> 11); o[4] = (uint32_t)((e3 - q0) >> 11);
}

// --------------------------------------------------------------------------
// Kernel A: (dequantise +) IDCT every block into the pixel plane and write the
// clamp-to-edge apron.  One block per lane; consecutive lanes take consecutive
// blocks of a block row so the 8-byte pixel-row stores of a wave coalesce into
// 512-byte segments.  Reference quantsmooth.h:2589-2620 (pass A + borders).
//   first   : iteration 0 -- multiply by the file's quantiser, flag
//             out-of-range products (reference :2597-2603)
//   rep_top / rep_bot : write the y = -1 / y = h apron rows by replication
//             (false for the interior edges of a multi-GPU band, whose apron
//             rows are halo rows received from the neighbouring band)
__global__ void __launch_bounds__(256)
qs_idct_plane_kernel(const QsConsts* __restrict__ cst, int16_t* __restrict__ coef,
                     uint8_t* __restrict__ plane, int wblk, int hblk, int pitch,
                     int first, int rep_top, int rep_bot, int* __restrict__ status) {
  const int nblk = wblk * hblk;
  const int blk = blockIdx.x * 256 + threadIdx.x;
  if (blk >= nblk) return;
  const int by = blk / wblk, bx = blk - by * wblk;

  uint4* cp = reinterpret_cast<uint4*>(coef) + (size_t)blk * 8;
  uint32_t ws[64];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    uint4 v = cp[j];
    uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ws[j * 8 + c * 2] = (uint32_t)(int32_t)(int16_t)(d[c] & 0xffff);
      ws[j * 8 + c * 2 + 1] = (uint32_t)((int32_t)d[c] >> 16);
    }
  }
  if (first) {
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t d[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int lo = (int32_t)ws[j * 8 + c * 2] * cst->qraw[j * 8 + c * 2];
        int hi = (int32_t)ws[j * 8 + c * 2 + 1] * cst->qraw[j * 8 + c * 2 + 1];
        bad |= ((unsigned)(lo + 0x800) > 0xfffu) | ((unsigned)(hi + 0x800) > 0xfffu);
        lo = (int16_t)lo; hi = (int16_t)hi;  // stored as JCOEF (reference :2599)
        ws[j * 8 + c * 2] = (uint32_t)lo; ws[j * 8 + c * 2 + 1] = (uint32_t)hi;
        d[c] = ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
      }
      cp[j] = make_uint4(d[0], d[1], d[2], d[3]);
    }
    if (bad) atomicOr(status, 1);
  }

  idct_pass1(ws);
  uint8_t* org = plane + (size_t)(by * 8 + 1) * pitch + QS_APRON_X + bx * 8;
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    uint32_t row[8]; int o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) row[j] = ws[y * 8 + j];
    idct_pass2_row(row, o);
    uint2 pk;
    pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
    pk.y = (uint32_t)o[4] | ((uint32_t)o[5] << 8) | ((uint32_t)o[6] << 16) | ((uint32_t)o[7] << 24);
    uint8_t* rp = org + (size_t)y * pitch;
    *reinterpret_cast<uint2*>(rp) = pk;
    const bool top = (y == 0 && by == 0 && rep_top), bot = (y == 7 && by == hblk - 1 && rep_bot);
    if (bx == 0) {
      rp[-1] = (uint8_t)o[0];
      if (top) rp[-1 - pitch] = (uint8_t)o[0];
      if (bot) rp[-1 + pitch] = (uint8_t)o[0];
    }
    if (bx == wblk - 1) {
      rp[8] = (uint8_t)o[7];
      if (top) rp[8 - pitch] = (uint8_t)o[7];
      if (bot) rp[8 + pitch] = (uint8_t)o[7];
    }
    if (top) *reinterpret_cast<uint2*>(rp - pitch) = pk;
    if (bot) *reinterpret_cast<uint2*>(rp + pitch) = pk;
  }
}

// --------------------------------------------------------------------------
// Kernel B: the recovery loop.  Reference quantsmooth.h:1396-1565 (main loop),
// :1566-1568 + :1823-1848 (rebalance), :2668-2689 (final clamp, optional).

#include "qs_devfn.h"

// 16-bit views of the dword columns.  may_alias: these accesses overlap the
// 32-bit accesses used for staging and for the IDCT refresh, and the compiler
// must not reorder one kind across the other (it does under strict aliasing).
typedef int16_t __attribute__((may_alias)) lds_i16;
__device__ __forceinline__ int lds_halfword(int i) {   // index of coefficient i in 16-bit units, pitch aside
#if QS_IDCT_DOT2
  const int r = i >> 3, x = i & 7;
  const int t = (0x21203130 >> (4 * r)) & 3;         // rows {0,4}->0 {2,6}->1 {7,5}->2 {3,1}->3
  const int h = (0x72 >> r) & 1;                      // rows 1, 4, 5, 6 are the high half
  return ((x * 4 + t) << 1) | h;
#else
  return i;
#endif
}
__device__ __forceinline__ int lds_coef(const uint32_t* col, int i) {
  const int hw = lds_halfword(i);
  const lds_i16* p = reinterpret_cast<const lds_i16*>(col + (hw >> 1) * QS_LDS_PITCH) + (hw & 1);
  return *p;
}
__device__ __forceinline__ void lds_set_coef(uint32_t* col, int i, int v) {
  const int hw = lds_halfword(i);
  lds_i16* p = reinterpret_cast<lds_i16*>(col + (hw >> 1) * QS_LDS_PITCH) + (hw & 1);
  *p = (lds_i16)v;
}

// One term of the weighted least-squares sums, reference quantsmooth.h:1519-1520:
//     t = max(R - |d|, 0); t *= t; x = d*t; y = w*t; num += x*y; den += y*y
// evaluated in a power-of-two-scaled domain that needs one instruction less:
// pixels are kept as p * 2^-12, so d' = d * 2^-12 and, with R' = R * 2^-12,
//     u' = clamp01(R' - |d'|)   -- ONE v_sub_f32 with the |.| input modifier and
//                                  the clamp output modifier; R = 2q <= 4094 < 2^12
//                                  so the upper clamp never fires and u' = t * 2^-12
//     t' = u'*u' = t^2 * 2^-24 (exact, t^2 < 2^24 is an integer)
//     x' = fl(d'*t') = x * 2^-36,  y' = fl(w*t') = y * 2^-24
//     num' = num * 2^-60, den' = den * 2^-48, num'/den' = (num/den) * 2^-12
// Scaling by powers of two commutes with IEEE rounding as long as nothing
// underflows; the smallest non-zero magnitudes are |x'| >= 2^-36 and
// |y'| >= min|w| * 2^-24 with min|w| ~ 2^-29 (checked when the tables are
// built, qs_host.cpp), so products stay above 2^-110 and sums of such terms are
// exact multiples of 2^-149.  Every rounding therefore happens on the same
// significand as in the reference's unscaled evaluation: bit-exact.
#define QS_PIX_SCALE 0.000244140625f /* 2^-12 */
#define QS_TERM_D(D, W) { \
    float u_ = __builtin_amdgcn_fmed3f(Rs - __builtin_fabsf(D), 0.0f, 1.0f); \
    float t_ = u_ * u_; \
    float x_ = (D) * t_; \
    float y_ = (W) * t_; \
    num = num + x_ * y_; \
    den = den + y_ * y_; }
#define QS_TERM(A, B, W) { float d_ = (A) - (B); QS_TERM_D(d_, W) }

#ifndef QS_PIN_DIFFS
#define QS_PIN_DIFFS 1
#endif
// skip the difference terms whose weight is structurally zero (see QS_TERM_OPT)
#ifndef QS_SKIP_ZERO_WEIGHTS
#define QS_SKIP_ZERO_WEIGHTS 1
#endif
// QS_PIN_EDGE=1 recomputes the 32 edge-pixel conversions at every anti-diagonal
// (no scratch: HBM traffic stays close to the algorithmic 256 B/block); 0 lets
// the compiler hoist them out of the loop, where they end up in 124 B/lane of
// scratch (+130 MB written and re-read per 8192^2 launch).  Measured A/B on
// MI355X: 1 is 1 % faster at 8192^2 and 3 % slower at 4096^2 -- a wash in time,
// so the variant without the spill traffic is the default.
#ifndef QS_PIN_EDGE
#define QS_PIN_EDGE 1
#endif
// Explicit double-buffered scalar weight prefetch (QS_STEP below).  Measured on
// MI355X: it makes 2-3 waves/SIMD as fast as 4, but at 4 waves/SIMD (the
// default) the compiler's own s_load placement is already covered by the other
// waves and is ~3% faster, so it is off by default and kept for the
// lower-occupancy, more-registers variants.
#ifndef QS_SMEM_PIPELINE
#define QS_SMEM_PIPELINE 0
#endif
typedef float qs_w16 __attribute__((ext_vector_type(16)));
// explicit scalar loads: the compiler does not know about them, so the wait is
// tied to the buffer through a "+s" operand (uses cannot move above it)
#define QS_SLOAD16(BUF, BYTEOFF) \
  asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(BUF) : "s"(tabp), "s"((uint32_t)(BYTEOFF)) : "memory")
#define QS_SWAIT(BUF) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(BUF) : : "memory")
// One pipeline step: wait for the chunk in CUR, then immediately issue the load
// of the following chunk into NXT.  The num/den operands pin the step between
// the accumulations of the previous chunk and those of this one -- without
// them the scheduler hoists/sinks the surrounding VALU work across the asm and
// the load ends up right in front of its own wait.
#define QS_STEP(CUR, NXT, BYTEOFF) \
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 %1, %4, %5" \
               : "+s"(CUR), "=&s"(NXT), "+v"(num), "+v"(den) \
               : "s"(tabp), "s"((uint32_t)(BYTEOFF)) : "memory")
#ifndef QS_SMOOTH_MIN_WAVES
#define QS_SMOOTH_MIN_WAVES 4 /* waves per SIMD the register allocator must leave room for */
#endif

__device__ __forceinline__ float byte_f(uint32_t v, int n) { return (float)((v >> (8 * n)) & 0xffu); }

// waves per workgroup: the waves of a workgroup share nothing (each has its own
// LDS slice), so the size only sets the dispatch granularity
#ifndef QS_WAVES_PER_WG
#define QS_WAVES_PER_WG 4
#endif

template <bool DIAG>
__global__ void __launch_bounds__(64 * QS_WAVES_PER_WG, QS_SMOOTH_MIN_WAVES)
qs_smooth_plane_kernel(const QsConsts* __restrict__ cst, int16_t* __restrict__ coef,
                       const uint8_t* __restrict__ plane, int wblk, int hblk, int pitch,
                       int rebalance, int final_clamp, int blk_begin, int blk_end) {
  // blocks [blk_begin, blk_end) of the plane (linear, row-major): the whole
  // plane, or the interior / the edge block rows of a band when the halo
  // exchange is overlapped with the interior (bands.py)
  __shared__ uint32_t lds_all[QS_WAVES_PER_WG][32 * QS_LDS_PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* lds = lds_all[wave];
  uint32_t* col = lds + lane;
  const int nblk = blk_end;
  const int base = blk_begin + (blockIdx.x * QS_WAVES_PER_WG + wave) * 64;
  if (base >= nblk) return;  // wave-uniform
  const int nvec = min(64, nblk - base) * 8;

  // ---- stage the wave's 64 blocks (8 KiB contiguous): 16 B per lane per
  // load, fully coalesced, transposed through LDS into per-lane columns.
  uint4* gsrc = reinterpret_cast<uint4*>(coef) + (size_t)base * 8;
  {
    const int m0 = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = j * 64 + lane;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (idx < nvec) v = gsrc[idx];
      uint32_t* dst = lds + m0 * QS_LDS_PITCH + (j * 8 + (lane >> 3));
      dst[0] = v.x; dst[QS_LDS_PITCH] = v.y; dst[2 * QS_LDS_PITCH] = v.z; dst[3 * QS_LDS_PITCH] = v.w;
    }
  }

  // ---- neighbour edge pixels from the frozen plane (reference :1396-1401),
  // kept packed (4 per VGPR): [0..1] row above, [2..3] row below,
  // [4..5] column to the left, [6..7] column to the right
  const int blk = min(base + lane, nblk - 1);
  const int by = blk / wblk, bx = blk - by * wblk;
  const uint8_t* org = plane + (size_t)(by * 8 + 1) * pitch + QS_APRON_X + bx * 8;
  uint32_t edge[8];
  {
    const uint2 t = *reinterpret_cast<const uint2*>(org - pitch);
    const uint2 b = *reinterpret_cast<const uint2*>(org + (size_t)8 * pitch);
    edge[0] = t.x; edge[1] = t.y; edge[2] = b.x; edge[3] = b.y;
    uint32_t l[8], r[8];
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      l[y] = org[(ptrdiff_t)y * pitch - 1];
      r[y] = org[(ptrdiff_t)y * pitch + 8];
    }
    edge[4] = l[0] | (l[1] << 8) | (l[2] << 16) | (l[3] << 24);
    edge[5] = l[4] | (l[5] << 8) | (l[6] << 16) | (l[7] << 24);
    edge[6] = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24);
    edge[7] = r[4] | (r[5] << 8) | (r[6] << 16) | (r[7] << 24);
  }
  wave_lds_sync();
#if QS_IDCT_DOT2
  {  // row-major pairs (c[r][2j], c[r][2j+1]) -> column pairs of QS_IDCT_DOT2, in place, own column only
    uint32_t rm[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) rm[m] = col[m * QS_LDS_PITCH];
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      constexpr int ra[4] = {0, 2, 7, 3}, rb[4] = {4, 6, 5, 1};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t a = rm[ra[t] * 4 + (x >> 1)], b = rm[rb[t] * 4 + (x >> 1)];
        const uint32_t lo = (x & 1) ? (a >> 16) : (a & 0xffffu);
        const uint32_t hi = (x & 1) ? (b & 0xffff0000u) : (b << 16);
        col[(4 * x + t) * QS_LDS_PITCH] = lo | hi;
      }
    }
  }
#endif

  constexpr int TS = DIAG ? 272 : 160;
  float px[64];   // own pixels * 2^-12
  float bd[32];   // own edge pixel minus neighbour pixel, * 2^-12 (top, bottom, left, right)

  // 14 zigzag anti-diagonals; the block's own pixels are re-derived from its
  // current coefficients at the start of each (reference :313-322, 1407-1409;
  // refreshing unconditionally is identical to refreshing "if stale").
  int kfirst = 63;
#if QS_SMEM_PIPELINE
  const float* tabp = cst->tab;
  qs_w16 WA, WB;
  int i_cur = 63;                        // natural index of zigzag position 63
  QS_SLOAD16(WA, (uint32_t)63 * (TS * 4));   // 63 & 7 != 0: the H section comes first
#endif
#pragma unroll 1
  for (int g = 0; g < 14; ++g) {
#ifdef QS_ABLATE_IDCT
    if (g == 0)
#endif
    {
      uint32_t ws[64];
#if QS_IDCT_DOT2
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        uint32_t o[8];
        idct_col_dot2(col[(4 * x + 0) * QS_LDS_PITCH], col[(4 * x + 1) * QS_LDS_PITCH],
                      col[(4 * x + 2) * QS_LDS_PITCH], col[(4 * x + 3) * QS_LDS_PITCH], o);
#pragma unroll
        for (int j = 0; j < 8; ++j) ws[j * 8 + x] = o[j];
      }
#else
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        const uint32_t d = col[m * QS_LDS_PITCH];
        ws[2 * m] = (uint32_t)(int32_t)(int16_t)(d & 0xffff);
        ws[2 * m + 1] = (uint32_t)((int32_t)d >> 16);
      }
      idct_pass1(ws);
#endif
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        uint32_t row[8]; int o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j] = ws[y * 8 + j];
        idct_pass2_row(row, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) px[y * 8 + j] = (float)o[j] * QS_PIX_SCALE;
      }
      // the 32 edge differences only change here, not per coefficient
      // (see QS_PIN_EDGE above for the optional opaque barrier)
#if QS_PIN_EDGE
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(edge[e]));
#endif
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        bd[x]      = px[x]         - byte_f(edge[0 + (x >> 2)], x & 3) * QS_PIX_SCALE;
        bd[8 + x]  = px[56 + x]    - byte_f(edge[2 + (x >> 2)], x & 3) * QS_PIX_SCALE;
        bd[16 + x] = px[x * 8]     - byte_f(edge[4 + (x >> 2)], x & 3) * QS_PIX_SCALE;
        bd[24 + x] = px[x * 8 + 7] - byte_f(edge[6 + (x >> 2)], x & 3) * QS_PIX_SCALE;
      }
    }
    // anti-diagonal g (walking down from k = 63) holds min(g + 1, 15 - g) coefficients
    const int len = min(g + 1, 15 - g);
    const int klast = max(kfirst - len + 1, 1);
#pragma unroll 1
    for (int k = kfirst; k >= klast; --k) {
#if QS_PIN_DIFFS
      // Keep the interior pixel differences inside the coefficient loop:
      // otherwise the compiler hoists all 112/210 of them out of the k-loop
      // (they only change per anti-diagonal) and pays for it in VGPRs /
      // scratch spills.
#pragma unroll
      for (int p = 0; p < 64; ++p) asm volatile("" : "+v"(px[p]));
#endif
      // quantiser scalars for the update below: fetched here, not inside the
      // divergent branch where their latency would be exposed every time
      int qk = cst->q[k], x1k = cst->x1[k], x2k = cst->x2[k];
      asm volatile("" : "+s"(qk), "+s"(x1k), "+s"(x2k));
#if QS_SMEM_PIPELINE
      // Weights stream through two 16-SGPR buffers (WA/WB): the s_load of the
      // next 16-float chunk is issued right after the wait for the current
      // one, so its latency hides behind ~140 VALU instructions instead of
      // stalling the wave ~20 times per coefficient (the compiler otherwise
      // places every scalar load a few instructions before its first use).
      // Section lengths before any chunk are even (H 4, B 2, V 4 chunks), so a
      // section always starts in WA; the first chunk of the next coefficient
      // is prefetched from the last chunk of this one, across the IDCT refresh
      // and the coefficient update.
      const int i = i_cur;
      const int i_nxt = cst->nat[k > 1 ? k - 1 : 1];
      const float Rs = cst->range[k];   // 2q * 2^-12
      const uint32_t kb = (uint32_t)k * (TS * 4);
      const uint32_t next_first = (uint32_t)(k > 1 ? k - 1 : 1) * (TS * 4) + ((i_nxt & 7) ? 0u : 256u);
      const uint32_t after_V = DIAG ? kb + 640u : next_first;
      const uint32_t after_B = (i > 7) ? kb + 384u : after_V;
      float num = 0.0f, den = 0.0f;

#define QS_ROWS_H(W, Y0) { \
        _Pragma("unroll") for (int yy = 0; yy < 2; ++yy) \
        _Pragma("unroll") for (int x = 0; x < 7; ++x) \
          QS_TERM(px[((Y0) + yy) * 8 + x], px[((Y0) + yy) * 8 + x + 1], W[yy * 8 + x]) }
#define QS_ROWS_V(W, Y0, NY) { \
        _Pragma("unroll") for (int yy = 0; yy < (NY); ++yy) \
        _Pragma("unroll") for (int x = 0; x < 8; ++x) \
          QS_TERM(px[((Y0) + yy) * 8 + x], px[((Y0) + yy) * 8 + x + 8], W[yy * 8 + x]) }
#define QS_ROW_D(W, Y) { \
        _Pragma("unroll") for (int x = 0; x < 7; ++x) { \
          QS_TERM(px[(Y) * 8 + x], px[(Y) * 8 + x + 9], W[x]) \
          QS_TERM(px[(Y) * 8 + x + 1], px[(Y) * 8 + x + 8], W[8 + x]) } }
#define QS_EDGE16(W, J0) { \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) QS_TERM_D(bd[(J0) + j], W[j]) }

      if (i & 7) {
        QS_STEP(WA, WB, kb + 64u);  QS_ROWS_H(WA, 0)
        QS_STEP(WB, WA, kb + 128u); QS_ROWS_H(WB, 2)
        QS_STEP(WA, WB, kb + 192u); QS_ROWS_H(WA, 4)
        QS_STEP(WB, WA, kb + 256u); QS_ROWS_H(WB, 6)
      }
      QS_STEP(WA, WB, kb + 320u);   QS_EDGE16(WA, 0)
      QS_STEP(WB, WA, after_B);     QS_EDGE16(WB, 16)
      if (i > 7) {
        QS_STEP(WA, WB, kb + 448u); QS_ROWS_V(WA, 0, 2)
        QS_STEP(WB, WA, kb + 512u); QS_ROWS_V(WB, 2, 2)
        QS_STEP(WA, WB, kb + 576u); QS_ROWS_V(WA, 4, 2)
        QS_STEP(WB, WA, after_V);   QS_ROWS_V(WB, 6, 1)
      }
      if (DIAG) {
        QS_STEP(WA, WB, kb + 704u);  QS_ROW_D(WA, 0)
        QS_STEP(WB, WA, kb + 768u);  QS_ROW_D(WB, 1)
        QS_STEP(WA, WB, kb + 832u);  QS_ROW_D(WA, 2)
        QS_STEP(WB, WA, kb + 896u);  QS_ROW_D(WB, 3)
        QS_STEP(WA, WB, kb + 960u);  QS_ROW_D(WA, 4)
        QS_STEP(WB, WA, kb + 1024u); QS_ROW_D(WB, 5)
        QS_STEP(WA, WB, next_first); QS_ROW_D(WA, 6)
        WA = WB;  // 7 chunks: restore the "first chunk is in WA" invariant
      }
      i_cur = i_nxt;
#else
      const int i = cst->nat[k];
      const float Rs = cst->range[k];   // 2q * 2^-12
      const float* __restrict__ w = cst->tab + k * TS;
      float num = 0.0f, den = 0.0f;

      // Structural zeros: for horizontal frequency u = i & 7 the weight
      // T[p] - T[p+1] vanishes exactly whenever (x + 1) * u is a multiple of 8
      // (the two cosines coincide): x = 1,3,5 for u = 4 and x = 3 for u = 2,4,6;
      // same for the vertical differences with v = i >> 3 (whole rows y).  A zero
      // weight makes y = 0, so the term adds +0 to both sums -- skipping it is
      // exact (the host verifies the table entries are 0.0f,
      // qs_hip_consts_build).  That is 7.7 % of all terms at q3.
      // Mechanism: the 48 "optional" terms are emitted as inline-asm blocks that
      // carry their own wave-uniform scalar test and branch, so the compiler
      // sees straight-line code (branching in C++ made hipcc spill 240-430
      // B/lane and run 1.7x slower).  skip4 = 1 when the frequency is 4, even = 1
      // when it is 2, 4 or 6.
#if QS_SKIP_ZERO_WEIGHTS
#define QS_TERM_OPT(COND, A, B, W) { float d_, t_; \
        asm volatile( \
          "s_cmp_lg_u32 %[c], 0\n\t" \
          "s_cbranch_scc1 1f\n\t" \
          "v_sub_f32 %[d], %[a], %[b]\n\t" \
          "v_sub_f32 %[t], %[r], |%[d]| clamp\n\t" \
          "v_mul_f32 %[t], %[t], %[t]\n\t" \
          "v_mul_f32 %[d], %[d], %[t]\n\t" \
          "v_mul_f32 %[t], %[w], %[t]\n\t" \
          "v_mul_f32 %[d], %[d], %[t]\n\t" \
          "v_add_f32 %[n], %[n], %[d]\n\t" \
          "v_mul_f32 %[d], %[t], %[t]\n\t" \
          "v_add_f32 %[e], %[e], %[d]\n" \
          "1:" \
          : [n] "+v"(num), [e] "+v"(den), [d] "=&v"(d_), [t] "=&v"(t_) \
          : [a] "v"(A), [b] "v"(B), [w] "s"(W), [r] "s"(Rs), [c] "s"(COND) : "scc"); }
#else
#define QS_TERM_OPT(COND, A, B, W) QS_TERM(A, B, W)
#endif
      if (i & 7) {
        const int u = __builtin_amdgcn_readfirstlane(i & 7);   // wave-uniform by construction
        // plain bit arithmetic (a compare would be materialised in a VGPR, which the "s" operands reject)
        const int skip4 = __builtin_amdgcn_readfirstlane((u >> 2) & (~u >> 1) & ~u & 1);
        const int even = __builtin_amdgcn_readfirstlane(~u & 1);
        (void)skip4; (void)even;
#pragma unroll
        for (int y = 0; y < 8; ++y)
#pragma unroll
          for (int x = 0; x < 7; ++x) {
            if (x == 3) QS_TERM_OPT(even, px[y * 8 + x], px[y * 8 + x + 1], w[y * 8 + x])
            else if (x & 1) QS_TERM_OPT(skip4, px[y * 8 + x], px[y * 8 + x + 1], w[y * 8 + x])
            else QS_TERM(px[y * 8 + x], px[y * 8 + x + 1], w[y * 8 + x])
          }
      }
#pragma unroll
      for (int j = 0; j < 32; ++j) QS_TERM_D(bd[j], w[64 + j])
      if (i > 7) {
        const int v = __builtin_amdgcn_readfirstlane(i >> 3);
        const int skip4 = __builtin_amdgcn_readfirstlane((v >> 2) & (~v >> 1) & ~v & 1);
        const int even = __builtin_amdgcn_readfirstlane(~v & 1);
        (void)skip4; (void)even;
#pragma unroll
        for (int y = 0; y < 7; ++y)
#pragma unroll
          for (int x = 0; x < 8; ++x) {
            if (y == 3) QS_TERM_OPT(even, px[y * 8 + x], px[y * 8 + x + 8], w[96 + y * 8 + x])
            else if (y & 1) QS_TERM_OPT(skip4, px[y * 8 + x], px[y * 8 + x + 8], w[96 + y * 8 + x])
            else QS_TERM(px[y * 8 + x], px[y * 8 + x + 8], w[96 + y * 8 + x])
          }
      }
#undef QS_TERM_OPT
      if (DIAG) {
#pragma unroll
        for (int y = 0; y < 7; ++y)
#pragma unroll
          for (int x = 0; x < 7; ++x) {
            QS_TERM(px[y * 8 + x], px[y * 8 + x + 9], w[160 + y * 16 + x])
            QS_TERM(px[y * 8 + x + 1], px[y * 8 + x + 8], w[168 + y * 16 + x])
          }
      }

#endif
      // num'/den' = (num/den) * 2^-12: undo the scale (exact), then round
#ifdef QS_ABLATE_UPDATE
      if (num == 123.456f) lds_set_coef(col, i, (int)den);
      continue;
#endif
      const int r = f2i_x86(round_half_away((num / den) * 4096.0f));
      if (r != 0) {
        const int c0 = lds_coef(col, i);
        int orig, lo, hi;
        interval(c0, qk, x1k, x2k, orig, lo, hi);
        int v = (int)((uint32_t)c0 - (uint32_t)r);  // wraps like the x86 build
        v = min(max(v, lo), hi);
        lds_set_coef(col, i, v);
      }
    }
    kfirst = klast - 1;
  }

#if QS_SMEM_PIPELINE
  QS_SWAIT(WA);  // drain the last (unused) prefetch before the wave moves on
#endif

  // ---- rebalance (reference :1823-1848): scale AC energy back towards the
  // quantised original, inside each coefficient's interval.
  if (rebalance) {
    long long m0 = 0, m1 = 0;
#pragma unroll 9
    for (int n = 1; n < 64; ++n) {
      const int c = lds_coef(col, n);
      int orig, lo, hi;
      interval(c, cst->qn[n], cst->x1n[n], cst->x2n[n], orig, lo, hi);
      m0 += (long long)(c * orig);
      m1 += (long long)(orig * orig);
    }
    if (m1 > m0) {
      const int mul = (int)(((m1 << 13) + (m0 >> 1)) / m0);
#pragma unroll 9
      for (int n = 1; n < 64; ++n) {
        const int c = lds_coef(col, n);
        int orig, lo, hi;
        interval(c, cst->qn[n], cst->x1n[n], cst->x2n[n], orig, lo, hi);
        int v = (c * mul + 0x1000) >> 13;
        v = min(max(v, lo), hi);
        lds_set_coef(col, n, v);
      }
    }
  }
#if QS_IDCT_DOT2
  {  // back to row-major pairs for the coalesced write-back
    uint32_t cp[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) cp[m] = col[m * QS_LDS_PITCH];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int t = pair_slot(r), h = pair_half(r);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t a = cp[4 * (2 * j) + t], b = cp[4 * (2 * j + 1) + t];
        const uint32_t lo = h ? (a >> 16) : (a & 0xffffu);
        const uint32_t hi = h ? (b & 0xffff0000u) : (b << 16);
        col[(r * 4 + j) * QS_LDS_PITCH] = lo | hi;
      }
    }
  }
#endif
  wave_lds_sync();

  // ---- write back, coalesced; optional final +-1023 clamp (reference :2680-2686)
  {
    // recompute the per-lane store addresses from an opaque copy of the lane
    // id, so that the 8 load addresses of the prologue do not stay live (and
    // get spilled) across the whole kernel
    int lane2 = lane;
    asm volatile("" : "+v"(lane2));
    const int lane = lane2;
    const int m0 = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = j * 64 + lane;
      const uint32_t* s = lds + m0 * QS_LDS_PITCH + (j * 8 + (lane >> 3));
      uint32_t d[4] = {s[0], s[QS_LDS_PITCH], s[2 * QS_LDS_PITCH], s[3 * QS_LDS_PITCH]};
      if (final_clamp) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          int lo = (int16_t)(d[c] & 0xffff), hi = (int32_t)d[c] >> 16;
          lo = min(max(lo, -1023), 1023); hi = min(max(hi, -1023), 1023);
          d[c] = ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
        }
      }
      if (idx < nvec) gsrc[idx] = make_uint4(d[0], d[1], d[2], d[3]);
    }
  }
}

// --------------------------------------------------------------------------
// Kernel C: stand-alone final clamp (used when the last smoothing launch did
// not carry it: cancelled runs, refresh-only components).  8 coefs per lane.
__global__ void __launch_bounds__(256)
qs_clamp_kernel(int16_t* __restrict__ coef, size_t nvec) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  uint4* p = reinterpret_cast<uint4*>(coef);
  for (; i < nvec; i += stride) {
    uint4 v = p[i];
    uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int lo = (int16_t)(d[c] & 0xffff), hi = (int32_t)d[c] >> 16;
      lo = min(max(lo, -1023), 1023); hi = min(max(hi, -1023), 1023);
      d[c] = ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
    }
    p[i] = make_uint4(d[0], d[1], d[2], d[3]);
  }
}

// Kernel D: dequantise only (reference :2551-2566, the path taken when the
// component is not smoothed but the job goes on: coef *= quant, no clamp).
__global__ void __launch_bounds__(256)
qs_dequant_kernel(const QsConsts* __restrict__ cst, int16_t* __restrict__ coef, size_t nvec) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  uint4* p = reinterpret_cast<uint4*>(coef);
  for (; i < nvec; i += stride) {
    uint4 v = p[i];
    uint32_t d[4] = {v.x, v.y, v.z, v.w};
    const int j = (int)(i & 7);  // which eighth of the block
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int lo = (int16_t)(d[c] & 0xffff), hi = (int32_t)d[c] >> 16;
      lo *= cst->qraw[j * 8 + c * 2]; hi *= cst->qraw[j * 8 + c * 2 + 1];
      d[c] = ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
    }
    p[i] = make_uint4(d[0], d[1], d[2], d[3]);
  }
}

// --------------------------------------------------------------------------
// launchers (C++ linkage, used by qs_host.cpp through qs_launch.h)
#include "qs_launch.h"

void qs_launch_idct_plane(const QsConsts* cst, int16_t* coef, uint8_t* plane, int wblk, int hblk,
                          int first, int rep_top, int rep_bot, int* status, hipStream_t s) {
  const int nblk = wblk * hblk;
  hipLaunchKernelGGL(qs_idct_plane_kernel, dim3((nblk + 255) / 256), dim3(256), 0, s,
                     cst, coef, plane, wblk, hblk, qs_plane_pitch(wblk), first, rep_top, rep_bot, status);
}

void qs_launch_smooth_plane(const QsConsts* cst, int16_t* coef, const uint8_t* plane, int wblk, int hblk,
                            int diag, int rebalance, int final_clamp, int blk_begin, int blk_end, hipStream_t s) {
  const int n = blk_end - blk_begin;
  if (n <= 0) return;
  const int per_wg = 64 * QS_WAVES_PER_WG;
  const dim3 grid((n + per_wg - 1) / per_wg), block(per_wg);
  if (diag)
    hipLaunchKernelGGL(qs_smooth_plane_kernel<true>, grid, block, 0, s,
                       cst, coef, plane, wblk, hblk, qs_plane_pitch(wblk), rebalance, final_clamp, blk_begin, blk_end);
  else
    hipLaunchKernelGGL(qs_smooth_plane_kernel<false>, grid, block, 0, s,
                       cst, coef, plane, wblk, hblk, qs_plane_pitch(wblk), rebalance, final_clamp, blk_begin, blk_end);
}

void qs_launch_clamp(int16_t* coef, size_t nblk, hipStream_t s) {
  const size_t nvec = nblk * 8;
  const int grid = (int)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
  hipLaunchKernelGGL(qs_clamp_kernel, dim3(grid ? grid : 1), dim3(256), 0, s, coef, nvec);
}

void qs_launch_dequant(const QsConsts* cst, int16_t* coef, size_t nblk, hipStream_t s) {
  const size_t nvec = nblk * 8;
  const int grid = (int)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
  hipLaunchKernelGGL(qs_dequant_kernel, dim3(grid ? grid : 1), dim3(256), 0, s, cst, coef, nvec);
}
