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
// kept as checked-in experiments, see LABNOTES.md): QS_SMOOTH_MIN_WAVES,
// QS_PIN_DIFFS, QS_PIN_EDGE, QS_SKIP_ZERO_WEIGHTS, QS_SMEM_PIPELINE,
// QS_IDCT_DOT2, QS_ABLATE_*.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "qs_device.h"

/* QS_LDS_PITCH: qs_device.h */

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

__device__ __forceinline__ uint32_t lshl13_add(uint32_t x, uint32_t b) {
  uint32_t r;
  asm("v_lshl_add_u32 %0, %1, 13, %2" : "=v"(r) : "v"(x), "v"(b));
  return r;
}

// 1-D LL&M inverse DCT butterfly, 13-bit constants, wrapping int32 arithmetic.
// Behaviour of reference idct.h:57-89.
// `bias` (the rounding constant of the descale that follows, plus the level shift
// in pass 2) enters through the two DC terms, from where it reaches all eight
// outputs exactly once: two adds instead of eight (integer ring arithmetic).
#ifndef QS_IDCT_FOLD_BIAS
#define QS_IDCT_FOLD_BIAS 1
#endif
__device__ __forceinline__ void idct8(uint32_t (&v)[8], uint32_t bias) {
  uint32_t z1, z2, z3, z4, z5, t0, t1, t2, t3, e0, e1, e2, e3;
  z2 = v[2]; z3 = v[6];
  z1 = mulc(z2 + z3, 4433);
  // a product with two consumers is made opaque: hipcc otherwise recomputes it inside two
  // v_mad_i32_i24 (it prices a multiply-add like an add; on gfx950 it costs about four)
  asm volatile("" : "+v"(z1));
  t2 = z1 - mulc(z3, 15137);
  t3 = z1 + mulc(z2, 6270);
  // (x << 13) + bias as ONE v_lshl_add_u32: left to itself hipcc emits v_mad_i32_i24 x, 8192, bias,
  // and integer multiplies cost about four adds on gfx950 (measured by ablation)
  t0 = lshl13_add(v[0] + v[4], bias);
  t1 = lshl13_add(v[0] - v[4], bias);
  e0 = t0 + t3; e3 = t0 - t3; e1 = t1 + t2; e2 = t1 - t2;
  t0 = v[7]; t1 = v[5]; t2 = v[3]; t3 = v[1];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; z4 = t1 + t3;
  z5 = mulc(z3 + z4, 9633);
  asm volatile("" : "+v"(z5));
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
#if QS_IDCT_FOLD_BIAS
    idct8(col, 1024u);
#pragma unroll
    for (int j = 0; j < 8; ++j) ws[j * 8 + x] = (uint32_t)((int32_t)col[j] >> 11);
#else
    idct8(col, 0u);
#pragma unroll
    for (int j = 0; j < 8; ++j) ws[j * 8 + x] = (uint32_t)((int32_t)(col[j] + 1024u) >> 11);
#endif
  }
}

// pass 2 for one row; folds +128 and rounding, clamps to 0..255
// (reference idct.h:509-538).
__device__ __forceinline__ void idct_pass2_row(uint32_t (&row)[8], int (&out)[8]) {
#if QS_IDCT_FOLD_BIAS
  idct8(row, 257u << 17);
#else
  idct8(row, 0u);
#endif
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // clamp BEFORE the shift (same result as clamping (x >> 18) to 0..255).
    // Toolchain hazard, ROCm 7.2 / gfx950: hipcc turns "shift, clamp to
    // 0..255, pack two bytes" into v_ashr_pk_u8_i32 and then ORs further
    // bytes into bits 31:16 of its result as if they were zero -- they are
    // not on MI355X (observed: corrupted pixels 6/7 of every row).  Clamping
    // first keeps that instruction out; csrc/Makefile greps the ISA for it.
#if QS_IDCT_FOLD_BIAS
    int z = (int32_t)row[j];
#else
    int z = (int32_t)(row[j] + (257u << 17));
#endif
    z = min(max(z, 0), (256 << 18) - 1);
    out[j] = z >> 18;
  }
}

// The recovery kernel keeps pixels as floats.  Only DIFFERENCES of pixels are
// ever used (QS_TERM), so a pixel p is stored as 2^11 + p * 2^-12 -- the float
// whose bit pattern is 0x45000000 | p (one ulp there is 2^-12): differences of
// two such floats are exact and equal (pa - pb) * 2^-12, what the scaled term
// arithmetic expects, and building one costs a single v_alignbit_b32 on the
// clamped pass-2 value (or one SDWA v_or on a packed neighbour byte) instead
// of shift + convert + multiply.  QS_PIX_MAGIC=0 keeps the plain p * 2^-12 form.
#ifndef QS_PIX_MAGIC
#define QS_PIX_MAGIC 1
#endif
#define QS_PIX_BITS 0x45000000u
__device__ __forceinline__ void idct_pass2_row_f(uint32_t (&row)[8], float (&out)[8]) {
#if QS_PIX_MAGIC
#if QS_IDCT_FOLD_BIAS
  idct8(row, 257u << 17);
#else
  idct8(row, 0u);
#endif
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#if QS_IDCT_FOLD_BIAS
    int z = (int32_t)row[j];
#else
    int z = (int32_t)(row[j] + (257u << 17));
#endif
    z = min(max(z, 0), (256 << 18) - 1);
    // {0x11400 : z} >> 18  =  (0x11400 << 14) | (z >> 18)  =  0x45000000 | pixel
    out[j] = __builtin_bit_cast(float, __builtin_amdgcn_alignbit(QS_PIX_BITS >> 14, (uint32_t)z, 18));
  }
#else
  int o[8];
  idct_pass2_row(row, o);
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = (float)o[j] * 0.000244140625f;
#endif
}
// byte n of a packed word of neighbour pixels, in the same representation
__device__ __forceinline__ float pix_from_byte(uint32_t v, int n) {
#if QS_PIX_MAGIC
  return __builtin_bit_cast(float, QS_PIX_BITS | ((v >> (8 * n)) & 0xffu));
#else
  return (float)((v >> (8 * n)) & 0xffu) * 0.000244140625f;
#endif
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
__device__ __forceinline__ void
idct_block_to_plane(const QsConsts* __restrict__ cst, int16_t* __restrict__ coef,
                    uint8_t* __restrict__ plane, int wblk, int hblk, int pitch,
                    int first, int rep_top, int rep_bot, int* __restrict__ status, int blk) {
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

__global__ void __launch_bounds__(256)
qs_idct_plane_kernel(const QsConsts* __restrict__ cst, int16_t* __restrict__ coef,
                     uint8_t* __restrict__ plane, int wblk, int hblk, int pitch,
                     int first, int rep_top, int rep_bot, int* __restrict__ status) {
  const int blk = blockIdx.x * 256 + threadIdx.x;
  if (blk >= wblk * hblk) return;
  idct_block_to_plane(cst, coef, plane, wblk, hblk, pitch, first, rep_top, rep_bot, status, blk);
}

#include "qs_devfn.h"

// pass A over a set of planes (whole planes, or bands whose halo-side apron rows are left alone)
__global__ void __launch_bounds__(256)
qs_idct_set_kernel(const QsPlaneSet set, int first) {
  const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (w >= set.wave0[set.n]) return;
  const int i = qs_set_find(set, w);
  const QsPlaneRef& r = set.ref[i];
  const int blk = (w - set.wave0[i]) * 64 + (threadIdx.x & 63);
  if (blk >= r.wblk * r.hblk) return;
  idct_block_to_plane(r.cst, r.coef, r.plane, r.wblk, r.hblk, r.pitch, first,
                      r.mode & QS_PLANE_REP_TOP, r.mode & QS_PLANE_REP_BOT, r.status, blk);
}

// --------------------------------------------------------------------------
// Kernel B: the recovery loop.  Reference quantsmooth.h:1396-1565 (main loop),
// :1566-1568 + :1823-1848 (rebalance), :2668-2689 (final clamp, optional).


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
// built, qs_tables.cpp), so products stay above 2^-110 and sums of such terms are
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
// Explicit double-buffered scalar weight prefetch (QS_STEP below), used by the
// low-occupancy kernel variant.  Measured on MI355X: with 4 waves per SIMD in
// flight the compiler's own s_load placement is covered by the other waves and
// is a few % faster; with 1-2 waves per SIMD (planes below ~190k blocks, e.g. a
// 1/8 band of an 8192^2 image) the explicit pipeline is 25-75 % faster.
typedef float qs_w16 __attribute__((ext_vector_type(16)));
// explicit scalar loads: the compiler does not know about them, so the wait is
// tied to the buffer through a "+s" operand (uses cannot move above it)
#define QS_SLOAD16(BUF, BYTEOFF) \
  asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(BUF) : "s"(tabp), "s"((uint32_t)__builtin_amdgcn_readfirstlane((int)(BYTEOFF))) : "memory")
#define QS_SWAIT(BUF) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(BUF) : : "memory")
// One pipeline step: wait for the chunk in CUR, then immediately issue the load
// of the following chunk into NXT.  The num/den operands pin the step between
// the accumulations of the previous chunk and those of this one -- without
// them the scheduler hoists/sinks the surrounding VALU work across the asm and
// the load ends up right in front of its own wait.
#define QS_STEP(CUR, NXT, BYTEOFF) \
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 %1, %4, %5" \
               : "+s"(CUR), "=&s"(NXT), "+v"(num), "+v"(den) \
               : "s"(tabp), "s"((uint32_t)__builtin_amdgcn_readfirstlane((int)(BYTEOFF))) : "memory")

// QS_REC_PREFETCH=1 (default): the per-coefficient scalars come from QsConsts::rec through ONE
// explicit scalar load at the top of the coefficient; 0 = left to the compiler, which fetches them
// with five global_load_dword + v_readfirstlane (kept for A/B runs)
#ifndef QS_REC_PREFETCH
#define QS_REC_PREFETCH 1
#endif
// QS_LAND_PIPELINE=1: chunk steps wait at their END for the load they issued at their start
// (nothing in flight across statements other than one straight-line run of terms), and the
// record of the next coefficient rides on the last step; 0 = wait at the start of the next step
#ifndef QS_LAND_PIPELINE
#define QS_LAND_PIPELINE 1
#endif
// per-coefficient record (QsConsts::rec), same discipline: issued right behind a weight-chunk
// load, complete after the next s_waitcnt lgkmcnt(0)
typedef int qs_i4 __attribute__((ext_vector_type(4)));
#define QS_SLOAD4(BUF, BYTEOFF) \
  asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(BUF) : "s"(recp), "s"((uint32_t)__builtin_amdgcn_readfirstlane((int)(BYTEOFF))) : "memory")
__device__ __forceinline__ float byte_f(uint32_t v, int n) { return (float)((v >> (8 * n)) & 0xffu); }

// waves per workgroup: the waves of a workgroup share nothing (each has its own
// LDS slice), so the size only sets the dispatch granularity
#ifndef QS_WAVES_PER_WG
#define QS_WAVES_PER_WG 4
#endif

// QS_FUSE_REBALANCE_SUMS: collect the rebalance step's two sums while the coefficients are
// updated (see qs_smooth_kernel.inc); 0 = separate pass over the block afterwards
#ifndef QS_FUSE_REBALANCE_SUMS
#define QS_FUSE_REBALANCE_SUMS 1
#endif
// QS_EDGE_INLINE=1: no register-resident edge differences (see qs_smooth_kernel.inc): with
// QS_SMOOTH_MIN_WAVES=4 the kernel fits 128 VGPRs = four waves per SIMD
#ifndef QS_EDGE_INLINE
#define QS_EDGE_INLINE 0
#endif
#ifndef QS_SMOOTH_OCCUPANCY       /* waves per SIMD the default kernel actually gets (tail-round rule) */
#define QS_SMOOTH_OCCUPANCY 3
#endif
// tail-round wave priority (see qs_smooth_kernel.inc); workgroups the chip holds at
// once = 256 CUs x 3 (the kernel's VGPR budget leaves room for 3 waves per SIMD and a
// workgroup puts one wave on each of a CU's four SIMDs)
// QS_REFRESH_SKIP=1: the refresh IDCT at an anti-diagonal start is skipped when no block of the
// wave changed a coefficient since the previous refresh (the reference's need_refresh,
// quantsmooth.h:1407-1409, made wave-uniform); exact by construction.
#ifndef QS_REFRESH_SKIP
#define QS_REFRESH_SKIP 1
#endif
// the nine instructions of one term as a string, operands by name (for multi-term asm blocks)
#define QS_TSTR(A, B, W) \
          "v_sub_f32 %[d], %[" #A "], %[" #B "]\n\t" \
          "v_sub_f32 %[t], %[r], |%[d]| clamp\n\t" \
          "v_mul_f32 %[t], %[t], %[t]\n\t" \
          "v_mul_f32 %[d], %[d], %[t]\n\t" \
          "v_mul_f32 %[t], %[" #W "], %[t]\n\t" \
          "v_mul_f32 %[d], %[d], %[t]\n\t" \
          "v_add_f32 %[n], %[n], %[d]\n\t" \
          "v_mul_f32 %[d], %[t], %[t]\n\t" \
          "v_add_f32 %[e], %[e], %[d]\n\t"
// one pixel row's seven horizontal differences (MODE 0), without x = 3 (MODE 1), without x = 1, 3, 5 (MODE 2)
#define QS_HROW_TERMS_0 QS_TSTR(p0, p1, w0) QS_TSTR(p1, p2, w1) QS_TSTR(p2, p3, w2) QS_TSTR(p3, p4, w3) QS_TSTR(p4, p5, w4) QS_TSTR(p5, p6, w5) QS_TSTR(p6, p7, w6)
#define QS_HROW_TERMS_1 QS_TSTR(p0, p1, w0) QS_TSTR(p1, p2, w1) QS_TSTR(p2, p3, w2) QS_TSTR(p4, p5, w4) QS_TSTR(p5, p6, w5) QS_TSTR(p6, p7, w6)
#define QS_HROW_TERMS_2 QS_TSTR(p0, p1, w0) QS_TSTR(p2, p3, w2) QS_TSTR(p4, p5, w4) QS_TSTR(p6, p7, w6)
#define QS_HROW_ASM(P, W, WO, MODE) { float d_, t_; \
        asm volatile(QS_HROW_TERMS_##MODE \
          : [n] "+v"(num), [e] "+v"(den), [d] "=&v"(d_), [t] "=&v"(t_) \
          : [p0] "v"((P)[0]), [p1] "v"((P)[1]), [p2] "v"((P)[2]), [p3] "v"((P)[3]), [p4] "v"((P)[4]), [p5] "v"((P)[5]), \
            [p6] "v"((P)[6]), [p7] "v"((P)[7]), [w0] "s"(W[(WO) + 0]), [w1] "s"(W[(WO) + 1]), [w2] "s"(W[(WO) + 2]), \
            [w3] "s"(W[(WO) + 3]), [w4] "s"(W[(WO) + 4]), [w5] "s"(W[(WO) + 5]), [w6] "s"(W[(WO) + 6]), [r] "s"(Rs)); }
// one row of eight vertical differences: row P against row Q (the next pixel row)
#define QS_VROW_ASM(P, Q, W, WO) { float d_, t_; \
        asm volatile(QS_TSTR(p0, q0, w0) QS_TSTR(p1, q1, w1) QS_TSTR(p2, q2, w2) QS_TSTR(p3, q3, w3) \
                     QS_TSTR(p4, q4, w4) QS_TSTR(p5, q5, w5) QS_TSTR(p6, q6, w6) QS_TSTR(p7, q7, w7) \
          : [n] "+v"(num), [e] "+v"(den), [d] "=&v"(d_), [t] "=&v"(t_) \
          : [p0] "v"((P)[0]), [p1] "v"((P)[1]), [p2] "v"((P)[2]), [p3] "v"((P)[3]), [p4] "v"((P)[4]), [p5] "v"((P)[5]), \
            [p6] "v"((P)[6]), [p7] "v"((P)[7]), [q0] "v"((Q)[0]), [q1] "v"((Q)[1]), [q2] "v"((Q)[2]), [q3] "v"((Q)[3]), \
            [q4] "v"((Q)[4]), [q5] "v"((Q)[5]), [q6] "v"((Q)[6]), [q7] "v"((Q)[7]), \
            [w0] "s"(W[(WO) + 0]), [w1] "s"(W[(WO) + 1]), [w2] "s"(W[(WO) + 2]), [w3] "s"(W[(WO) + 3]), \
            [w4] "s"(W[(WO) + 4]), [w5] "s"(W[(WO) + 5]), [w6] "s"(W[(WO) + 6]), [w7] "s"(W[(WO) + 7]), [r] "s"(Rs)); }
// QS_PHASE_PRIO=1: a wave runs the slow phases of the coefficient walk -- the refresh IDCT (integer butterflies, a third
// of them half-rate multiplies) and the coefficient update (division, interval, LDS read-modify-write) -- at RAISED wave
// priority (s_setprio) and its term streams at the base priority.  With three waves per SIMD the pipe idles when two of
// them are in a slow phase at once and the third cannot fill it alone (tools/timeline.py); at raised priority a wave leaves
// its slow phase as fast as the hardware allows and such overlaps become rare.  QS_PRIO_REFRESH / _UPDATE / _TERMS: the levels.
// Measured (profiles/r03l_phase_prio, one session, 20 steps each): q3 226.8 -> 255.3 M blocks/s (+12.6 %), q4 144.8 -> 166.3
// (+15 %); refresh only +10 %, update only +2 %; the levels do not matter (1 / 2 / 3 alike), the OPPOSITE assignment
// (terms above the slow phases) is 5 % slower than no priorities at all.  Identical results (same hashes).
#ifndef QS_PHASE_PRIO
#define QS_PHASE_PRIO 1
#endif
#ifndef QS_PRIO_REFRESH
#define QS_PRIO_REFRESH 3
#endif
#ifndef QS_PRIO_UPDATE
#define QS_PRIO_UPDATE 3
#endif
#ifndef QS_PRIO_TERMS
#define QS_PRIO_TERMS 0
#endif
// ... and the same in the diagonal-parallel (small-plane) kernel
#ifndef QS_DP_PHASE_PRIO
#define QS_DP_PHASE_PRIO 1
#endif
// QS_EARLY_C0=1: the coefficient's LDS read is issued before the division of its update step (latency hiding)
#ifndef QS_EARLY_C0
#define QS_EARLY_C0 1
#endif
// QS_TIMELINE=1 (measurement builds only, tools/timeline.py): every wave of qs_smooth_plane_kernel accumulates the
// shader cycles (s_memtime) it spends in the three phases of the coefficient walk -- refresh IDCT, term stream,
// coefficient update -- plus its staging prologue and its whole life, and lane 0 stores them into a device buffer
// registered through qs_hip_debug_timeline().  Each stamp is a scalar-memory read with its own wait (~3 % perturbation).
#ifndef QS_TIMELINE
#define QS_TIMELINE 0
#endif
#if QS_TIMELINE
__device__ unsigned long long* qs_tl_buf = nullptr;
#define QS_TL_NOW(T) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T) : : "memory")
extern "C" int qs_hip_debug_timeline(void* dev_buf) {
  unsigned long long* p = static_cast<unsigned long long*>(dev_buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(qs_tl_buf), &p, sizeof p) == hipSuccess ? 0 : -1;
}
#endif
// QS_SECTION_SPEC=1: the zero-weight skip is decided once per horizontal / vertical section (three
// specialised copies of the section) instead of by a scalar compare-and-branch in front of 48 terms
// (=1, one opaque asm block per term: +0.7 % before the phase priorities of QS_PHASE_PRIO -- inside the noise --, +2.1 % with
//  them, 258.0 against 252.5-253.0 M blocks/s, and 4-5 % for a lone wave per SIMD; =2, one asm block per pixel row: 7 % slower)
#ifndef QS_SECTION_SPEC
#define QS_SECTION_SPEC 1
#endif
#ifndef QS_TAIL_PRIO
#define QS_TAIL_PRIO 1
#endif
#define QS_RESIDENT_WG (256 * QS_SMOOTH_OCCUPANCY * 4 / QS_WAVES_PER_WG)
// measurement only: extra (unused) LDS dwords per wave, to cap how many workgroups a CU holds
// without touching the code (occupancy experiments: 2400 -> 2 waves per SIMD), see LABNOTES.md
#ifndef QS_LDS_EXTRA
#define QS_LDS_EXTRA 0
#endif

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
// the same nine instructions as one opaque block (QS_SECTION_SPEC: the three specialised copies of a section must
// not share subexpressions, or hipcc hoists ~100 pixel differences above the branch and spills them)
#define QS_TERM_ASM(A, B, W) { float d_, t_; \
        asm volatile( \
          "v_sub_f32 %[d], %[a], %[b]\n\t" \
          "v_sub_f32 %[t], %[r], |%[d]| clamp\n\t" \
          "v_mul_f32 %[t], %[t], %[t]\n\t" \
          "v_mul_f32 %[d], %[d], %[t]\n\t" \
          "v_mul_f32 %[t], %[w], %[t]\n\t" \
          "v_mul_f32 %[d], %[d], %[t]\n\t" \
          "v_add_f32 %[n], %[n], %[d]\n\t" \
          "v_mul_f32 %[d], %[t], %[t]\n\t" \
          "v_add_f32 %[e], %[e], %[d]" \
          : [n] "+v"(num), [e] "+v"(den), [d] "=&v"(d_), [t] "=&v"(t_) \
          : [a] "v"(A), [b] "v"(B), [w] "s"(W), [r] "s"(Rs)); }

// Two instantiations of the kernel body (qs_smooth_kernel.inc):
//   qs_smooth_plane_kernel      the default: scalar weights streamed through an
//                               explicit double buffer (QS_STEP), 168 VGPRs, no scratch;
//                               does not depend on other waves to hide latency
//   qs_smooth_plane_kernel_alt  weight loads left to the compiler, 128 VGPRs (4 waves
//                               per SIMD); kept for A/B runs (QS_FORCE_VARIANT=0)
#define QS_SMOOTH_KERNEL_NAME qs_smooth_plane_kernel
#define QS_SMEM_PIPELINE 1
#ifndef QS_SMOOTH_MIN_WAVES
#define QS_SMOOTH_MIN_WAVES 3
#endif
#include "qs_smooth_kernel_r03.inc"
#undef QS_SMOOTH_KERNEL_NAME
#undef QS_SMEM_PIPELINE
#undef QS_SMOOTH_MIN_WAVES

// the same kernel over a set of planes (job / batch layer): parameters come
// from the plane set in the kernarg segment instead of from scalar arguments
#define QS_SMOOTH_KERNEL_NAME qs_smooth_set_kernel
#define QS_SMEM_PIPELINE 1
#define QS_SMOOTH_MIN_WAVES 3
#define QS_SMOOTH_SET 1
#include "qs_smooth_kernel_r03.inc"
#undef QS_SMOOTH_KERNEL_NAME
#undef QS_SMEM_PIPELINE
#undef QS_SMOOTH_MIN_WAVES
#undef QS_SMOOTH_SET

#define QS_SMOOTH_KERNEL_NAME qs_smooth_plane_kernel_alt
#define QS_SMEM_PIPELINE 0
#define QS_SMOOTH_MIN_WAVES 4
#include "qs_smooth_kernel_r03.inc"
#undef QS_SMOOTH_KERNEL_NAME
#undef QS_SMEM_PIPELINE
#undef QS_SMOOTH_MIN_WAVES

// Small planes: the diagonal-parallel form of the same pass (qs_smooth_dp_kernel.inc)
#ifndef QS_DP_SHARED_REFRESH
#define QS_DP_SHARED_REFRESH 1
#endif
#define QS_DP_KERNEL_NAME qs_smooth_dp_kernel
#include "qs_smooth_dp_kernel_r03.inc"
#undef QS_DP_KERNEL_NAME
#define QS_DP_KERNEL_NAME qs_smooth_dp_set_kernel
#define QS_DP_SET 1
#include "qs_smooth_dp_kernel_r03.inc"
#undef QS_DP_KERNEL_NAME
#undef QS_DP_SET

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
// launchers (C++ linkage, used by qs_planes.cpp and qs_job.cpp through qs_launch.h)
#include "qs_launch.h"

void qs_launch_idct_plane(const QsConsts* cst, int16_t* coef, uint8_t* plane, int wblk, int hblk,
                          int first, int rep_top, int rep_bot, int* status, hipStream_t s) {
  const int nblk = wblk * hblk;
  hipLaunchKernelGGL(qs_idct_plane_kernel, dim3((nblk + 255) / 256), dim3(256), 0, s,
                     cst, coef, plane, wblk, hblk, qs_plane_pitch(wblk), first, rep_top, rep_bot, status);
}

// Which form of pass B a launch of `groups` 64-block groups gets.  Measured on MI355X
// (tools/bench_sizes.py, profiles/r02f_small_planes/run11_*; us per launch, q3 / q4):
//   groups   one block per lane   diagonal-parallel, 4 waves   2 waves
//   <= 256        199 / 295             82 / 114               122 / 176
//      512        203 / 301             99 / 142               143 / 211
//      768        208 / 309            128 / 195               147 / 218
//     1024        215 / 319            180 / 266               157 / 235
//     1536        255 / 380            241 / 362               208 / 325
//     2048        264 / 397            312 / 473               316 / 484
// 4 waves per group = one per SIMD, three workgroups per CU at the kernel's 3 waves per SIMD: all
// resident up to 768 groups; 2 waves per group keeps up to 1280 groups resident (LDS-bound: 26.7 KB
// per workgroup) and wins up to 1536; beyond that the chip is full anyway and one block per lane
// does the least work.  (6 waves per group were tried: a 6-wave workgroup puts two waves on two of
// the four SIMDs, only ONE such workgroup fits a CU, and it is no faster than 4 waves even below 256
// groups.)  QS_HIP_DP=0 switches the small-plane kernel off (A/B runs, tests of both forms);
// QS_HIP_DP_GROUPS / QS_HIP_DP_GROUPS2 move the two limits.
#define QS_DP_WAVES 4
static int qs_dp_waves(int groups) {
  static const int on = [] { const char* v = getenv("QS_HIP_DP"); return v ? atoi(v) : 1; }();
  static const int lim = [] { const char* v = getenv("QS_HIP_DP_GROUPS"); return v ? atoi(v) : 768; }();
  static const int lim2 = [] { const char* v = getenv("QS_HIP_DP_GROUPS2"); return v ? atoi(v) : 1536; }();
  if (!on) return 0;
  return groups <= lim ? QS_DP_WAVES : groups <= lim2 ? 2 : 0;
}

// (round 4: the product's launchers take the NEXT iteration's pixel plane -- the shipped kernels write it themselves.
//  These round-3 kernels do not: the launchers here keep the interface alive for the variant builds with the unfused
//  order -- pass B without the clamp, a stand-alone pass A into the next plane, then the clamp.  Whole planes only.)
static void qs_r03_smooth_plane(const QsConsts* cst, int16_t* coef, const uint8_t* plane, int wblk, int hblk,
                                int diag, int rebalance, int final_clamp, int blk_begin, int blk_end, hipStream_t s);
void qs_launch_smooth_plane(const QsConsts* cst, int16_t* coef, const uint8_t* plane, uint8_t* plane_next, int rep_top, int rep_bot,
                            int wblk, int hblk, int diag, int rebalance, int final_clamp, int blk_begin, int blk_end, hipStream_t s) {
  if (!plane_next) { qs_r03_smooth_plane(cst, coef, plane, wblk, hblk, diag, rebalance, final_clamp, blk_begin, blk_end, s); return; }
  if (blk_begin != 0 || blk_end != wblk * hblk) abort();     // (the experiments build has no partial-range pass A)
  qs_r03_smooth_plane(cst, coef, plane, wblk, hblk, diag, rebalance, 0, blk_begin, blk_end, s);
  qs_launch_idct_plane(cst, coef, plane_next, wblk, hblk, 0, rep_top, rep_bot, nullptr, s);
  if (final_clamp) qs_launch_clamp(coef, (size_t)wblk * hblk, s);
}
static void qs_r03_smooth_plane(const QsConsts* cst, int16_t* coef, const uint8_t* plane, int wblk, int hblk,
                                int diag, int rebalance, int final_clamp, int blk_begin, int blk_end, hipStream_t s) {
  const int n = blk_end - blk_begin;
  if (n <= 0) return;
  if (const int nw = qs_dp_waves((n + 63) / 64)) {
    const dim3 g((n + 63) / 64), b(64 * nw);
    const int pitch_ = qs_plane_pitch(wblk);
#define QS_GO_DP(D, W) hipLaunchKernelGGL((qs_smooth_dp_kernel<D, W>), g, b, 0, s, cst, coef, plane, wblk, hblk, pitch_, rebalance, final_clamp, blk_begin, blk_end)
    if (nw == 2) { if (diag) QS_GO_DP(true, 2); else QS_GO_DP(false, 2); }
    else if (diag) QS_GO_DP(true, QS_DP_WAVES); else QS_GO_DP(false, QS_DP_WAVES);
#undef QS_GO_DP
    return;
  }
  const int per_wg = 64 * QS_WAVES_PER_WG;
  const dim3 grid((n + per_wg - 1) / per_wg), block(per_wg);
  // The pipelined kernel is the default at every size: measured A/B on MI355X it
  // is 40-60 % faster than the compiler-scheduled one on planes that leave SIMDs
  // with 1-2 waves (1448^2 .. 2880^2, i.e. also a 1/8 band of 8192^2) and 0-2 %
  // faster on 4096^2 .. 8192^2.  QS_FORCE_VARIANT=0 selects the other for A/B runs.
#ifdef QS_FORCE_VARIANT
  const bool alt = QS_FORCE_VARIANT == 0;
#else
  const bool alt = false;
#endif
  const int pitch = qs_plane_pitch(wblk);
#define QS_GO(K) hipLaunchKernelGGL(K, grid, block, 0, s, cst, coef, plane, wblk, hblk, pitch, rebalance, final_clamp, blk_begin, blk_end)
  if (alt) { if (diag) QS_GO(qs_smooth_plane_kernel_alt<true>); else QS_GO(qs_smooth_plane_kernel_alt<false>); }
  else     { if (diag) QS_GO(qs_smooth_plane_kernel<true>);     else QS_GO(qs_smooth_plane_kernel<false>); }
#undef QS_GO
}

void qs_launch_idct_set(const QsPlaneSet& set, int first, hipStream_t s) {
  const int nw = set.wave0[set.n];
  if (nw <= 0) return;
  hipLaunchKernelGGL(qs_idct_set_kernel, dim3((nw + 3) / 4), dim3(256), 0, s, set, first);
}

static void qs_r03_smooth_set(const QsPlaneSet& set, int diag, int final_clamp, hipStream_t s);
void qs_launch_smooth_set(const QsPlaneSet& set, int diag, int final_clamp, hipStream_t s) {
  bool any_next = false;
  for (int i = 0; i < set.n; ++i) any_next = any_next || set.ref[i].plane_next;
  if (!any_next) { qs_r03_smooth_set(set, diag, final_clamp, s); return; }
  qs_r03_smooth_set(set, diag, 0, s);                        // unfused order: pass B, pass A into the next planes, clamp
  for (int i = 0; i < set.n; ++i) {
    const QsPlaneRef& r = set.ref[i];
    if (r.plane_next)
      qs_launch_idct_plane(r.cst, r.coef, r.plane_next, r.wblk, r.hblk, 0, r.mode & QS_PLANE_REP_TOP, r.mode & QS_PLANE_REP_BOT, nullptr, s);
    if (final_clamp) qs_launch_clamp(r.coef, (size_t)r.wblk * r.hblk, s);
  }
}
static void qs_r03_smooth_set(const QsPlaneSet& set, int diag, int final_clamp, hipStream_t s) {
  const int nw = set.wave0[set.n];
  if (nw <= 0) return;
  if (const int dw = qs_dp_waves(nw)) {
    const dim3 g(nw), b(64 * dw);
#define QS_GO_DP(D, W) hipLaunchKernelGGL((qs_smooth_dp_set_kernel<D, W>), g, b, 0, s, set, final_clamp)
    if (dw == 2) { if (diag) QS_GO_DP(true, 2); else QS_GO_DP(false, 2); }
    else if (diag) QS_GO_DP(true, QS_DP_WAVES); else QS_GO_DP(false, QS_DP_WAVES);
#undef QS_GO_DP
    return;
  }
  const dim3 grid((nw + QS_WAVES_PER_WG - 1) / QS_WAVES_PER_WG), block(64 * QS_WAVES_PER_WG);
  if (diag) hipLaunchKernelGGL(qs_smooth_set_kernel<true>, grid, block, 0, s, set, final_clamp);
  else      hipLaunchKernelGGL(qs_smooth_set_kernel<false>, grid, block, 0, s, set, final_clamp);
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
