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
// for the coalesced-load transpose).  147 VGPRs => 3 waves per SIMD.
//
// This translation unit holds the SHIPPED kernels only.  The variants that were built and
// measured on the way (and lost) are kept, compilable, under tools/experiments/ and are
// built only by tools/build_variants.sh; LABNOTES.md has the numbers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "qs_device.h"

/* QS_LDS_PITCH (65 dwords per coefficient-pair row: 64 lanes + 1 pad) comes from qs_device.h: the host packs LDS offsets into QsConsts::rec */

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
    idct8(col, 1024u);
#pragma unroll
    for (int j = 0; j < 8; ++j) ws[j * 8 + x] = (uint32_t)((int32_t)col[j] >> 11);
  }
}

// pass 2 for one row; folds +128 and rounding, clamps to 0..255
// (reference idct.h:509-538).
__device__ __forceinline__ void idct_pass2_row(uint32_t (&row)[8], int (&out)[8]) {
  idct8(row, 257u << 17);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // clamp BEFORE the shift (same result as clamping (x >> 18) to 0..255).
    // Toolchain hazard, ROCm 7.2 / gfx950: hipcc turns "shift, clamp to
    // 0..255, pack two bytes" into v_ashr_pk_u8_i32 and then ORs further
    // bytes into bits 31:16 of its result as if they were zero -- they are
    // not on MI355X (observed: corrupted pixels 6/7 of every row).  Clamping
    // first keeps that instruction out; csrc/Makefile greps the ISA for it.
    int z = (int32_t)row[j];
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
// of shift + convert + multiply.
#define QS_PIX_BITS 0x45000000u
__device__ __forceinline__ void idct_pass2_row_f(uint32_t (&row)[8], float (&out)[8]) {
  idct8(row, 257u << 17);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int z = (int32_t)row[j];
    z = min(max(z, 0), (256 << 18) - 1);
    // {0x11400 : z} >> 18  =  (0x11400 << 14) | (z >> 18)  =  0x45000000 | pixel
    out[j] = __builtin_bit_cast(float, __builtin_amdgcn_alignbit(QS_PIX_BITS >> 14, (uint32_t)z, 18));
  }
}
// byte n of a packed word of neighbour pixels, in the same representation
__device__ __forceinline__ float pix_from_byte(uint32_t v, int n) {
  return __builtin_bit_cast(float, QS_PIX_BITS | ((v >> (8 * n)) & 0xffu));
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
// one pixel row of a block (8 clamped pixels o[]) into the plane at org (the block's pixel (0, 0)), plus the parts of the
// clamp-to-edge apron that row owns: the columns x = -1 / x = w of edge blocks, and the replicated rows y = -1 / y = h
__device__ __forceinline__ void
store_pixel_row(uint8_t* __restrict__ org, int pitch, int y, const int (&o)[8], int bx, int by, int wblk, int hblk,
                int rep_top, int rep_bot) {
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

// 64 coefficients in registers (ws, natural order, sign-extended) -> the block's 64 pixels in the plane, plus the
// parts of the clamp-to-edge apron this block owns.  Shared by pass A and by the fused epilogue of pass B.
__device__ __forceinline__ void
idct_ws_to_plane(uint32_t (&ws)[64], uint8_t* __restrict__ plane, int wblk, int hblk, int pitch,
                 int rep_top, int rep_bot, int blk) {
  const int by = blk / wblk, bx = blk - by * wblk;
  idct_pass1(ws);
  uint8_t* org = plane + (size_t)(by * 8 + 1) * pitch + QS_APRON_X + bx * 8;
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    uint32_t row[8]; int o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) row[j] = ws[y * 8 + j];
    idct_pass2_row(row, o);
    store_pixel_row(org, pitch, y, o, bx, by, wblk, hblk, rep_top, rep_bot);
  }
}

__device__ __forceinline__ void
idct_block_to_plane(const QsConsts* __restrict__ cst, int16_t* __restrict__ coef,
                    uint8_t* __restrict__ plane, int wblk, int hblk, int pitch,
                    int first, int rep_top, int rep_bot, int* __restrict__ status, int blk) {
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
  idct_ws_to_plane(ws, plane, wblk, hblk, pitch, rep_top, rep_bot, blk);
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
__device__ __forceinline__ int lds_coef(const uint32_t* col, int i) {
  const lds_i16* p = reinterpret_cast<const lds_i16*>(col + (i >> 1) * QS_LDS_PITCH) + (i & 1);
  return *p;
}
__device__ __forceinline__ void lds_set_coef(uint32_t* col, int i, int v) {
  lds_i16* p = reinterpret_cast<lds_i16*>(col + (i >> 1) * QS_LDS_PITCH) + (i & 1);
  *p = (lds_i16)v;
}

// The block's 64 coefficients out of the lane's LDS column, sign-extended, for the refresh IDCT: 64 ds_read_i16.
// hipcc merges adjacent 16-bit LDS loads into dword reads and pays 48 VALU sign extensions per refresh for it (it
// optimises for LDS instructions; this kernel is bound by VALU issue slots while the LDS pipe idles), so the reads are
// issued by hand: four statements of sixteen loads, each with its own wait inside (form (i) of the guide's inline-asm
// rules: loads and their s_waitcnt in ONE statement, early-clobber outputs -- nothing is in flight at a statement's end).
// Row r of the column (coefficients 2r, 2r + 1) sits r * QS_LDS_PITCH dwords = r * 260 bytes in.
static_assert(QS_LDS_PITCH * 4 == 260, "the literal offsets below assume a 260-byte row pitch");
#define QS_LDS16_GROUP(BASE, W, N0) \
  asm volatile("ds_read_i16 %0, %16\n\tds_read_i16 %1, %16 offset:2\n\t" \
               "ds_read_i16 %2, %16 offset:260\n\tds_read_i16 %3, %16 offset:262\n\t" \
               "ds_read_i16 %4, %16 offset:520\n\tds_read_i16 %5, %16 offset:522\n\t" \
               "ds_read_i16 %6, %16 offset:780\n\tds_read_i16 %7, %16 offset:782\n\t" \
               "ds_read_i16 %8, %16 offset:1040\n\tds_read_i16 %9, %16 offset:1042\n\t" \
               "ds_read_i16 %10, %16 offset:1300\n\tds_read_i16 %11, %16 offset:1302\n\t" \
               "ds_read_i16 %12, %16 offset:1560\n\tds_read_i16 %13, %16 offset:1562\n\t" \
               "ds_read_i16 %14, %16 offset:1820\n\tds_read_i16 %15, %16 offset:1822\n\t" \
               "s_waitcnt lgkmcnt(0)" \
               : "=&v"(W[(N0) + 0]), "=&v"(W[(N0) + 1]), "=&v"(W[(N0) + 2]), "=&v"(W[(N0) + 3]), \
                 "=&v"(W[(N0) + 4]), "=&v"(W[(N0) + 5]), "=&v"(W[(N0) + 6]), "=&v"(W[(N0) + 7]), \
                 "=&v"(W[(N0) + 8]), "=&v"(W[(N0) + 9]), "=&v"(W[(N0) + 10]), "=&v"(W[(N0) + 11]), \
                 "=&v"(W[(N0) + 12]), "=&v"(W[(N0) + 13]), "=&v"(W[(N0) + 14]), "=&v"(W[(N0) + 15]) \
               : "v"(BASE) : "memory")
__device__ __forceinline__ void lds_read_block_i16(const uint32_t* col, uint32_t (&ws)[64]) {
  typedef const __attribute__((address_space(3))) uint32_t* lds_ptr;    // generic -> LDS address space: the DS byte address
  const uint32_t a0 = (uint32_t)(uintptr_t)(lds_ptr)col;
  const uint32_t a1 = a0 + 8 * 260, a2 = a0 + 16 * 260, a3 = a0 + 24 * 260;
  QS_LDS16_GROUP(a0, ws, 0);
  QS_LDS16_GROUP(a1, ws, 16);
  QS_LDS16_GROUP(a2, ws, 32);
  QS_LDS16_GROUP(a3, ws, 48);
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

// Scalar operands of the coefficient walk are fetched by hand-placed scalar loads: the compiler does not know
// about them, so every wait is tied to its buffer through a "+s" operand (uses cannot move above it).
typedef float qs_w16 __attribute__((ext_vector_type(16)));     // one 16-weight chunk
typedef int qs_i4 __attribute__((ext_vector_type(4)));         // one per-coefficient record (QsConsts::rec)
#define QS_SWAIT(BUF) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(BUF) : : "memory")

// waves per workgroup: the waves of a workgroup share nothing (each has its own
// LDS slice), so the size only sets the dispatch granularity
#define QS_WAVES_PER_WG 4
// register budget: 3 waves per SIMD (168 VGPRs; the kernel uses 147).  Measured: 2 waves per SIMD -11 %, a
// 4-wave budget (128 VGPRs, edge differences recomputed per term) -7.5 % (LABNOTES.md 4.2).
#define QS_SMOOTH_MIN_WAVES 3
// workgroups the chip holds at once = 256 CUs x 3 (a workgroup puts one wave on each of a CU's four SIMDs):
// the tail-round rule of qs_smooth_kernel.inc
#define QS_RESIDENT_WG (256 * QS_SMOOTH_MIN_WAVES * 4 / QS_WAVES_PER_WG)
// wave priority of the slow phases (refresh IDCT, coefficient update) and of the term streams, see
// qs_smooth_kernel.inc.  Any raised level works alike (1 / 2 / 3 measured equal); the OPPOSITE assignment is 5 %
// slower than no priorities at all.
#define QS_PRIO_SLOW 3
#define QS_PRIO_TERMS 0

// One term with a structurally-zero weight for some frequencies (small-plane kernel): the nine instructions carry
// their own wave-uniform scalar test and branch, so the compiler sees straight-line code (branching in C++ made
// hipcc spill 240-430 B/lane and run 1.7x slower).
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
// the same nine instructions as one opaque block (the section-specialised copies of qs_smooth_kernel.inc)
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

// the term on a difference that is already in a register (8 instructions)
#define QS_TERM_D_ASM(D, W) { float d_, t_; \
        asm volatile( \
          "v_sub_f32 %[t], %[r], |%[a]| clamp\n\t" \
          "v_mul_f32 %[t], %[t], %[t]\n\t" \
          "v_mul_f32 %[d], %[a], %[t]\n\t" \
          "v_mul_f32 %[t], %[w], %[t]\n\t" \
          "v_mul_f32 %[d], %[d], %[t]\n\t" \
          "v_add_f32 %[n], %[n], %[d]\n\t" \
          "v_mul_f32 %[d], %[t], %[t]\n\t" \
          "v_add_f32 %[e], %[e], %[d]" \
          : [n] "+v"(num), [e] "+v"(den), [d] "=&v"(d_), [t] "=&v"(t_) \
          : [a] "v"(D), [w] "s"(W), [r] "s"(Rs)); }
// The first QS_HOIST horizontal differences (rows 0, 1 and most of row 2) are kept in registers between refreshes
// instead of being re-formed for every coefficient: the kernel has ~20 VGPRs to spare under its 3-waves-per-SIMD
// budget (168).  Measured A/B, identical results (profiles/r04d_hoist): 14, 20 and 26 hoisted differences are all
// 2.3 % faster per plane launch at 8192^2 (1.597 -> 1.560 ms), 3 % at 128-512 block rows, +0.7 % in the 12-plane bench.
#define QS_HOIST 20
#ifndef QS_HOIST_DIAG
#define QS_HOIST_DIAG 32   /* the DIAGONALS kernel: 152 VGPRs at 20 */
#endif
// Column-restricted pass 1 of the refresh (qs_smooth_kernel.inc): 1 = park the pass-1 outputs of two block columns in LDS
#ifndef QS_Q1_SKIP
#define QS_Q1_SKIP 1   /* coefficients whose quantiser is 1 skip their term stream (QS_REC_Q1, qs_device.h): exact */
#endif
#ifndef QS_STASH
#define QS_STASH 1
#endif

// The recovery kernel (qs_smooth_kernel.inc): one plane, and a set of planes (job / batch layer: parameters come
// from the plane set in the kernarg segment instead of from scalar arguments)
#define QS_SMOOTH_KERNEL_NAME qs_smooth_plane_kernel
#include "qs_smooth_kernel.inc"
#undef QS_SMOOTH_KERNEL_NAME
#define QS_SMOOTH_KERNEL_NAME qs_smooth_set_kernel
#define QS_SMOOTH_SET 1
#include "qs_smooth_kernel.inc"
#undef QS_SMOOTH_KERNEL_NAME
#undef QS_SMOOTH_SET

// Small planes: the diagonal-parallel form of the same pass (qs_smooth_dp_kernel.inc)
#define QS_DP_KERNEL_NAME qs_smooth_dp_kernel
#include "qs_smooth_dp_kernel.inc"
#undef QS_DP_KERNEL_NAME
#define QS_DP_KERNEL_NAME qs_smooth_dp_set_kernel
#define QS_DP_SET 1
#include "qs_smooth_dp_kernel.inc"
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

// plane_next: the pixel plane of the next iteration (null: none), written by the kernel's fused epilogue (both forms of pass B)
void qs_launch_smooth_plane(const QsConsts* cst, int16_t* coef, const uint8_t* plane, uint8_t* plane_next, int rep_top, int rep_bot,
                            int wblk, int hblk, int diag, int rebalance, int final_clamp, int blk_begin, int blk_end, hipStream_t s) {
  const int n = blk_end - blk_begin;
  if (n <= 0) return;
  const int pitch = qs_plane_pitch(wblk);
  const int rep = (rep_top ? QS_PLANE_REP_TOP : 0) | (rep_bot ? QS_PLANE_REP_BOT : 0);
  if (const int nw = qs_dp_waves((n + 63) / 64)) {
    const dim3 g((n + 63) / 64), b(64 * nw);
#define QS_GO_DP(D, W) hipLaunchKernelGGL((qs_smooth_dp_kernel<D, W>), g, b, 0, s, cst, coef, plane, plane_next, rep, wblk, hblk, pitch, rebalance, final_clamp, blk_begin, blk_end)
    if (nw == 2) { if (diag) QS_GO_DP(true, 2); else QS_GO_DP(false, 2); }
    else if (diag) QS_GO_DP(true, QS_DP_WAVES); else QS_GO_DP(false, QS_DP_WAVES);
#undef QS_GO_DP
    return;
  }
  const int per_wg = 64 * QS_WAVES_PER_WG;
  const dim3 grid((n + per_wg - 1) / per_wg), block(per_wg);
#define QS_GO(K) hipLaunchKernelGGL(K, grid, block, 0, s, cst, coef, plane, plane_next, rep, wblk, hblk, pitch, rebalance, final_clamp, blk_begin, blk_end)
  if (diag) QS_GO(qs_smooth_plane_kernel<true>); else QS_GO(qs_smooth_plane_kernel<false>);
#undef QS_GO
}

void qs_launch_idct_set(const QsPlaneSet& set, int first, hipStream_t s) {
  const int nw = set.wave0[set.n];
  if (nw <= 0) return;
  hipLaunchKernelGGL(qs_idct_set_kernel, dim3((nw + 3) / 4), dim3(256), 0, s, set, first);
}

// planes whose QsPlaneRef::plane_next is set get the next iteration's pixel plane written by the same launch
void qs_launch_smooth_set(const QsPlaneSet& set, int diag, int final_clamp, hipStream_t s) {
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
