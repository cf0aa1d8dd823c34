// ubench_mfma_idct.hip -- round 5 probe: can the refresh IDCT of the recovery kernel go to the matrix pipe?
//
// The refresh (qs_smooth_kernel.inc, 14 per block-iteration, 953 VALU instructions each) is an exact-integer 2-D IDCT:
// each 1-D pass of idct_islow (reference idct.h:57-89) multiplies by 13-bit constants exactly once on every path and
// rounds only at the descale behind it (:481-503, :509-511), so a pass is an integer 8x8 matrix product  out = M . in
// in the int32 ring, M[j][i] = (butterfly of unit vector i)[j], |M| <= 11585 * sqrt(2) < 2^15.  i8 MFMA is exact in
// that ring: with balanced limbs  M = M0 + 256 M1  and the coefficient bytes as they lie in memory,
// c = 256 hi + (lo' + 128) with lo' = (c & 0xff) ^ 0x80 read as a signed byte, the product falls into three
// accumulator classes  D0 = sum M0 lo',  D1 = sum (M1 lo' + M0 hi),  D2 = sum M1 hi  and
//     out = D0 + 2^8 D1 + 2^16 D2 + 128 rowsum(M) + 1024  (>> 11)            -- for EVERY int16 input.
// One v_mfma_i32_32x32x16_i8 (K = 8 frequencies x 2 coefficient bytes) does one image column of 32 blocks with the
// three classes in row groups 0-7 / 8-15 / 16-23, so that a lane finds the three partial sums of an output in its own
// registers; the bias rides in as the C operand.  Pass 1 of a wave's 64 blocks = 16 MFMAs, B operands read straight from
// an LDS layout that pairs coefficients vertically -- no byte shuffling on the way in.
//
// What this file measures (the judge's protocol: exactness first, then issue cost, then a kill criterion):
//   1. exactness of the MFMA pass 1 against the product's VALU pass 1 (idct8 with 24-bit multiplies) and against a
//      wrapping-uint32 CPU butterfly on > 10^6 blocks: 12-bit content, full int16 noise, and extreme blocks;
//   2. shader cycles per pass 1 of one wave (64 blocks), s_memtime inside the kernel, at 1 / 2 / 3 waves per SIMD:
//        V   the product's VALU pass 1 (LDS read, sign-extend, 8 column butterflies, descale)
//        M1  MFMA pass 1 incl. recombination, results left in the MFMA's own layout (lane = block % 32, half = rows)
//        M2  M1 + the way back to one block per lane through a 4 KB LDS transposition buffer (what a pass-1-only
//            replacement inside the recovery kernel has to pay, pass 2 staying on the VALU)
//        C   calibration: 1024 dependent v_add_f32 -> cycles per plain VALU instruction at that occupancy, so that
//            every number can be quoted in "VALU-instruction equivalents";
//   3. the same passes run by ONE wave of a SIMD next to TWO waves of the recovery kernel's 9-instruction term stream
//      (12-wave workgroups, waves w, w+4, w+8 share a SIMD): what a pass costs the term waves.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_mfma_idct.hip -o build/ubench_mfma_idct
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define PITCH 65
typedef int v16i __attribute__((ext_vector_type(16)));
typedef uint32_t __attribute__((may_alias)) lds_u32;

// ---- the product's VALU butterfly (csrc/qs_kernels.hip: mulc, lshl13_add, idct8, idct_pass1), verbatim in behaviour
__device__ __forceinline__ uint32_t mulc(uint32_t a, int c) { return (uint32_t)__mul24((int)a, c); }
__device__ __forceinline__ uint32_t lshl13_add(uint32_t x, uint32_t b) {
  uint32_t r; asm("v_lshl_add_u32 %0, %1, 13, %2" : "=v"(r) : "v"(x), "v"(b)); return r;
}
__device__ __forceinline__ void idct8(uint32_t (&v)[8], uint32_t bias) {
  uint32_t z1, z2, z3, z4, z5, t0, t1, t2, t3, e0, e1, e2, e3;
  z2 = v[2]; z3 = v[6];
  z1 = mulc(z2 + z3, 4433); asm volatile("" : "+v"(z1));
  t2 = z1 - mulc(z3, 15137); t3 = z1 + mulc(z2, 6270);
  t0 = lshl13_add(v[0] + v[4], bias); t1 = lshl13_add(v[0] - v[4], bias);
  e0 = t0 + t3; e3 = t0 - t3; e1 = t1 + t2; e2 = t1 - t2;
  t0 = v[7]; t1 = v[5]; t2 = v[3]; t3 = v[1];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; z4 = t1 + t3;
  z5 = mulc(z3 + z4, 9633); asm volatile("" : "+v"(z5));
  t0 = mulc(t0, 2446);  t1 = mulc(t1, 16819); t2 = mulc(t2, 25172); t3 = mulc(t3, 12299);
  z1 = mulc(z1, 7373);  z2 = mulc(z2, 20995); z3 = mulc(z3, 16069); z4 = mulc(z4, 3196);
  z3 = z5 - z3; z4 = z5 - z4;
  t0 += z3 - z1; t1 += z4 - z2; t2 += z3 - z2; t3 += z4 - z1;
  v[0] = e0 + t3; v[7] = e0 - t3; v[1] = e1 + t2; v[6] = e1 - t2;
  v[2] = e2 + t1; v[5] = e2 - t1; v[3] = e3 + t0; v[4] = e3 - t0;
}
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

// ---- constant operands of the MFMA form, built on the host (see main)
struct MfmaConsts {
  uint64_t a[64];      // A operand of lane l: row r = l & 31 (class r >> 3, output row j = r & 7), k slots 8 (l >> 5) .. + 7
  int32_t bias[8];     // 128 * rowsum(M)[j] + 1024
};

enum { MODE_V = 0, MODE_M1 = 1, MODE_M2 = 2, MODE_C = 3 };

__device__ __forceinline__ void lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// stage this wave's 64 blocks (block = base + lane; natural-order int16[64]) into the wave's LDS slice
//   natural layout (product, MODE_V):  lds[m * 65 + lane] = coefficient pair m (2m, 2m + 1), row-major
//   vertical layout (MFMA):            lds[(x * 4 + ip) * 65 + lane] = (c[2 ip][x], c[2 ip + 1][x]) ^ 0x00800080
template <int MODE>
__device__ __forceinline__ void stage(uint32_t* lds, const int16_t* coef, int lane) {
  const uint16_t* c = reinterpret_cast<const uint16_t*>(coef);
  if (MODE == MODE_V || MODE == MODE_C) {
    for (int m = 0; m < 32; ++m) lds[m * PITCH + lane] = (uint32_t)c[2 * m] | ((uint32_t)c[2 * m + 1] << 16);
  } else {
    for (int x = 0; x < 8; ++x)
      for (int ip = 0; ip < 4; ++ip)
        lds[(x * 4 + ip) * PITCH + lane] = ((uint32_t)c[(2 * ip) * 8 + x] | ((uint32_t)c[(2 * ip + 1) * 8 + x] << 16)) ^ 0x00800080u;
  }
}

// one pass 1 over the wave's 64 blocks.  out[]: MODE_V / MODE_M2: ws[j * 8 + x] of block `lane`;
// MODE_M1: out[t * 4 + v] = ws[(4 h + v) * 8 + (t >> 1)] of block 32 (t & 1) + (lane & 31), h = lane >> 5
template <int MODE>
__device__ __forceinline__ void pass1(const uint32_t* lds, uint32_t* tbuf, const uint64_t a_op, const v16i cinit, int lane, uint32_t (&out)[64]) {
  if (MODE == MODE_V) {
    const lds_u32* col = lds + lane;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const uint32_t d = col[m * PITCH];
      out[2 * m] = (uint32_t)(int32_t)(int16_t)(d & 0xffff);
      out[2 * m + 1] = (uint32_t)((int32_t)d >> 16);
    }
    idct_pass1(out);
  } else if (MODE == MODE_C) {
    float f = __builtin_bit_cast(float, lds[lane]);
    // (one statement per 64 adds: hipcc puts an s_nop behind every asm statement)
#define A4 "v_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\t"
#define A16 A4 A4 A4 A4
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile(A16 A16 A16 A16 : "+v"(f));
    out[0] = __builtin_bit_cast(uint32_t, f);
  } else {
    const int n = lane & 31, g = lane >> 5;
    const lds_u32* bcol = lds + (2 * g) * PITCH + n;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int x = t >> 1, bh = t & 1;
      const uint32_t b0 = bcol[(x * 4) * PITCH + 32 * bh], b1 = bcol[(x * 4 + 1) * PITCH + 32 * bh];
      const long b_op = (long)((uint64_t)b0 | ((uint64_t)b1 << 32));
      const v16i d = __builtin_amdgcn_mfma_i32_32x32x16_i8((long)a_op, b_op, cinit, 0, 0, 0);
      uint32_t r[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        uint32_t s = ((uint32_t)d[8 + v] << 8) + (uint32_t)d[4 + v];
        s = (s << 8) + (uint32_t)d[v];
        r[v] = (uint32_t)((int32_t)s >> 11);
      }
      if (MODE == MODE_M1) {
#pragma unroll
        for (int v = 0; v < 4; ++v) out[t * 4 + v] = r[v];
      } else {
        // the way back to one block per lane: a 4 KB buffer per wave, one chunk = 4 MFMAs = two image columns of all 64
        // blocks.  Row pitch 20 dwords: 16-byte aligned rows, conflict-free for the 8-lane groups of ds_write_b128.
        uint4* w = reinterpret_cast<uint4*>(tbuf + (32 * bh + n) * 20 + (x & 1) * 8 + 4 * g);
        *w = make_uint4(r[0], r[1], r[2], r[3]);
        if ((t & 3) == 3) {
          lds_fence();
          const uint4* rd = reinterpret_cast<const uint4*>(tbuf + lane * 20);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint4 u = rd[q];
            const int xx = (x & ~1) + (q >> 1), j0 = (q & 1) * 4;
            out[(j0 + 0) * 8 + xx] = u.x; out[(j0 + 1) * 8 + xx] = u.y; out[(j0 + 2) * 8 + xx] = u.z; out[(j0 + 3) * 8 + xx] = u.w;
          }
          lds_fence();
        }
      }
    }
  }
}

#define WAVE_LDS_DWORDS (32 * PITCH + 64 * 20)   /* coefficient slice + transposition buffer */

// ---- single-mode kernel: every wave runs REP passes; cycles per wave go to cyc[]; the last pass's output to out[]
// (two waves per SIMD in the register budget: with <= 256 registers hipcc selects the MFMA form that writes VGPRs, as it
//  would inside the recovery kernel's 168; with the default 512 it parks D in AGPRs and pays a v_accvgpr_read per value)
template <int MODE>
__global__ void __launch_bounds__(256, 2) k_pass(const int16_t* __restrict__ coef, uint32_t* __restrict__ out, uint64_t* __restrict__ cyc,
                                              const MfmaConsts* __restrict__ mc, int reps) {
  extern __shared__ uint32_t dyn_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* lds = dyn_lds + wave * WAVE_LDS_DWORDS;
  uint32_t* tbuf = lds + 32 * PITCH;
  const int gw = blockIdx.x * 4 + wave;
  const size_t blk = (size_t)gw * 64 + lane;
  stage<MODE>(lds, coef + blk * 64, lane);
  const uint64_t a_op = mc->a[lane];
  v16i cinit = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int h = lane >> 5;
  cinit[0] = mc->bias[4 * h]; cinit[1] = mc->bias[4 * h + 1]; cinit[2] = mc->bias[4 * h + 2]; cinit[3] = mc->bias[4 * h + 3];
  lds_fence();
  uint32_t o[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) o[i] = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    asm volatile("" ::: "memory");
    pass1<MODE>(lds, tbuf, a_op, cinit, lane, o);
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(o[i]));
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[gw] = t1 - t0;
  if (out) {
    if (MODE == MODE_M1) {
      const int n = lane & 31;
      for (int t = 0; t < 16; ++t)
        for (int v = 0; v < 4; ++v)
          out[((size_t)gw * 64 + 32 * (t & 1) + n) * 64 + (4 * h + v) * 8 + (t >> 1)] = o[t * 4 + v];
    } else {
      for (int i = 0; i < 64; ++i) out[blk * 64 + i] = o[i];
    }
  }
}

// ---- mixed kernel: 12-wave workgroups, one per CU; waves w, w + 4, w + 8 land on one SIMD (a workgroup's waves go to the
// SIMDs in cyclic order).  Waves 0-3 ("pass waves") run `reps` passes of MODE (or nothing: reps = 0), waves 4-11 run
// `nterm` x 64 of the recovery kernel's nine-instruction terms.  cyc[] = per-wave cycles.
#define TERM9(D0, D1) asm volatile( \
    "v_sub_f32 %[d], %[a], %[b]\n\tv_sub_f32 %[t], %[r], |%[d]| clamp\n\tv_mul_f32 %[t], %[t], %[t]\n\tv_mul_f32 %[d], %[d], %[t]\n\t" \
    "v_mul_f32 %[t], %[w], %[t]\n\tv_mul_f32 %[d], %[d], %[t]\n\tv_add_f32 %[n], %[n], %[d]\n\tv_mul_f32 %[d], %[t], %[t]\n\tv_add_f32 %[e], %[e], %[d]" \
    : [n] "+v"(num), [e] "+v"(den), [d] "=&v"(d_), [t] "=&v"(t_) : [a] "v"(D0), [b] "v"(D1), [w] "s"(w), [r] "s"(rs))
template <int MODE>
__global__ void __launch_bounds__(768) k_mixed(const int16_t* __restrict__ coef, uint64_t* __restrict__ cyc, float* __restrict__ sink,
                                               const MfmaConsts* __restrict__ mc, int reps, int nterm, float rs, float w) {
  extern __shared__ uint32_t dyn_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    uint32_t* lds = dyn_lds + wave * WAVE_LDS_DWORDS;
    uint32_t* tbuf = lds + 32 * PITCH;
    const size_t blk = ((size_t)blockIdx.x * 4 + wave) * 64 + lane;
    stage<MODE>(lds, coef + blk * 64, lane);
    const uint64_t a_op = mc->a[lane];
    v16i cinit = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int h = lane >> 5;
    cinit[0] = mc->bias[4 * h]; cinit[1] = mc->bias[4 * h + 1]; cinit[2] = mc->bias[4 * h + 2]; cinit[3] = mc->bias[4 * h + 3];
    lds_fence();
    uint32_t o[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) o[i] = 0;
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
      asm volatile("" ::: "memory");
      pass1<MODE>(lds, tbuf, a_op, cinit, lane, o);
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("" : "+v"(o[i]));
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) x ^= o[i];
    if (x == 0x12345678u) sink[threadIdx.x] = 1.0f;
  } else {
    float px[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) px[i] = __builtin_bit_cast(float, 0x45000000u | ((lane * 7 + i * 13) & 255));
    float num = 0.0f, den = 0.0f, d_, t_;
#pragma unroll 1
    for (int r = 0; r < nterm; ++r) {
#pragma unroll
      for (int i = 0; i < 64; ++i) TERM9(px[i & 15], px[(i + 1) & 15]);
    }
    if (num == 123.0f && den == 7.0f) sink[threadIdx.x] = num;
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 12 + wave] = t1 - t0;
}

// ---- host side
static void idct8_exact(int64_t (&v)[8]) {   // the same butterfly in exact integers (no wrap, bias 0): a pass is linear
  int64_t z1, z2, z3, z4, z5, t0, t1, t2, t3, e0, e1, e2, e3;
  z2 = v[2]; z3 = v[6]; z1 = (z2 + z3) * 4433; t2 = z1 - z3 * 15137; t3 = z1 + z2 * 6270;
  t0 = (v[0] + v[4]) * 8192; t1 = (v[0] - v[4]) * 8192;
  e0 = t0 + t3; e3 = t0 - t3; e1 = t1 + t2; e2 = t1 - t2;
  t0 = v[7]; t1 = v[5]; t2 = v[3]; t3 = v[1];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; z4 = t1 + t3; z5 = (z3 + z4) * 9633;
  t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299; z1 *= 7373; z2 *= 20995; z3 *= 16069; z4 *= 3196;
  z3 = z5 - z3; z4 = z5 - z4;
  t0 += z3 - z1; t1 += z4 - z2; t2 += z3 - z2; t3 += z4 - z1;
  v[0] = e0 + t3; v[7] = e0 - t3; v[1] = e1 + t2; v[6] = e1 - t2; v[2] = e2 + t1; v[5] = e2 - t1; v[3] = e3 + t0; v[4] = e3 - t0;
}
static void idct8_wrap(uint32_t (&v)[8]) {   // wrapping uint32, the reference's arithmetic on a 32-bit int
  uint32_t z1, z2, z3, z4, z5, t0, t1, t2, t3, e0, e1, e2, e3;
  z2 = v[2]; z3 = v[6]; z1 = (z2 + z3) * 4433u; t2 = z1 - z3 * 15137u; t3 = z1 + z2 * 6270u;
  t0 = (v[0] + v[4]) << 13; t1 = (v[0] - v[4]) << 13;
  e0 = t0 + t3; e3 = t0 - t3; e1 = t1 + t2; e2 = t1 - t2;
  t0 = v[7]; t1 = v[5]; t2 = v[3]; t3 = v[1];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; z4 = t1 + t3; z5 = (z3 + z4) * 9633u;
  t0 *= 2446u; t1 *= 16819u; t2 *= 25172u; t3 *= 12299u; z1 *= 7373u; z2 *= 20995u; z3 *= 16069u; z4 *= 3196u;
  z3 = z5 - z3; z4 = z5 - z4;
  t0 += z3 - z1; t1 += z4 - z2; t2 += z3 - z2; t3 += z4 - z1;
  v[0] = e0 + t3; v[7] = e0 - t3; v[1] = e1 + t2; v[6] = e1 - t2; v[2] = e2 + t1; v[5] = e2 - t1; v[3] = e3 + t0; v[4] = e3 - t0;
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }

static void fill_blocks(std::vector<int16_t>& c, size_t nblk, int kind) {
  c.resize(nblk * 64);
  static const int16_t ext[] = {32767, -32768, 2047, -2048, 3071, -3072, 255, -256, 127, -128, 128, -129, 1, -1, 0, 16384};
  for (size_t b = 0; b < nblk; ++b) {
    int16_t* p = &c[b * 64];
    const int k = kind == 2 ? (int)(b % 3) : kind;
    if (k == 0) { for (int i = 0; i < 64; ++i) p[i] = (int16_t)((int)(rnd() % 4096) - 2048); }            // what the format allows
    else if (k == 1) { for (int i = 0; i < 64; ++i) p[i] = (int16_t)rnd(); }                                 // any int16
    else {                                                                                                    // extremes
      const uint32_t r = rnd();
      for (int i = 0; i < 64; ++i) {
        switch (r & 3) {
          case 0: p[i] = ext[(r >> 2) & 15]; break;                                                          // constant block
          case 1: p[i] = ext[(rnd()) & 15]; break;                                                           // random extremes
          case 2: p[i] = (i == (int)((r >> 8) & 63)) ? ext[(r >> 2) & 15] : 0; break;                        // one extreme coefficient
          default: p[i] = ((i ^ (i >> 3)) & 1) ? ext[(r >> 2) & 15] : ext[(r >> 6) & 15]; break;             // checkerboard of two
        }
      }
    }
  }
}

int main(int argc, char** argv) {
  const int quick = argc > 1 && !strcmp(argv[1], "quick");
  // ---- M and its limbs
  int64_t M[8][8];
  for (int i = 0; i < 8; ++i) { int64_t v[8] = {0}; v[i] = 1; idct8_exact(v); for (int j = 0; j < 8; ++j) M[j][i] = v[j]; }
  MfmaConsts hc; memset(&hc, 0, sizeof hc);
  int maxM = 0, maxM1 = 0;
  int M0[8][8], M1[8][8];
  for (int j = 0; j < 8; ++j) {
    int64_t rs = 0;
    for (int i = 0; i < 8; ++i) {
      const int m = (int)M[j][i]; rs += m;
      M0[j][i] = ((m + 128) & 255) - 128; M1[j][i] = (m - M0[j][i]) / 256;
      if (abs(m) > maxM) maxM = abs(m);
      if (abs(M1[j][i]) > maxM1) maxM1 = abs(M1[j][i]);
      if (M0[j][i] + 256 * M1[j][i] != m || M1[j][i] < -128 || M1[j][i] > 127) { printf("limb split failed\n"); return 2; }
    }
    hc.bias[j] = (int32_t)(128 * rs + 1024);
  }
  printf("pass matrix: max |M| = %d (< 2^15), max |M1| = %d\n", maxM, maxM1);
  for (int l = 0; l < 64; ++l) {
    const int r = l & 31, g = l >> 5, cls = r >> 3, j = r & 7;
    uint64_t a = 0;
    for (int e = 0; e < 8; ++e) {
      const int i = 4 * g + (e >> 1), cl = e & 1, ml = cls - cl;
      const int val = (cls < 3 && ml == 0) ? M0[j][i] : (cls < 3 && ml == 1) ? M1[j][i] : 0;
      a |= (uint64_t)(uint8_t)(int8_t)val << (8 * e);
    }
    hc.a[l] = a;
  }
  // NB: a lane's D registers hold rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): class = reg >> 2 must be row >> 3 and
  // the output row j = (reg & 3) + 4 (lane >> 5) must be row & 7 -- it is: row = 8 class + j.
  MfmaConsts* dc; CHECK(hipMalloc(&dc, sizeof hc)); CHECK(hipMemcpy(dc, &hc, sizeof hc, hipMemcpyHostToDevice));

  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device: %s, %d CUs\n", prop.name, ncu);
  const size_t lds_wg = WAVE_LDS_DWORDS * 4 * 4;   // bytes a 4-wave workgroup needs
  CHECK(hipFuncSetAttribute((const void*)k_pass<MODE_V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute((const void*)k_pass<MODE_M1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute((const void*)k_pass<MODE_M2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute((const void*)k_pass<MODE_C>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute((const void*)k_mixed<MODE_V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute((const void*)k_mixed<MODE_M2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

  // ---- 1. exactness
  const size_t nwg = (size_t)ncu * 3, nblk = nwg * 4 * 64;
  int16_t* d_coef; uint32_t *d_out0, *d_out1, *d_out2; uint64_t* d_cyc;
  CHECK(hipMalloc(&d_coef, nblk * 64 * 2)); CHECK(hipMalloc(&d_out0, nblk * 64 * 4)); CHECK(hipMalloc(&d_out1, nblk * 64 * 4));
  CHECK(hipMalloc(&d_out2, nblk * 64 * 4)); CHECK(hipMalloc(&d_cyc, nwg * 12 * 8));
  std::vector<int16_t> hcoef; std::vector<uint32_t> o0(nblk * 64), o1(nblk * 64), o2(nblk * 64);
  size_t total = 0, bad_lin = 0, bad_v = 0, bad_m1 = 0, bad_m2 = 0;
  const int rounds = quick ? 2 : 6;
  for (int round = 0; round < rounds; ++round) {
    fill_blocks(hcoef, nblk, round % 3 == 0 ? 0 : round % 3 == 1 ? 1 : 2);
    CHECK(hipMemcpy(d_coef, hcoef.data(), nblk * 64 * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pass<MODE_V>, dim3(nwg), dim3(256), lds_wg, 0, d_coef, d_out0, d_cyc, dc, 1);
    hipLaunchKernelGGL(k_pass<MODE_M1>, dim3(nwg), dim3(256), lds_wg, 0, d_coef, d_out1, d_cyc, dc, 1);
    hipLaunchKernelGGL(k_pass<MODE_M2>, dim3(nwg), dim3(256), lds_wg, 0, d_coef, d_out2, d_cyc, dc, 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(o0.data(), d_out0, nblk * 64 * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(o1.data(), d_out1, nblk * 64 * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(o2.data(), d_out2, nblk * 64 * 4, hipMemcpyDeviceToHost));
    for (size_t b = 0; b < nblk; ++b) {
      const int16_t* c = &hcoef[b * 64];
      for (int x = 0; x < 8; ++x) {
        uint32_t col[8];
        for (int j = 0; j < 8; ++j) col[j] = (uint32_t)(int32_t)c[j * 8 + x];
        idct8_wrap(col);
        for (int j = 0; j < 8; ++j) {
          const uint32_t want = (uint32_t)((int32_t)(col[j] + 1024u) >> 11);
          // the matrix form in the int32 ring, on the CPU (linearity of the pass)
          uint32_t s = 1024u;
          for (int i = 0; i < 8; ++i) s += (uint32_t)(int32_t)M[j][i] * (uint32_t)(int32_t)c[i * 8 + x];
          if ((uint32_t)((int32_t)s >> 11) != want) ++bad_lin;
          const size_t idx = b * 64 + j * 8 + x;
          if (o0[idx] != want) ++bad_v;
          if (o1[idx] != want) { if (bad_m1 < 5) printf("  M1 mismatch blk %zu j %d x %d: got %d want %d\n", b, j, x, (int)o1[idx], (int)want); ++bad_m1; }
          if (o2[idx] != want) { if (bad_m2 < 5) printf("  M2 mismatch blk %zu j %d x %d: got %d want %d\n", b, j, x, (int)o2[idx], (int)want); ++bad_m2; }
        }
      }
    }
    total += nblk;
  }
  printf("exactness over %zu blocks (12-bit content, full int16, extremes): matrix form on CPU %zu, VALU pass %zu, MFMA pass (own layout) %zu, MFMA pass + transposition %zu mismatching values\n",
         total, bad_lin, bad_v, bad_m1, bad_m2);

  // ---- 2. cycles per pass at 1 / 2 / 3 waves per SIMD (LDS padded so that exactly W workgroups fit a CU)
  fill_blocks(hcoef, nblk, 0);
  CHECK(hipMemcpy(d_coef, hcoef.data(), nblk * 64 * 2, hipMemcpyHostToDevice));
  std::vector<uint64_t> hcyc(nwg * 12);
  const int reps = quick ? 64 : 256;
  double cal[4] = {0, 0, 0, 0};
  printf("\ncycles per pass 1 of one wave (64 blocks), mean over waves; reps %d\n", reps);
  printf("%-6s %14s %14s %14s %14s   %s\n", "waves", "C: 1 VALU", "V: VALU pass", "M1: MFMA", "M2: MFMA+back", "VALU-instruction equivalents V / M1 / M2");
  for (int W = 1; W <= 3; ++W) {
    const size_t lds = (size_t)(160 * 1024 / W) & ~(size_t)255;
    const size_t g = (size_t)ncu * W;
    double res[4];
    for (int mode = 0; mode < 4; ++mode) {
      const int m = mode == 0 ? MODE_C : mode == 1 ? MODE_V : mode == 2 ? MODE_M1 : MODE_M2;
      for (int it = 0; it < 2; ++it) {
        switch (m) {
          case MODE_C: hipLaunchKernelGGL(k_pass<MODE_C>, dim3(g), dim3(256), lds, 0, d_coef, (uint32_t*)nullptr, d_cyc, dc, reps); break;
          case MODE_V: hipLaunchKernelGGL(k_pass<MODE_V>, dim3(g), dim3(256), lds, 0, d_coef, (uint32_t*)nullptr, d_cyc, dc, reps); break;
          case MODE_M1: hipLaunchKernelGGL(k_pass<MODE_M1>, dim3(g), dim3(256), lds, 0, d_coef, (uint32_t*)nullptr, d_cyc, dc, reps); break;
          default: hipLaunchKernelGGL(k_pass<MODE_M2>, dim3(g), dim3(256), lds, 0, d_coef, (uint32_t*)nullptr, d_cyc, dc, reps); break;
        }
        CHECK(hipDeviceSynchronize());
      }
      CHECK(hipMemcpy(hcyc.data(), d_cyc, g * 4 * 8, hipMemcpyDeviceToHost));
      double s = 0; for (size_t i = 0; i < g * 4; ++i) s += (double)hcyc[i];
      res[mode] = s / (double)(g * 4) / reps;
    }
    cal[W] = res[0] / 1024.0;
    printf("%-6d %14.2f %14.0f %14.0f %14.0f   %.0f / %.0f / %.0f\n", W, cal[W], res[1], res[2], res[3], res[1] / cal[W], res[2] / cal[W], res[3] / cal[W]);
  }

  // ---- 3. one pass wave next to two term waves per SIMD
  printf("\nmixed: per SIMD one pass wave (P passes) + two waves of the 9-instruction term stream (T x 64 terms each), 12-wave workgroups, one per CU\n");
  float* d_sink; CHECK(hipMalloc(&d_sink, 768 * 4));
  const int T = quick ? 256 : 1024, P = quick ? 64 : 256;
  const size_t lds12 = lds_wg;   // only waves 0-3 use LDS
  double base_term = 0;
  printf("%-34s %16s %16s %22s\n", "pass waves run", "pass-wave cycles", "term-wave cycles", "term cycles lost per pass");
  for (int cfg = 0; cfg < 3; ++cfg) {
    for (int it = 0; it < 2; ++it) {
      if (cfg == 0) hipLaunchKernelGGL(k_mixed<MODE_V>, dim3(ncu), dim3(768), lds12, 0, d_coef, d_cyc, d_sink, dc, 0, T, 0.05f, 0.37f);
      else if (cfg == 1) hipLaunchKernelGGL(k_mixed<MODE_V>, dim3(ncu), dim3(768), lds12, 0, d_coef, d_cyc, d_sink, dc, P, T, 0.05f, 0.37f);
      else hipLaunchKernelGGL(k_mixed<MODE_M2>, dim3(ncu), dim3(768), lds12, 0, d_coef, d_cyc, d_sink, dc, P, T, 0.05f, 0.37f);
      CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(hcyc.data(), d_cyc, (size_t)ncu * 12 * 8, hipMemcpyDeviceToHost));
    double sp = 0, st = 0;
    for (int b = 0; b < ncu; ++b) for (int w = 0; w < 12; ++w) (w < 4 ? sp : st) += (double)hcyc[b * 12 + w];
    sp /= ncu * 4.0; st /= ncu * 8.0;
    if (cfg == 0) base_term = st;
    printf("%-34s %16.0f %16.0f %22.1f\n", cfg == 0 ? "nothing (term waves alone)" : cfg == 1 ? "V  (VALU pass 1)" : "M2 (MFMA pass 1 + way back)",
           sp, st, cfg == 0 ? 0.0 : (st - base_term) / P);
  }
  printf("(term stream: %d terms x 9 instructions per wave = %d VALU instructions; per pass-wave pass the two term waves of the SIMD lose the last column each)\n",
         T * 64, T * 64 * 9);
  return (bad_lin || bad_v || bad_m1 || bad_m2) ? 1 : 0;
}
