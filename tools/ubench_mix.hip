// ubench_mix.hip -- round 4: issue-rate and exactness probes behind the "pixel differences as packed f16, consumed by
// v_fma_mix_f32" form of the recovery kernel's term (8 instructions instead of 9 for an interior term, 72 instead of 96
// VGPRs of per-block state), plus the questions the mid-size launch regime raised:
//   * what does the s_nop hipcc puts behind every inline-asm statement cost a lone wave / two waves per SIMD?
//   * does a wave64 whose upper 32 lanes are masked off issue VALU any faster?
//   * rates of the instructions the refresh would need to produce packed-f16 differences.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_mix.hip -o build/ubench_mix
// Output: G wave-instructions/s over the chip and the implied cycles per wave-instruction per SIMD at 2.4 GHz
// (1024 SIMDs) for W = 1..4 waves per SIMD, then the exactness check (must print 0 mismatches).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define ITER 2048

// ---- the shipped 9-op term, serial (as in tools/ubench_imul.hip) --------------------------------------------------
#define TERM9(D, T, A, B, TAIL)                        \
  "v_sub_f32 " D ", " A ", " B "\n"                    \
  "v_sub_f32 " T ", %[r], |" D "| clamp\n"             \
  "v_mul_f32 " T ", " T ", " T "\n"                    \
  "v_mul_f32 " D ", " D ", " T "\n"                    \
  "v_mul_f32 " T ", %[w], " T "\n"                     \
  "v_mul_f32 " D ", " D ", " T "\n"                    \
  "v_add_f32 %[num], %[num], " D "\n"                  \
  "v_mul_f32 " D ", " T ", " T "\n"                    \
  "v_add_f32 %[den], %[den], " D "\n" TAIL
// ---- the candidate 8-op term: the difference comes as one f16 half of a VGPR -------------------------------------
//   u = clamp01(R - |d|)  : v_fma_mix_f32  (-|d.h|) * 1.0 + R, clamp       (exact: one rounding of an exact value)
//   t = u * u
//   x = fl(d * t)         : v_fma_mix_f32  d.h * t + (-0)                   (one rounding = v_mul_f32)
//   y = w * t ; num += x * y ; den += y * y
#define TERM8(D, T, P, H, TAIL)                                                              \
  "v_fma_mix_f32 " T ", -|" P "|, 1.0, %[r] op_sel:[" H ",0,0] op_sel_hi:[1,0,0] clamp\n"    \
  "v_mul_f32 " T ", " T ", " T "\n"                                                          \
  "v_fma_mix_f32 " D ", " P ", " T ", neg(0) op_sel:[" H ",0,0] op_sel_hi:[1,0,0]\n"         \
  "v_mul_f32 " T ", %[w], " T "\n"                                                           \
  "v_mul_f32 " D ", " D ", " T "\n"                                                          \
  "v_add_f32 %[num], %[num], " D "\n"                                                        \
  "v_mul_f32 " D ", " T ", " T "\n"                                                          \
  "v_add_f32 %[den], %[den], " D "\n" TAIL

#define OPS9 : [num] "+v"(num), [den] "+v"(den), [d0] "=&v"(d0), [t0] "=&v"(t0) \
             : [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [r] "s"(r), [w] "s"(w)
#define X16_9(TAIL) \
  TERM9("%[d0]", "%[t0]", "%[p0]", "%[p1]", TAIL) TERM9("%[d0]", "%[t0]", "%[p1]", "%[p2]", TAIL) TERM9("%[d0]", "%[t0]", "%[p2]", "%[p3]", TAIL) TERM9("%[d0]", "%[t0]", "%[p0]", "%[p3]", TAIL) \
  TERM9("%[d0]", "%[t0]", "%[p0]", "%[p1]", TAIL) TERM9("%[d0]", "%[t0]", "%[p1]", "%[p2]", TAIL) TERM9("%[d0]", "%[t0]", "%[p2]", "%[p3]", TAIL) TERM9("%[d0]", "%[t0]", "%[p0]", "%[p3]", TAIL) \
  TERM9("%[d0]", "%[t0]", "%[p0]", "%[p1]", TAIL) TERM9("%[d0]", "%[t0]", "%[p1]", "%[p2]", TAIL) TERM9("%[d0]", "%[t0]", "%[p2]", "%[p3]", TAIL) TERM9("%[d0]", "%[t0]", "%[p0]", "%[p3]", TAIL) \
  TERM9("%[d0]", "%[t0]", "%[p0]", "%[p1]", TAIL) TERM9("%[d0]", "%[t0]", "%[p1]", "%[p2]", TAIL) TERM9("%[d0]", "%[t0]", "%[p2]", "%[p3]", TAIL) TERM9("%[d0]", "%[t0]", "%[p0]", "%[p3]", TAIL)
#define X16_8(TAIL) \
  TERM8("%[d0]", "%[t0]", "%[p0]", "0", TAIL) TERM8("%[d0]", "%[t0]", "%[p0]", "1", TAIL) TERM8("%[d0]", "%[t0]", "%[p1]", "0", TAIL) TERM8("%[d0]", "%[t0]", "%[p1]", "1", TAIL) \
  TERM8("%[d0]", "%[t0]", "%[p2]", "0", TAIL) TERM8("%[d0]", "%[t0]", "%[p2]", "1", TAIL) TERM8("%[d0]", "%[t0]", "%[p3]", "0", TAIL) TERM8("%[d0]", "%[t0]", "%[p3]", "1", TAIL) \
  TERM8("%[d0]", "%[t0]", "%[p0]", "0", TAIL) TERM8("%[d0]", "%[t0]", "%[p0]", "1", TAIL) TERM8("%[d0]", "%[t0]", "%[p1]", "0", TAIL) TERM8("%[d0]", "%[t0]", "%[p1]", "1", TAIL) \
  TERM8("%[d0]", "%[t0]", "%[p2]", "0", TAIL) TERM8("%[d0]", "%[t0]", "%[p2]", "1", TAIL) TERM8("%[d0]", "%[t0]", "%[p3]", "0", TAIL) TERM8("%[d0]", "%[t0]", "%[p3]", "1", TAIL)

__global__ void k_term9(float* out, float r, float w) {
  float num = 0, den = 0, d0, t0;
  float p0 = threadIdx.x * 1e-3f, p1 = p0 + 1e-3f, p2 = p0 + 2e-3f, p3 = p0 + 3e-3f;
  for (int i = 0; i < ITER; ++i) asm volatile(X16_9("") OPS9);
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;
}
__global__ void k_term9_nop(float* out, float r, float w) {           // an s_nop 0 behind every term, as hipcc emits
  float num = 0, den = 0, d0, t0;
  float p0 = threadIdx.x * 1e-3f, p1 = p0 + 1e-3f, p2 = p0 + 2e-3f, p3 = p0 + 3e-3f;
  for (int i = 0; i < ITER; ++i) asm volatile(X16_9("s_nop 0\n") OPS9);
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;
}
__global__ void k_term8(float* out, float r, float w) {
  float num = 0, den = 0, d0, t0;
  uint32_t p0 = 0x1c001800u + threadIdx.x, p1 = p0 + 0x00010001u, p2 = p1 + 0x00010001u, p3 = p2 + 0x00010001u;   // small f16 pairs
  for (int i = 0; i < ITER; ++i) asm volatile(X16_8("") OPS9);
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;
}
__global__ void k_term8_nop(float* out, float r, float w) {
  float num = 0, den = 0, d0, t0;
  uint32_t p0 = 0x1c001800u + threadIdx.x, p1 = p0 + 0x00010001u, p2 = p1 + 0x00010001u, p3 = p2 + 0x00010001u;
  for (int i = 0; i < ITER; ++i) asm volatile(X16_8("s_nop 0\n") OPS9);
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;
}
// the 9-op term with only the lower 32 lanes active (EXEC[63:32] = 0)
__global__ void k_term9_half(float* out, float r, float w) {
  float num = 0, den = 0, d0, t0;
  float p0 = threadIdx.x * 1e-3f, p1 = p0 + 1e-3f, p2 = p0 + 2e-3f, p3 = p0 + 3e-3f;
  if (threadIdx.x < 32)
    for (int i = 0; i < ITER; ++i) asm volatile(X16_9("") OPS9);
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;
}

// ---- single instructions: 8 independent chains / one dependent chain, 32 instructions per loop trip ---------------
#define DEF_KERNEL(NAME, INS)                                                                              \
  __global__ void NAME##_ind(int* out, int a, float sr) {                                                  \
    int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    for (int i = 0; i < ITER; ++i) {                                                                       \
      asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                   INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"(sr)); \
    }                                                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                   \
  }                                                                                                        \
  __global__ void NAME##_dep(int* out, int a, float sr) {                                                  \
    int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    for (int i = 0; i < ITER; ++i) {                                                                       \
      asm volatile(INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) \
                   INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) \
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"(sr)); \
    }                                                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                   \
  }
#define I_MULF(n)     "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_MIX_U(n)    "v_fma_mix_f32 %" #n ", -|%" #n "|, 1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp\n"
#define I_MIX_X(n)    "v_fma_mix_f32 %" #n ", %8, %" #n ", neg(0) op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define I_PKADD16(n)  "v_pk_add_f16 %" #n ", %" #n ", %8 neg_lo:[0,1] neg_hi:[0,1]\n"
#define I_PKADD16S(n) "v_pk_add_f16 %" #n ", %" #n ", %8 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
#define I_CVTPK(n)    "v_cvt_pkrtz_f16_f32 %" #n ", %" #n ", %8\n"
#define I_MED3(n)     "v_med3_i32 %" #n ", %" #n ", 0, %8\n"
#define I_LSHLOR(n)   "v_lshl_or_b32 %" #n ", %" #n ", 16, %8\n"
#define I_ANDOR(n)    "v_and_or_b32 %" #n ", %" #n ", %8, %8\n"
#define I_ALIGN(n)    "v_alignbit_b32 %" #n ", %8, %" #n ", 18\n"
#define I_LSHR(n)     "v_lshrrev_b32 %" #n ", 18, %" #n "\n"
#define I_BFE(n)      "v_bfe_u32 %" #n ", %" #n ", 18, 8\n"
#define I_PKSUBI16(n) "v_pk_sub_i16 %" #n ", %" #n ", %8\n"
#define I_CVTF16I(n)  "v_cvt_f16_i16 %" #n ", %" #n "\n"
#define I_PKMULF16(n) "v_pk_mul_f16 %" #n ", %" #n ", %8\n"
#define I_SUBSDWA(n)  "v_sub_f32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
DEF_KERNEL(k_mulf, I_MULF)
DEF_KERNEL(k_mix_u, I_MIX_U)
DEF_KERNEL(k_mix_x, I_MIX_X)
DEF_KERNEL(k_pkadd16, I_PKADD16)
DEF_KERNEL(k_pkadd16s, I_PKADD16S)
DEF_KERNEL(k_cvtpk, I_CVTPK)
DEF_KERNEL(k_med3, I_MED3)
DEF_KERNEL(k_lshlor, I_LSHLOR)
DEF_KERNEL(k_andor, I_ANDOR)
DEF_KERNEL(k_align, I_ALIGN)
DEF_KERNEL(k_lshr, I_LSHR)
DEF_KERNEL(k_bfe, I_BFE)
DEF_KERNEL(k_pksubi16, I_PKSUBI16)
DEF_KERNEL(k_cvtf16i, I_CVTF16I)
DEF_KERNEL(k_pkmulf16, I_PKMULF16)

// ---- exactness: the 8-op form against the 9-op form, bit for bit ---------------------------------------------------
// One lane per (difference d, range R) pair; every lane accumulates NW weights.  d = -255..255 as d * 2^-12, both as
// the difference of two "magic" f32 pixels (what the shipped kernel holds) and as an f16 half (lo and hi tested).
__global__ void k_check(const float* __restrict__ wts, int nw, const int* __restrict__ ranges, int nr, unsigned* bad, unsigned* checked) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int di = gid % 511, ri = gid / 511;
  if (ri >= nr) return;
  const int d = di - 255;
  const float Rs = (float)ranges[ri] * 0.000244140625f;
  const int pa = d >= 0 ? d : 0, pb = d >= 0 ? 0 : -d;                      // pixels with pa - pb = d
  const float fa = __builtin_bit_cast(float, 0x45000000u | (uint32_t)pa), fb = __builtin_bit_cast(float, 0x45000000u | (uint32_t)pb);
  const _Float16 hd = (_Float16)((float)d * 0.000244140625f);               // exact: |d| <= 255 < 2^11
  const uint32_t hbits = (uint32_t)__builtin_bit_cast(unsigned short, hd);
  const uint32_t plo = hbits | 0x3c000000u, phi = (hbits << 16) | 0x00003c00u;   // the other half holds 1.0 (must not matter)
  float n9 = 0, e9 = 0, n8l = 0, e8l = 0, n8h = 0, e8h = 0, d0, t0;
  for (int k = 0; k < nw; ++k) {
    const float w = wts[k];
    asm volatile(TERM9("%[d0]", "%[t0]", "%[a]", "%[b]", "")
                 : [num] "+v"(n9), [den] "+v"(e9), [d0] "=&v"(d0), [t0] "=&v"(t0) : [a] "v"(fa), [b] "v"(fb), [r] "s"(Rs), [w] "s"(w));
    asm volatile(TERM8("%[d0]", "%[t0]", "%[p]", "0", "")
                 : [num] "+v"(n8l), [den] "+v"(e8l), [d0] "=&v"(d0), [t0] "=&v"(t0) : [p] "v"(plo), [r] "s"(Rs), [w] "s"(w));
    asm volatile(TERM8("%[d0]", "%[t0]", "%[p]", "1", "")
                 : [num] "+v"(n8h), [den] "+v"(e8h), [d0] "=&v"(d0), [t0] "=&v"(t0) : [p] "v"(phi), [r] "s"(Rs), [w] "s"(w));
    const bool ok = __builtin_bit_cast(uint32_t, n9) == __builtin_bit_cast(uint32_t, n8l) && __builtin_bit_cast(uint32_t, e9) == __builtin_bit_cast(uint32_t, e8l) &&
                    __builtin_bit_cast(uint32_t, n9) == __builtin_bit_cast(uint32_t, n8h) && __builtin_bit_cast(uint32_t, e9) == __builtin_bit_cast(uint32_t, e8h);
    if (!ok) atomicAdd(bad, 1u);
  }
  atomicAdd(checked, (unsigned)nw);
}
// packed-f16 differences of "magic" f16 pixels (0x3400 | p = 0.25 + p * 2^-12): exact, equal to (pa - pb) * 2^-12
__global__ void k_check_pk(unsigned* bad) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int pa = gid & 255, pb = (gid >> 8) & 255;
  const uint32_t A = (0x3400u | pa) | ((0x3400u | pb) << 16), B = (0x3400u | pb) | ((0x3400u | pa) << 16);
  uint32_t D;
  asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(D) : "v"(A), "v"(B));
  const _Float16 want_lo = (_Float16)((float)(pa - pb) * 0.000244140625f), want_hi = (_Float16)((float)(pb - pa) * 0.000244140625f);
  const uint32_t W = (uint32_t)__builtin_bit_cast(unsigned short, want_lo) | ((uint32_t)__builtin_bit_cast(unsigned short, want_hi) << 16);
  // +0 / -0: a difference of equal pixels must be +0 in both halves (x = d * t stays +-0 either way; checked for the record)
  if (D != W) atomicAdd(bad, 1u);
}

// ---------------------------------------------------------------------------
static void* g_out;
template <class F>
static void measure(const char* name, int instr_per_trip, F launch) {
  printf("%-22s", name);
  for (int w = 1; w <= 4; ++w) {
    const int blocks = 1024 * w;                       // one 64-lane wave per block, W waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(blocks);                                    // warm-up
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      launch(blocks);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double winstr = (double)blocks * ITER * instr_per_trip;    // wave-instructions
    const double rate = winstr / (best * 1e-3);
    printf("  W=%d %7.1f G/s (%4.2f cyc, %6.3f ms)", w, rate * 1e-9, 2.4e9 * 1024 / rate, best);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  printf("\n");
}
#define INT_PAIR(K, N) \
  measure(#K " ind", N, [](int b) { hipLaunchKernelGGL(K##_ind, dim3(b), dim3(64), 0, 0, (int*)g_out, 0x3c003c00, 0.25f); }); \
  measure(#K " dep", N, [](int b) { hipLaunchKernelGGL(K##_dep, dim3(b), dim3(64), 0, 0, (int*)g_out, 0x3c003c00, 0.25f); });

int main() {
  hipMalloc(&g_out, (size_t)4096 * 64 * 4);
  printf("wave-instructions/s over the chip (G/s), cycles per wave-instruction per SIMD @2.4 GHz, launch ms; W = waves per SIMD\n");
  printf("(term rows count 9 resp. 8 VALU instructions per term; the s_nop is not counted: compare the ms)\n");
  measure("term9", 16 * 9, [](int b) { hipLaunchKernelGGL(k_term9, dim3(b), dim3(64), 0, 0, (float*)g_out, 0.5f, 0.25f); });
  measure("term9 + s_nop", 16 * 9, [](int b) { hipLaunchKernelGGL(k_term9_nop, dim3(b), dim3(64), 0, 0, (float*)g_out, 0.5f, 0.25f); });
  measure("term8 (fma_mix)", 16 * 8, [](int b) { hipLaunchKernelGGL(k_term8, dim3(b), dim3(64), 0, 0, (float*)g_out, 0.5f, 0.25f); });
  measure("term8 + s_nop", 16 * 8, [](int b) { hipLaunchKernelGGL(k_term8_nop, dim3(b), dim3(64), 0, 0, (float*)g_out, 0.5f, 0.25f); });
  measure("term9, 32 lanes", 16 * 9, [](int b) { hipLaunchKernelGGL(k_term9_half, dim3(b), dim3(64), 0, 0, (float*)g_out, 0.5f, 0.25f); });
  INT_PAIR(k_mulf, 32) INT_PAIR(k_mix_u, 32) INT_PAIR(k_mix_x, 32) INT_PAIR(k_pkadd16, 32) INT_PAIR(k_pkadd16s, 32)
  INT_PAIR(k_cvtpk, 32) INT_PAIR(k_med3, 32) INT_PAIR(k_lshlor, 32) INT_PAIR(k_andor, 32) INT_PAIR(k_align, 32)
  INT_PAIR(k_lshr, 32) INT_PAIR(k_bfe, 32) INT_PAIR(k_pksubi16, 32) INT_PAIR(k_cvtf16i, 32) INT_PAIR(k_pkmulf16, 32)

  // ---- exactness
  const int NW = 4096, NR = 512;
  float* hw = (float*)malloc(NW * 4); int* hr = (int*)malloc(NR * 4);
  srand(12345);
  for (int k = 0; k < NW; ++k) {                       // weights: signed, magnitudes 2^-29 .. 4 like the tables', some exact zeros
    const int e = -29 + rand() % 32;
    float m = 1.0f + (float)(rand() & 0x7fffff) / 8388608.0f;
    hw[k] = (rand() & 1 ? -1.0f : 1.0f) * ldexpf(m, e);
    if (k % 97 == 0) hw[k] = 0.0f;
  }
  for (int k = 0; k < NR; ++k) hr[k] = k < 300 ? 2 * (k + 1) : 2 * (1 + rand() % 2047);   // R = 2q, q = 1..2047
  float* dw; int* dr; unsigned *dbad, *dchk;
  hipMalloc(&dw, NW * 4); hipMalloc(&dr, NR * 4); hipMalloc(&dbad, 4); hipMalloc(&dchk, 4);
  hipMemcpy(dw, hw, NW * 4, hipMemcpyHostToDevice); hipMemcpy(dr, hr, NR * 4, hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 4); hipMemset(dchk, 0, 4);
  hipLaunchKernelGGL(k_check, dim3((511 * NR + 255) / 256), dim3(256), 0, 0, dw, NW, dr, NR, dbad, dchk);
  unsigned bad = 0, chk = 0;
  hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost); hipMemcpy(&chk, dchk, 4, hipMemcpyDeviceToHost);
  printf("exactness: 8-op (fma_mix, f16 difference in the low / high half) vs 9-op running sums: %u mismatching steps of %u x 511 x %d\n", bad, (unsigned)NW, NR);
  (void)chk;
  hipMemset(dbad, 0, 4);
  hipLaunchKernelGGL(k_check_pk, dim3(256), dim3(256), 0, 0, dbad);
  hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
  printf("exactness: v_pk_add_f16 differences of magic-f16 pixels (65536 pairs): %u mismatches\n", bad);
  return 0;
}
