// ubench_imul.hip -- gfx950 issue-rate probes behind the IDCT-refresh question of round 2:
// which integer multiply forms are cheaper than v_mul_i32_i24, and what does a wave gain from
// instruction-level parallelism (two interleaved term chains) at 1..4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_imul.hip -o build/ubench_imul
// Output: G wave-instructions/s summed over the chip and the implied cycles per wave-instruction
// per SIMD at 2.4 GHz (1024 SIMDs), for W = 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ITER 2048

// 8 registers; IND: 8 independent chains (one instruction each, 4 rounds = 32 instructions per
// loop trip); DEP: one dependent chain of 32 instructions
#define DEF_KERNEL(NAME, INS)                                                                              \
  __global__ void NAME##_ind(int* out, int a) {                                                            \
    int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    for (int i = 0; i < ITER; ++i) {                                                                       \
      asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                   INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a)); \
    }                                                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                   \
  }                                                                                                        \
  __global__ void NAME##_dep(int* out, int a) {                                                            \
    int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    for (int i = 0; i < ITER; ++i) {                                                                       \
      asm volatile(INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) \
                   INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) \
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a)); \
    }                                                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                   \
  }

#define I_ADD(n)      "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_MUL24(n)    "v_mul_i32_i24 %" #n ", %" #n ", %8\n"
#define I_MAD24(n)    "v_mad_i32_i24 %" #n ", %" #n ", %8, %" #n "\n"
#define I_MADI16(n)   "v_mad_i32_i16 %" #n ", %" #n ", %8, %" #n "\n"
#define I_MADU16(n)   "v_mad_u32_u16 %" #n ", %" #n ", %8, %" #n "\n"
#define I_PKMUL16(n)  "v_pk_mul_lo_u16 %" #n ", %" #n ", %8\n"
#define I_PKMAD16(n)  "v_pk_mad_i16 %" #n ", %" #n ", %8, %" #n "\n"
#define I_MULLO32(n)  "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_DOT2(n)     "v_dot2_i32_i16 %" #n ", %" #n ", %8, %" #n "\n"
#define I_MULF(n)     "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_LSHLADD(n)  "v_lshl_add_u32 %" #n ", %" #n ", 3, %8\n"
#define I_CVT(n)      "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define I_MULHI24(n)  "v_mul_hi_i32_i24 %" #n ", %" #n ", %8\n"
#define I_PERM(n)     "v_perm_b32 %" #n ", %" #n ", %8, %" #n "\n"
#define I_MADF16(n)   "v_pk_fma_f16 %" #n ", %" #n ", %8, %" #n "\n"

DEF_KERNEL(k_add, I_ADD)
DEF_KERNEL(k_mul24, I_MUL24)
DEF_KERNEL(k_mad24, I_MAD24)
DEF_KERNEL(k_madi16, I_MADI16)
DEF_KERNEL(k_madu16, I_MADU16)
DEF_KERNEL(k_pkmul16, I_PKMUL16)
DEF_KERNEL(k_pkmad16, I_PKMAD16)
DEF_KERNEL(k_mullo32, I_MULLO32)
DEF_KERNEL(k_dot2, I_DOT2)
DEF_KERNEL(k_mulf, I_MULF)
DEF_KERNEL(k_lshladd, I_LSHLADD)
DEF_KERNEL(k_cvt, I_CVT)
DEF_KERNEL(k_mulhi24, I_MULHI24)
DEF_KERNEL(k_perm, I_PERM)

// The recovery kernel's 9-op term (qs_kernels.hip: QS_TERM) as the compiler emits it -- one strictly
// serial chain per term -- and the same work with TWO terms interleaved (every instruction's
// producer is two slots back).  16 terms per loop trip in both.
#define TERM1(D, T, A, B)                              \
  "v_sub_f32 " D ", " A ", " B "\n"                    \
  "v_sub_f32 " T ", %[r], |" D "| clamp\n"             \
  "v_mul_f32 " T ", " T ", " T "\n"                    \
  "v_mul_f32 " D ", " D ", " T "\n"                    \
  "v_mul_f32 " T ", %[w], " T "\n"                     \
  "v_mul_f32 " D ", " D ", " T "\n"                    \
  "v_add_f32 %[num], %[num], " D "\n"                  \
  "v_mul_f32 " D ", " T ", " T "\n"                    \
  "v_add_f32 %[den], %[den], " D "\n"
#define TERM2(D0, T0, A0, B0, D1, T1, A1, B1)          \
  "v_sub_f32 " D0 ", " A0 ", " B0 "\n"                 \
  "v_sub_f32 " D1 ", " A1 ", " B1 "\n"                 \
  "v_sub_f32 " T0 ", %[r], |" D0 "| clamp\n"           \
  "v_sub_f32 " T1 ", %[r], |" D1 "| clamp\n"           \
  "v_mul_f32 " T0 ", " T0 ", " T0 "\n"                 \
  "v_mul_f32 " T1 ", " T1 ", " T1 "\n"                 \
  "v_mul_f32 " D0 ", " D0 ", " T0 "\n"                 \
  "v_mul_f32 " D1 ", " D1 ", " T1 "\n"                 \
  "v_mul_f32 " T0 ", %[w], " T0 "\n"                   \
  "v_mul_f32 " T1 ", %[w], " T1 "\n"                   \
  "v_mul_f32 " D0 ", " D0 ", " T0 "\n"                 \
  "v_mul_f32 " D1 ", " D1 ", " T1 "\n"                 \
  "v_add_f32 %[num], %[num], " D0 "\n"                 \
  "v_mul_f32 " D0 ", " T0 ", " T0 "\n"                 \
  "v_add_f32 %[num], %[num], " D1 "\n"                 \
  "v_mul_f32 " D1 ", " T1 ", " T1 "\n"                 \
  "v_add_f32 %[den], %[den], " D0 "\n"                 \
  "v_add_f32 %[den], %[den], " D1 "\n"
#define TERM_OPS : [num] "+v"(num), [den] "+v"(den), [d0] "=&v"(d0), [t0] "=&v"(t0), [d1] "=&v"(d1), [t1] "=&v"(t1) \
                 : [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [r] "s"(r), [w] "s"(w)

__global__ void k_term_serial(float* out, float r, float w) {
  float num = 0, den = 0, d0, t0, d1, t1;
  float p0 = threadIdx.x * 1e-3f, p1 = p0 + 1e-3f, p2 = p0 + 2e-3f, p3 = p0 + 3e-3f;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(
      TERM1("%[d0]", "%[t0]", "%[p0]", "%[p1]") TERM1("%[d0]", "%[t0]", "%[p1]", "%[p2]") TERM1("%[d0]", "%[t0]", "%[p2]", "%[p3]") TERM1("%[d0]", "%[t0]", "%[p0]", "%[p3]")
      TERM1("%[d0]", "%[t0]", "%[p0]", "%[p1]") TERM1("%[d0]", "%[t0]", "%[p1]", "%[p2]") TERM1("%[d0]", "%[t0]", "%[p2]", "%[p3]") TERM1("%[d0]", "%[t0]", "%[p0]", "%[p3]")
      TERM1("%[d0]", "%[t0]", "%[p0]", "%[p1]") TERM1("%[d0]", "%[t0]", "%[p1]", "%[p2]") TERM1("%[d0]", "%[t0]", "%[p2]", "%[p3]") TERM1("%[d0]", "%[t0]", "%[p0]", "%[p3]")
      TERM1("%[d0]", "%[t0]", "%[p0]", "%[p1]") TERM1("%[d0]", "%[t0]", "%[p1]", "%[p2]") TERM1("%[d0]", "%[t0]", "%[p2]", "%[p3]") TERM1("%[d0]", "%[t0]", "%[p0]", "%[p3]")
      TERM_OPS);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den + d1 + t1;
}
#define PAIR(A0, B0, A1, B1) TERM2("%[d0]", "%[t0]", A0, B0, "%[d1]", "%[t1]", A1, B1)
__global__ void k_term_pair(float* out, float r, float w) {
  float num = 0, den = 0, d0, t0, d1, t1;
  float p0 = threadIdx.x * 1e-3f, p1 = p0 + 1e-3f, p2 = p0 + 2e-3f, p3 = p0 + 3e-3f;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(
      PAIR("%[p0]", "%[p1]", "%[p1]", "%[p2]") PAIR("%[p2]", "%[p3]", "%[p0]", "%[p3]")
      PAIR("%[p0]", "%[p1]", "%[p1]", "%[p2]") PAIR("%[p2]", "%[p3]", "%[p0]", "%[p3]")
      PAIR("%[p0]", "%[p1]", "%[p1]", "%[p2]") PAIR("%[p2]", "%[p3]", "%[p0]", "%[p3]")
      PAIR("%[p0]", "%[p1]", "%[p1]", "%[p2]") PAIR("%[p2]", "%[p3]", "%[p0]", "%[p3]")
      TERM_OPS);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;
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
    // cycles per wave-instruction per SIMD at 2.4 GHz, if the chip were evenly loaded
    printf("  W=%d %7.1f G/s (%4.2f cyc)", w, rate * 1e-9, 2.4e9 * 1024 / rate);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  printf("\n");
}
#define INT_PAIR(K, N) \
  measure(#K " ind", N, [](int b) { hipLaunchKernelGGL(K##_ind, dim3(b), dim3(64), 0, 0, (int*)g_out, 3); }); \
  measure(#K " dep", N, [](int b) { hipLaunchKernelGGL(K##_dep, dim3(b), dim3(64), 0, 0, (int*)g_out, 3); });

int main() {
  hipMalloc(&g_out, (size_t)4096 * 64 * 4);
  printf("wave-instructions/s over the chip (G/s) and cycles per wave-instruction per SIMD @2.4 GHz; W = waves per SIMD\n");
  INT_PAIR(k_add, 32) INT_PAIR(k_mulf, 32) INT_PAIR(k_lshladd, 32) INT_PAIR(k_cvt, 32) INT_PAIR(k_perm, 32)
  INT_PAIR(k_mul24, 32) INT_PAIR(k_mad24, 32) INT_PAIR(k_mulhi24, 32) INT_PAIR(k_mullo32, 32)
  INT_PAIR(k_madi16, 32) INT_PAIR(k_madu16, 32) INT_PAIR(k_pkmul16, 32) INT_PAIR(k_pkmad16, 32) INT_PAIR(k_dot2, 32)
  measure("term serial (9 op)", 16 * 9, [](int b) { hipLaunchKernelGGL(k_term_serial, dim3(b), dim3(64), 0, 0, (float*)g_out, 0.5f, 0.25f); });
  measure("term pairs  (9 op)", 16 * 9, [](int b) { hipLaunchKernelGGL(k_term_pair, dim3(b), dim3(64), 0, 0, (float*)g_out, 0.5f, 0.25f); });
  return 0;
}
