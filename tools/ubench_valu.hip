// ubench_valu.hip -- VALU issue-rate probes for gfx950 (decides the layout of
// the recovery kernel: packed vs unpacked f32, 24-bit vs 32-bit integer mul).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o build/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define ITER 4096

// 8 independent chains, 2 instructions each per step
__global__ void k_mul_add(float* out, float a) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(
      "v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n"
      "v_mul_f32 %1, %1, %8\n v_add_f32 %1, %1, %8\n"
      "v_mul_f32 %2, %2, %8\n v_add_f32 %2, %2, %8\n"
      "v_mul_f32 %3, %3, %8\n v_add_f32 %3, %3, %8\n"
      "v_mul_f32 %4, %4, %8\n v_add_f32 %4, %4, %8\n"
      "v_mul_f32 %5, %5, %8\n v_add_f32 %5, %5, %8\n"
      "v_mul_f32 %6, %6, %8\n v_add_f32 %6, %6, %8\n"
      "v_mul_f32 %7, %7, %8\n v_add_f32 %7, %7, %8\n"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void k_pk_mul_add(float* out, float a) {
  v2f x0 = {(float)threadIdx.x, 1}, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  v2f aa = {a, a};
  for (int i = 0; i < ITER; ++i) {
    asm volatile(
      "v_pk_mul_f32 %0, %0, %8\n v_pk_add_f32 %0, %0, %8\n"
      "v_pk_mul_f32 %1, %1, %8\n v_pk_add_f32 %1, %1, %8\n"
      "v_pk_mul_f32 %2, %2, %8\n v_pk_add_f32 %2, %2, %8\n"
      "v_pk_mul_f32 %3, %3, %8\n v_pk_add_f32 %3, %3, %8\n"
      "v_pk_mul_f32 %4, %4, %8\n v_pk_add_f32 %4, %4, %8\n"
      "v_pk_mul_f32 %5, %5, %8\n v_pk_add_f32 %5, %5, %8\n"
      "v_pk_mul_f32 %6, %6, %8\n v_pk_add_f32 %6, %6, %8\n"
      "v_pk_mul_f32 %7, %7, %8\n v_pk_add_f32 %7, %7, %8\n"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(aa));
  }
  v2f s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

__global__ void k_fma(float* out, float a) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(
      "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n"
      "v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %1, %1, %8, %8\n"
      "v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %2, %2, %8, %8\n"
      "v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
      "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %4, %4, %8, %8\n"
      "v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %5, %5, %8, %8\n"
      "v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %6, %6, %8, %8\n"
      "v_fma_f32 %7, %7, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

// sgpr operand + abs modifier + max (the VOP3 forms the kernel uses)
__global__ void k_sub_abs_max(float* out, float a) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(
      "v_sub_f32 %0, %8, |%0|\n v_max_f32 %0, 0, %0\n"
      "v_sub_f32 %1, %8, |%1|\n v_max_f32 %1, 0, %1\n"
      "v_sub_f32 %2, %8, |%2|\n v_max_f32 %2, 0, %2\n"
      "v_sub_f32 %3, %8, |%3|\n v_max_f32 %3, 0, %3\n"
      "v_sub_f32 %4, %8, |%4|\n v_max_f32 %4, 0, %4\n"
      "v_sub_f32 %5, %8, |%5|\n v_max_f32 %5, 0, %5\n"
      "v_sub_f32 %6, %8, |%6|\n v_max_f32 %6, 0, %6\n"
      "v_sub_f32 %7, %8, |%7|\n v_max_f32 %7, 0, %7\n"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void k_mul_lo(int* out, int a) {
  int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(
      "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
      "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
      "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
      "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void k_mul_i24(int* out, int a) {
  int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(
      "v_mul_i32_i24 %0, %0, %8\n v_mul_i32_i24 %1, %1, %8\n v_mul_i32_i24 %2, %2, %8\n v_mul_i32_i24 %3, %3, %8\n"
      "v_mul_i32_i24 %4, %4, %8\n v_mul_i32_i24 %5, %5, %8\n v_mul_i32_i24 %6, %6, %8\n v_mul_i32_i24 %7, %7, %8\n"
      "v_mad_i32_i24 %0, %0, %8, %8\n v_mad_i32_i24 %1, %1, %8, %8\n v_mad_i32_i24 %2, %2, %8, %8\n v_mad_i32_i24 %3, %3, %8, %8\n"
      "v_mad_i32_i24 %4, %4, %8, %8\n v_mad_i32_i24 %5, %5, %8, %8\n v_mad_i32_i24 %6, %6, %8, %8\n v_mad_i32_i24 %7, %7, %8, %8\n"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

// dependent chain: one accumulator, add after add (latency probe)
__global__ void k_dep_add(float* out, float a) {
  float x0 = threadIdx.x;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(REP8("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n") : "+v"(x0) : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0;
}


// fma with one inline-constant operand (2 VGPR reads)
__global__ void k_fma_c(float* out, float a) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(
      "v_fma_f32 %0, %0, 1.0, %8\n v_fma_f32 %0, %0, 1.0, %8\n"
      "v_fma_f32 %1, %1, 1.0, %8\n v_fma_f32 %1, %1, 1.0, %8\n"
      "v_fma_f32 %2, %2, 1.0, %8\n v_fma_f32 %2, %2, 1.0, %8\n"
      "v_fma_f32 %3, %3, 1.0, %8\n v_fma_f32 %3, %3, 1.0, %8\n"
      "v_fma_f32 %4, %4, 1.0, %8\n v_fma_f32 %4, %4, 1.0, %8\n"
      "v_fma_f32 %5, %5, 1.0, %8\n v_fma_f32 %5, %5, 1.0, %8\n"
      "v_fma_f32 %6, %6, 1.0, %8\n v_fma_f32 %6, %6, 1.0, %8\n"
      "v_fma_f32 %7, %7, 1.0, %8\n v_fma_f32 %7, %7, 1.0, %8\n"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// v_fma_mix_f32 reading f16 halves
__global__ void k_fma_mix(float* out, float a) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(
      "v_fma_mix_f32 %0, %0, 1.0, %8 op_sel_hi:[1,0,1]\n v_fma_mix_f32 %0, %0, 1.0, %8 op_sel_hi:[1,0,1]\n"
      "v_fma_mix_f32 %1, %1, 1.0, %8 op_sel_hi:[1,0,1]\n v_fma_mix_f32 %1, %1, 1.0, %8 op_sel_hi:[1,0,1]\n"
      "v_fma_mix_f32 %2, %2, 1.0, %8 op_sel_hi:[1,0,1]\n v_fma_mix_f32 %2, %2, 1.0, %8 op_sel_hi:[1,0,1]\n"
      "v_fma_mix_f32 %3, %3, 1.0, %8 op_sel_hi:[1,0,1]\n v_fma_mix_f32 %3, %3, 1.0, %8 op_sel_hi:[1,0,1]\n"
      "v_fma_mix_f32 %4, %4, 1.0, %8 op_sel_hi:[1,0,1]\n v_fma_mix_f32 %4, %4, 1.0, %8 op_sel_hi:[1,0,1]\n"
      "v_fma_mix_f32 %5, %5, 1.0, %8 op_sel_hi:[1,0,1]\n v_fma_mix_f32 %5, %5, 1.0, %8 op_sel_hi:[1,0,1]\n"
      "v_fma_mix_f32 %6, %6, 1.0, %8 op_sel_hi:[1,0,1]\n v_fma_mix_f32 %6, %6, 1.0, %8 op_sel_hi:[1,0,1]\n"
      "v_fma_mix_f32 %7, %7, 1.0, %8 op_sel_hi:[1,0,1]\n v_fma_mix_f32 %7, %7, 1.0, %8 op_sel_hi:[1,0,1]\n"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// the 9-op term of the recovery kernel, 4 independent terms, as the kernel issues it
__global__ void k_term9(float* out, float a) {
  float p0 = threadIdx.x * 1e-3f, p1 = p0 + 1e-3f, p2 = p0 + 2e-3f, p3 = p0 + 3e-3f, p4 = p0 + 4e-3f;
  float num = 0, den = 0;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(
      "v_sub_f32 v20, %2, %3\n v_sub_f32 v21, %3, %4\n v_sub_f32 v22, %4, %5\n v_sub_f32 v23, %5, %6\n"
      "v_sub_f32 v24, %7, |v20| clamp\n v_sub_f32 v25, %7, |v21| clamp\n v_sub_f32 v26, %7, |v22| clamp\n v_sub_f32 v27, %7, |v23| clamp\n"
      "v_mul_f32 v24, v24, v24\n v_mul_f32 v25, v25, v25\n v_mul_f32 v26, v26, v26\n v_mul_f32 v27, v27, v27\n"
      "v_mul_f32 v20, v20, v24\n v_mul_f32 v21, v21, v25\n v_mul_f32 v22, v22, v26\n v_mul_f32 v23, v23, v27\n"
      "v_mul_f32 v24, %7, v24\n v_mul_f32 v25, %7, v25\n v_mul_f32 v26, %7, v26\n v_mul_f32 v27, %7, v27\n"
      "v_mul_f32 v20, v20, v24\n v_mul_f32 v21, v21, v25\n v_mul_f32 v22, v22, v26\n v_mul_f32 v23, v23, v27\n"
      "v_mul_f32 v24, v24, v24\n v_mul_f32 v25, v25, v25\n v_mul_f32 v26, v26, v26\n v_mul_f32 v27, v27, v27\n"
      "v_add_f32 %0, %0, v20\n v_add_f32 %1, %1, v24\n v_add_f32 %0, %0, v21\n v_add_f32 %1, %1, v25\n"
      "v_add_f32 %0, %0, v22\n v_add_f32 %1, %1, v26\n v_add_f32 %0, %0, v23\n v_add_f32 %1, %1, v27\n"
      : "+v"(num), "+v"(den) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "s"(a)
      : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;
}


// the same 9-op term, ONE term at a time, strictly dependent (what hipcc emits
// under register pressure): 4 terms back to back
#define TERM_SERIAL(PA, PB) \
      "v_sub_f32 v20, " PA ", " PB "\n v_sub_f32 v24, %7, |v20| clamp\n v_mul_f32 v24, v24, v24\n" \
      "v_mul_f32 v20, v20, v24\n v_mul_f32 v24, %7, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\n" \
      "v_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
__global__ void k_term9_serial(float* out, float a) {
  float p0 = threadIdx.x * 1e-3f, p1 = p0 + 1e-3f, p2 = p0 + 2e-3f, p3 = p0 + 3e-3f, p4 = p0 + 4e-3f;
  float num = 0, den = 0;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(TERM_SERIAL("%2", "%3") TERM_SERIAL("%3", "%4") TERM_SERIAL("%4", "%5") TERM_SERIAL("%5", "%6")
      : "+v"(num), "+v"(den) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "s"(a) : "v20", "v24");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;
}
// two terms interleaved
#define TERM_X2(PA, PB, PC, PD) \
      "v_sub_f32 v20, " PA ", " PB "\n v_sub_f32 v21, " PC ", " PD "\n" \
      "v_sub_f32 v24, %7, |v20| clamp\n v_sub_f32 v25, %7, |v21| clamp\n" \
      "v_mul_f32 v24, v24, v24\n v_mul_f32 v25, v25, v25\n" \
      "v_mul_f32 v20, v20, v24\n v_mul_f32 v21, v21, v25\n" \
      "v_mul_f32 v24, %7, v24\n v_mul_f32 v25, %7, v25\n" \
      "v_mul_f32 v20, v20, v24\n v_mul_f32 v21, v21, v25\n" \
      "v_mul_f32 v24, v24, v24\n v_mul_f32 v25, v25, v25\n" \
      "v_add_f32 %0, %0, v20\n v_add_f32 %1, %1, v24\n v_add_f32 %0, %0, v21\n v_add_f32 %1, %1, v25\n"
__global__ void k_term9_x2(float* out, float a) {
  float p0 = threadIdx.x * 1e-3f, p1 = p0 + 1e-3f, p2 = p0 + 2e-3f, p3 = p0 + 3e-3f, p4 = p0 + 4e-3f;
  float num = 0, den = 0;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(TERM_X2("%2", "%3", "%3", "%4") TERM_X2("%4", "%5", "%5", "%6")
      : "+v"(num), "+v"(den) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "s"(a) : "v20", "v21", "v24", "v25");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;
}


// instruction-footprint probe: the same serial term stream, but N distinct
// terms of straight-line code per loop trip (the real kernel has 144-242)
#define T4 TERM_SERIAL("%2", "%3") TERM_SERIAL("%3", "%4") TERM_SERIAL("%4", "%5") TERM_SERIAL("%5", "%6")
#define T16 T4 T4 T4 T4
#define T64 T16 T16 T16 T16
#define T256 T64 T64 T64 T64
template <int N>
__global__ void k_term9_big(float* out, float a) {
  float p0 = threadIdx.x * 1e-3f, p1 = p0 + 1e-3f, p2 = p0 + 2e-3f, p3 = p0 + 3e-3f, p4 = p0 + 4e-3f;
  float num = 0, den = 0;
  for (int i = 0; i < ITER * 4 / N; ++i) {
    if (N == 64) asm volatile(T64 : "+v"(num), "+v"(den) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "s"(a) : "v20", "v24");
    if (N == 256) asm volatile(T256 : "+v"(num), "+v"(den) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "s"(a) : "v20", "v24");
    if (N == 1024) { asm volatile(T256 T256 : "+v"(num), "+v"(den) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "s"(a) : "v20", "v24");
                     asm volatile(T256 T256 : "+v"(num), "+v"(den) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "s"(a) : "v20", "v24"); }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;
}


__global__ void k_term9_wide(float* out, float a) {
  float num = 0, den = 0; float p = ((threadIdx.x * 2654435761u) >> 20) * (1.0f / 4096.0f / 16.0f);
  asm volatile(
      "v_mul_f32 v64, 0.828125, %3\n"
      "v_mul_f32 v65, 0.453125, %3\n"
      "v_mul_f32 v66, 0.078125, %3\n"
      "v_mul_f32 v67, 0.656250, %3\n"
      "v_mul_f32 v68, 0.281250, %3\n"
      "v_mul_f32 v69, 0.859375, %3\n"
      "v_mul_f32 v70, 0.484375, %3\n"
      "v_mul_f32 v71, 0.109375, %3\n"
      "v_mul_f32 v72, 0.687500, %3\n"
      "v_mul_f32 v73, 0.312500, %3\n"
      "v_mul_f32 v74, 0.890625, %3\n"
      "v_mul_f32 v75, 0.515625, %3\n"
      "v_mul_f32 v76, 0.140625, %3\n"
      "v_mul_f32 v77, 0.718750, %3\n"
      "v_mul_f32 v78, 0.343750, %3\n"
      "v_mul_f32 v79, 0.921875, %3\n"
      "v_mul_f32 v80, 0.546875, %3\n"
      "v_mul_f32 v81, 0.171875, %3\n"
      "v_mul_f32 v82, 0.750000, %3\n"
      "v_mul_f32 v83, 0.375000, %3\n"
      "v_mul_f32 v84, 0.953125, %3\n"
      "v_mul_f32 v85, 0.578125, %3\n"
      "v_mul_f32 v86, 0.203125, %3\n"
      "v_mul_f32 v87, 0.781250, %3\n"
      "v_mul_f32 v88, 0.406250, %3\n"
      "v_mul_f32 v89, 0.984375, %3\n"
      "v_mul_f32 v90, 0.609375, %3\n"
      "v_mul_f32 v91, 0.234375, %3\n"
      "v_mul_f32 v92, 0.812500, %3\n"
      "v_mul_f32 v93, 0.437500, %3\n"
      "v_mul_f32 v94, 0.062500, %3\n"
      "v_mul_f32 v95, 0.640625, %3\n"
      "v_mul_f32 v96, 0.265625, %3\n"
      "v_mul_f32 v97, 0.843750, %3\n"
      "v_mul_f32 v98, 0.468750, %3\n"
      "v_mul_f32 v99, 0.093750, %3\n"
      "v_mul_f32 v100, 0.671875, %3\n"
      "v_mul_f32 v101, 0.296875, %3\n"
      "v_mul_f32 v102, 0.875000, %3\n"
      "v_mul_f32 v103, 0.500000, %3\n"
      "v_mul_f32 v104, 0.125000, %3\n"
      "v_mul_f32 v105, 0.703125, %3\n"
      "v_mul_f32 v106, 0.328125, %3\n"
      "v_mul_f32 v107, 0.906250, %3\n"
      "v_mul_f32 v108, 0.531250, %3\n"
      "v_mul_f32 v109, 0.156250, %3\n"
      "v_mul_f32 v110, 0.734375, %3\n"
      "v_mul_f32 v111, 0.359375, %3\n"
      "v_mul_f32 v112, 0.937500, %3\n"
      "v_mul_f32 v113, 0.562500, %3\n"
      "v_mul_f32 v114, 0.187500, %3\n"
      "v_mul_f32 v115, 0.765625, %3\n"
      "v_mul_f32 v116, 0.390625, %3\n"
      "v_mul_f32 v117, 0.968750, %3\n"
      "v_mul_f32 v118, 0.593750, %3\n"
      "v_mul_f32 v119, 0.218750, %3\n"
      "v_mul_f32 v120, 0.796875, %3\n"
      "v_mul_f32 v121, 0.421875, %3\n"
      "v_mul_f32 v122, 0.046875, %3\n"
      "v_mul_f32 v123, 0.625000, %3\n"
      "v_mul_f32 v124, 0.250000, %3\n"
      "v_mul_f32 v125, 0.828125, %3\n"
      "v_mul_f32 v126, 0.453125, %3\n"
      "v_mul_f32 v127, 0.078125, %3\n"
      : : "s"(a), "s"(a), "s"(a), "v"(p) : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
  for (int i = 0; i < ITER * 4 / 112; ++i) {
    asm volatile(
      "v_sub_f32 v20, v64, v65\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v65, v66\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v66, v67\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v67, v68\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v68, v69\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v69, v70\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v70, v71\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v72, v73\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v73, v74\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v74, v75\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v75, v76\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v76, v77\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v77, v78\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v78, v79\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v80, v81\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v81, v82\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v82, v83\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v83, v84\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v84, v85\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v85, v86\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v86, v87\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v88, v89\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v89, v90\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v90, v91\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v91, v92\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v92, v93\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v93, v94\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v94, v95\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v96, v97\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v97, v98\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v98, v99\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v99, v100\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v100, v101\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v101, v102\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v102, v103\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v104, v105\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v105, v106\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v106, v107\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v107, v108\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v108, v109\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v109, v110\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v110, v111\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v112, v113\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v113, v114\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v114, v115\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v115, v116\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v116, v117\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v117, v118\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v118, v119\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v120, v121\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v121, v122\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v122, v123\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v123, v124\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v124, v125\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v125, v126\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v126, v127\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v64, v72\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v65, v73\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v66, v74\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v67, v75\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v68, v76\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v69, v77\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v70, v78\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v71, v79\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v72, v80\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v73, v81\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v74, v82\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v75, v83\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v76, v84\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v77, v85\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v78, v86\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v79, v87\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v80, v88\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v81, v89\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v82, v90\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v83, v91\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v84, v92\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v85, v93\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v86, v94\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v87, v95\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v88, v96\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v89, v97\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v90, v98\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v91, v99\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v92, v100\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v93, v101\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v94, v102\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v95, v103\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v96, v104\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v97, v105\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v98, v106\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v99, v107\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v100, v108\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v101, v109\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v102, v110\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v103, v111\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v104, v112\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v105, v113\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v106, v114\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v107, v115\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v108, v116\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v109, v117\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v110, v118\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v111, v119\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v112, v120\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v113, v121\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v114, v122\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v115, v123\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v116, v124\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v117, v125\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v118, v126\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v119, v127\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, %2, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      : "+v"(num), "+v"(den) : "s"(a) : "v20", "v24", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;
}


__global__ void k_term9_smem(float* out, const float* tab, float a) {
  float num = 0, den = 0; float p = ((threadIdx.x * 2654435761u) >> 20) * (1.0f / 4096.0f / 16.0f);
  asm volatile(
      "v_mul_f32 v64, 0.046875, %0\n"
      "v_mul_f32 v65, 0.625000, %0\n"
      "v_mul_f32 v66, 0.250000, %0\n"
      "v_mul_f32 v67, 0.828125, %0\n"
      "v_mul_f32 v68, 0.453125, %0\n"
      "v_mul_f32 v69, 0.078125, %0\n"
      "v_mul_f32 v70, 0.656250, %0\n"
      "v_mul_f32 v71, 0.281250, %0\n"
      "v_mul_f32 v72, 0.859375, %0\n"
      "v_mul_f32 v73, 0.484375, %0\n"
      "v_mul_f32 v74, 0.109375, %0\n"
      "v_mul_f32 v75, 0.687500, %0\n"
      "v_mul_f32 v76, 0.312500, %0\n"
      "v_mul_f32 v77, 0.890625, %0\n"
      "v_mul_f32 v78, 0.515625, %0\n"
      "v_mul_f32 v79, 0.140625, %0\n"
      "v_mul_f32 v80, 0.718750, %0\n"
      "v_mul_f32 v81, 0.343750, %0\n"
      "v_mul_f32 v82, 0.921875, %0\n"
      "v_mul_f32 v83, 0.546875, %0\n"
      "v_mul_f32 v84, 0.171875, %0\n"
      "v_mul_f32 v85, 0.750000, %0\n"
      "v_mul_f32 v86, 0.375000, %0\n"
      "v_mul_f32 v87, 0.953125, %0\n"
      "v_mul_f32 v88, 0.578125, %0\n"
      "v_mul_f32 v89, 0.203125, %0\n"
      "v_mul_f32 v90, 0.781250, %0\n"
      "v_mul_f32 v91, 0.406250, %0\n"
      "v_mul_f32 v92, 0.984375, %0\n"
      "v_mul_f32 v93, 0.609375, %0\n"
      "v_mul_f32 v94, 0.234375, %0\n"
      "v_mul_f32 v95, 0.812500, %0\n"
      "v_mul_f32 v96, 0.437500, %0\n"
      "v_mul_f32 v97, 0.062500, %0\n"
      "v_mul_f32 v98, 0.640625, %0\n"
      "v_mul_f32 v99, 0.265625, %0\n"
      "v_mul_f32 v100, 0.843750, %0\n"
      "v_mul_f32 v101, 0.468750, %0\n"
      "v_mul_f32 v102, 0.093750, %0\n"
      "v_mul_f32 v103, 0.671875, %0\n"
      "v_mul_f32 v104, 0.296875, %0\n"
      "v_mul_f32 v105, 0.875000, %0\n"
      "v_mul_f32 v106, 0.500000, %0\n"
      "v_mul_f32 v107, 0.125000, %0\n"
      "v_mul_f32 v108, 0.703125, %0\n"
      "v_mul_f32 v109, 0.328125, %0\n"
      "v_mul_f32 v110, 0.906250, %0\n"
      "v_mul_f32 v111, 0.531250, %0\n"
      "v_mul_f32 v112, 0.156250, %0\n"
      "v_mul_f32 v113, 0.734375, %0\n"
      "v_mul_f32 v114, 0.359375, %0\n"
      "v_mul_f32 v115, 0.937500, %0\n"
      "v_mul_f32 v116, 0.562500, %0\n"
      "v_mul_f32 v117, 0.187500, %0\n"
      "v_mul_f32 v118, 0.765625, %0\n"
      "v_mul_f32 v119, 0.390625, %0\n"
      "v_mul_f32 v120, 0.968750, %0\n"
      "v_mul_f32 v121, 0.593750, %0\n"
      "v_mul_f32 v122, 0.218750, %0\n"
      "v_mul_f32 v123, 0.796875, %0\n"
      "v_mul_f32 v124, 0.421875, %0\n"
      "v_mul_f32 v125, 0.046875, %0\n"
      "v_mul_f32 v126, 0.625000, %0\n"
      "v_mul_f32 v127, 0.250000, %0\n"
      : : "v"(p) : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
  unsigned off = (blockIdx.x & 63) * 1088u;
  asm volatile("s_load_dwordx16 s[36:51], %0, %1\n s_load_dwordx16 s[52:67], %0, %1" : : "s"(tab), "s"(off) : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
  for (int i = 0; i < ITER * 4 / 112; ++i) {
    asm volatile(
      "s_waitcnt lgkmcnt(0)\n s_load_dwordx16 s[52:67], %3, %4\n"
      "v_sub_f32 v20, v64, v65\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s36, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v65, v66\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s37, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v66, v67\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s38, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v67, v68\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s39, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v68, v69\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s40, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v69, v70\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s41, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v70, v71\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s42, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v72, v73\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s43, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v73, v74\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s44, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v74, v75\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s45, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v75, v76\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s46, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v76, v77\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s47, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v77, v78\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s48, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v78, v79\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s49, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v80, v81\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s50, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v81, v82\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s51, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "s_waitcnt lgkmcnt(0)\n s_load_dwordx16 s[36:51], %3, %4\n"
      "v_sub_f32 v20, v82, v83\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s52, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v83, v84\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s53, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v84, v85\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s54, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v85, v86\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s55, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v86, v87\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s56, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v88, v89\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s57, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v89, v90\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s58, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v90, v91\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s59, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v91, v92\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s60, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v92, v93\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s61, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v93, v94\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s62, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v94, v95\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s63, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v96, v97\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s64, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v97, v98\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s65, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v98, v99\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s66, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v99, v100\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s67, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "s_waitcnt lgkmcnt(0)\n s_load_dwordx16 s[52:67], %3, %4\n"
      "v_sub_f32 v20, v100, v101\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s36, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v101, v102\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s37, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v102, v103\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s38, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v104, v105\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s39, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v105, v106\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s40, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v106, v107\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s41, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v107, v108\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s42, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v108, v109\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s43, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v109, v110\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s44, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v110, v111\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s45, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v112, v113\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s46, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v113, v114\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s47, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v114, v115\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s48, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v115, v116\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s49, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v116, v117\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s50, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v117, v118\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s51, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "s_waitcnt lgkmcnt(0)\n s_load_dwordx16 s[36:51], %3, %4\n"
      "v_sub_f32 v20, v118, v119\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s52, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v120, v121\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s53, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v121, v122\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s54, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v122, v123\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s55, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v123, v124\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s56, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v124, v125\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s57, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v125, v126\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s58, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v126, v127\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s59, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v64, v72\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s60, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v65, v73\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s61, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v66, v74\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s62, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v67, v75\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s63, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v68, v76\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s64, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v69, v77\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s65, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v70, v78\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s66, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v71, v79\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s67, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "s_waitcnt lgkmcnt(0)\n s_load_dwordx16 s[52:67], %3, %4\n"
      "v_sub_f32 v20, v72, v80\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s36, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v73, v81\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s37, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v74, v82\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s38, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v75, v83\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s39, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v76, v84\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s40, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v77, v85\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s41, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v78, v86\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s42, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v79, v87\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s43, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v80, v88\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s44, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v81, v89\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s45, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v82, v90\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s46, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v83, v91\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s47, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v84, v92\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s48, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v85, v93\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s49, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v86, v94\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s50, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v87, v95\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s51, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "s_waitcnt lgkmcnt(0)\n s_load_dwordx16 s[36:51], %3, %4\n"
      "v_sub_f32 v20, v88, v96\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s52, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v89, v97\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s53, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v90, v98\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s54, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v91, v99\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s55, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v92, v100\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s56, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v93, v101\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s57, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v94, v102\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s58, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v95, v103\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s59, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v96, v104\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s60, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v97, v105\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s61, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v98, v106\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s62, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v99, v107\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s63, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v100, v108\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s64, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v101, v109\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s65, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v102, v110\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s66, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v103, v111\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s67, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "s_waitcnt lgkmcnt(0)\n s_load_dwordx16 s[52:67], %3, %4\n"
      "v_sub_f32 v20, v104, v112\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s36, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v105, v113\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s37, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v106, v114\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s38, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v107, v115\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s39, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v108, v116\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s40, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v109, v117\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s41, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v110, v118\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s42, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v111, v119\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s43, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v112, v120\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s44, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v113, v121\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s45, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v114, v122\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s46, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v115, v123\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s47, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v116, v124\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s48, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v117, v125\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s49, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v118, v126\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s50, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v119, v127\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s51, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      : "+v"(num), "+v"(den) : "s"(a), "s"(tab), "s"(off) : "v20", "v24", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
    off = (off + 1088u) & 65535u;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;
}

__global__ void k_term9_sgprs(float* out, const float* tab, float a) {
  float num = 0, den = 0; float p = ((threadIdx.x * 2654435761u) >> 20) * (1.0f / 4096.0f / 16.0f);
  asm volatile(
      "v_mul_f32 v64, 0.046875, %0\n"
      "v_mul_f32 v65, 0.625000, %0\n"
      "v_mul_f32 v66, 0.250000, %0\n"
      "v_mul_f32 v67, 0.828125, %0\n"
      "v_mul_f32 v68, 0.453125, %0\n"
      "v_mul_f32 v69, 0.078125, %0\n"
      "v_mul_f32 v70, 0.656250, %0\n"
      "v_mul_f32 v71, 0.281250, %0\n"
      "v_mul_f32 v72, 0.859375, %0\n"
      "v_mul_f32 v73, 0.484375, %0\n"
      "v_mul_f32 v74, 0.109375, %0\n"
      "v_mul_f32 v75, 0.687500, %0\n"
      "v_mul_f32 v76, 0.312500, %0\n"
      "v_mul_f32 v77, 0.890625, %0\n"
      "v_mul_f32 v78, 0.515625, %0\n"
      "v_mul_f32 v79, 0.140625, %0\n"
      "v_mul_f32 v80, 0.718750, %0\n"
      "v_mul_f32 v81, 0.343750, %0\n"
      "v_mul_f32 v82, 0.921875, %0\n"
      "v_mul_f32 v83, 0.546875, %0\n"
      "v_mul_f32 v84, 0.171875, %0\n"
      "v_mul_f32 v85, 0.750000, %0\n"
      "v_mul_f32 v86, 0.375000, %0\n"
      "v_mul_f32 v87, 0.953125, %0\n"
      "v_mul_f32 v88, 0.578125, %0\n"
      "v_mul_f32 v89, 0.203125, %0\n"
      "v_mul_f32 v90, 0.781250, %0\n"
      "v_mul_f32 v91, 0.406250, %0\n"
      "v_mul_f32 v92, 0.984375, %0\n"
      "v_mul_f32 v93, 0.609375, %0\n"
      "v_mul_f32 v94, 0.234375, %0\n"
      "v_mul_f32 v95, 0.812500, %0\n"
      "v_mul_f32 v96, 0.437500, %0\n"
      "v_mul_f32 v97, 0.062500, %0\n"
      "v_mul_f32 v98, 0.640625, %0\n"
      "v_mul_f32 v99, 0.265625, %0\n"
      "v_mul_f32 v100, 0.843750, %0\n"
      "v_mul_f32 v101, 0.468750, %0\n"
      "v_mul_f32 v102, 0.093750, %0\n"
      "v_mul_f32 v103, 0.671875, %0\n"
      "v_mul_f32 v104, 0.296875, %0\n"
      "v_mul_f32 v105, 0.875000, %0\n"
      "v_mul_f32 v106, 0.500000, %0\n"
      "v_mul_f32 v107, 0.125000, %0\n"
      "v_mul_f32 v108, 0.703125, %0\n"
      "v_mul_f32 v109, 0.328125, %0\n"
      "v_mul_f32 v110, 0.906250, %0\n"
      "v_mul_f32 v111, 0.531250, %0\n"
      "v_mul_f32 v112, 0.156250, %0\n"
      "v_mul_f32 v113, 0.734375, %0\n"
      "v_mul_f32 v114, 0.359375, %0\n"
      "v_mul_f32 v115, 0.937500, %0\n"
      "v_mul_f32 v116, 0.562500, %0\n"
      "v_mul_f32 v117, 0.187500, %0\n"
      "v_mul_f32 v118, 0.765625, %0\n"
      "v_mul_f32 v119, 0.390625, %0\n"
      "v_mul_f32 v120, 0.968750, %0\n"
      "v_mul_f32 v121, 0.593750, %0\n"
      "v_mul_f32 v122, 0.218750, %0\n"
      "v_mul_f32 v123, 0.796875, %0\n"
      "v_mul_f32 v124, 0.421875, %0\n"
      "v_mul_f32 v125, 0.046875, %0\n"
      "v_mul_f32 v126, 0.625000, %0\n"
      "v_mul_f32 v127, 0.250000, %0\n"
      : : "v"(p) : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
  unsigned off = (blockIdx.x & 63) * 1088u;
  asm volatile("s_load_dwordx16 s[36:51], %0, %1\n s_load_dwordx16 s[52:67], %0, %1" : : "s"(tab), "s"(off) : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
  for (int i = 0; i < ITER * 4 / 112; ++i) {
    asm volatile(
      "v_sub_f32 v20, v64, v65\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s36, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v65, v66\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s37, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v66, v67\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s38, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v67, v68\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s39, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v68, v69\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s40, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v69, v70\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s41, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v70, v71\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s42, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v72, v73\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s43, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v73, v74\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s44, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v74, v75\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s45, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v75, v76\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s46, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v76, v77\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s47, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v77, v78\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s48, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v78, v79\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s49, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v80, v81\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s50, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v81, v82\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s51, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v82, v83\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s52, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v83, v84\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s53, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v84, v85\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s54, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v85, v86\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s55, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v86, v87\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s56, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v88, v89\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s57, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v89, v90\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s58, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v90, v91\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s59, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v91, v92\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s60, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v92, v93\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s61, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v93, v94\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s62, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v94, v95\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s63, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v96, v97\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s64, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v97, v98\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s65, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v98, v99\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s66, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v99, v100\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s67, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v100, v101\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s36, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v101, v102\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s37, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v102, v103\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s38, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v104, v105\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s39, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v105, v106\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s40, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v106, v107\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s41, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v107, v108\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s42, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v108, v109\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s43, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v109, v110\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s44, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v110, v111\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s45, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v112, v113\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s46, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v113, v114\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s47, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v114, v115\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s48, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v115, v116\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s49, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v116, v117\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s50, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v117, v118\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s51, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v118, v119\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s52, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v120, v121\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s53, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v121, v122\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s54, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v122, v123\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s55, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v123, v124\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s56, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v124, v125\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s57, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v125, v126\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s58, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v126, v127\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s59, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v64, v72\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s60, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v65, v73\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s61, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v66, v74\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s62, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v67, v75\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s63, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v68, v76\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s64, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v69, v77\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s65, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v70, v78\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s66, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v71, v79\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s67, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v72, v80\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s36, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v73, v81\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s37, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v74, v82\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s38, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v75, v83\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s39, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v76, v84\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s40, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v77, v85\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s41, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v78, v86\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s42, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v79, v87\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s43, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v80, v88\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s44, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v81, v89\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s45, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v82, v90\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s46, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v83, v91\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s47, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v84, v92\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s48, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v85, v93\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s49, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v86, v94\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s50, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v87, v95\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s51, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v88, v96\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s52, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v89, v97\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s53, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v90, v98\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s54, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v91, v99\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s55, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v92, v100\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s56, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v93, v101\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s57, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v94, v102\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s58, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v95, v103\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s59, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v96, v104\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s60, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v97, v105\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s61, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v98, v106\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s62, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v99, v107\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s63, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v100, v108\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s64, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v101, v109\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s65, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v102, v110\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s66, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v103, v111\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s67, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v104, v112\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s36, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v105, v113\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s37, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v106, v114\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s38, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v107, v115\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s39, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v108, v116\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s40, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v109, v117\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s41, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v110, v118\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s42, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v111, v119\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s43, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v112, v120\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s44, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v113, v121\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s45, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v114, v122\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s46, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v115, v123\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s47, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v116, v124\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s48, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v117, v125\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s49, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v118, v126\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s50, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      "v_sub_f32 v20, v119, v127\n v_sub_f32 v24, %2, |v20| clamp\n v_mul_f32 v24, v24, v24\nv_mul_f32 v20, v20, v24\n v_mul_f32 v24, s51, v24\n v_mul_f32 v20, v20, v24\n v_add_f32 %0, %0, v20\nv_mul_f32 v20, v24, v24\n v_add_f32 %1, %1, v20\n"
      : "+v"(num), "+v"(den) : "s"(a), "s"(tab), "s"(off) : "v20", "v24", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
    off = (off + 1088u) & 65535u;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;
}


__global__ void k_dot2c(int* out, int a) {
  int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITER; ++i) {
    asm volatile(
      "v_dot2c_i32_i16 %0, 0xd6301151, %8\n v_dot2c_i32_i16 %1, 0xd6301151, %8\n v_dot2c_i32_i16 %2, 0xd6301151, %8\n v_dot2c_i32_i16 %3, 0xd6301151, %8\n"
      "v_dot2c_i32_i16 %4, 0xd6301151, %8\n v_dot2c_i32_i16 %5, 0xd6301151, %8\n v_dot2c_i32_i16 %6, 0xd6301151, %8\n v_dot2c_i32_i16 %7, 0xd6301151, %8\n"
      "v_dot2c_i32_i16 %0, 0x12341151, %8\n v_dot2c_i32_i16 %1, 0x12341151, %8\n v_dot2c_i32_i16 %2, 0x12341151, %8\n v_dot2c_i32_i16 %3, 0x12341151, %8\n"
      "v_dot2c_i32_i16 %4, 0x12341151, %8\n v_dot2c_i32_i16 %5, 0x12341151, %8\n v_dot2c_i32_i16 %6, 0x12341151, %8\n v_dot2c_i32_i16 %7, 0x12341151, %8\n"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <class K, class T>
static void run(const char* name, K kern, T* out, T arg, double lane_ops_per_thread_iter, int pk) {
  static const int kWps[] = {1, 2, 3, 4, 8};
  for (int wps : kWps) {
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, arg);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, grid, block, 0, 0, out, arg);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double instr = (double)grid.x * 256 * ITER * lane_ops_per_thread_iter;
    printf("%-14s waves/SIMD=%d  %.3f ms  %.2f T lane-instr/s  (%.2f T lane-ops/s)\n", name, wps, ms,
           instr / ms * 1e-9, instr * pk / ms * 1e-9);
  }
}

int main(int argc, char** argv) {
  const bool term_only = argc > 1 && argv[1][0] == 't';
  float* out; hipMalloc(&out, sizeof(float) * 256 * 8 * 256);
  float* tab; hipMalloc(&tab, 1 << 20); hipMemset(tab, 0x3c, 1 << 20);
  for (int wps = 1; wps <= 4; wps *= 2) {
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_term9_smem, grid, block, 0, 0, out, tab, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_term9_smem, grid, block, 0, 0, out, tab, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    int trips = ITER * 4 / 112;
    double instr = (double)grid.x * 256 * trips * 112 * 9;
    printf("term9+smem stream waves/SIMD=%d  %.3f ms  %.2f T lane-instr/s\n", wps, ms, instr / ms * 1e-9);
  }
  for (int wps = 1; wps <= 4; wps *= 2) {
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_term9_sgprs, grid, block, 0, 0, out, tab, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_term9_sgprs, grid, block, 0, 0, out, tab, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    int trips = ITER * 4 / 112;
    double instr = (double)grid.x * 256 * trips * 112 * 9;
    printf("term9 16 sgprs no load waves/SIMD=%d  %.3f ms  %.2f T lane-instr/s\n", wps, ms, instr / ms * 1e-9);
  }
  if (term_only) {
    run("term9 x4", k_term9, out, 0.5f, 36, 1);
    run("term9 serial", k_term9_serial, out, 0.5f, 36, 1);
    run("term9 x2", k_term9_x2, out, 0.5f, 36, 1);
    run("mul+add", k_mul_add, out, 1.0001f, 16, 1);
    run("dep_add", k_dep_add, out, 1.0001f, 16, 1);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'f') {        // instruction-footprint probe only
    run("term9 body64", k_term9_big<64>, out, 0.5f, 36, 1);
    run("term9 body256", k_term9_big<256>, out, 0.5f, 36, 1);
    run("term9 body1024", k_term9_big<1024>, out, 0.5f, 36, 1);
    return 0;
  }
  run("mul+add", k_mul_add, out, 1.0001f, 16, 1);
  run("pk_mul+pk_add", k_pk_mul_add, out, 1.0001f, 16, 2);
  run("fma", k_fma, out, 1.0001f, 16, 1);
  run("fma_const", k_fma_c, out, 1.0001f, 16, 1);
  run("fma_mix", k_fma_mix, out, 1.0001f, 16, 1);
  run("term9 x4", k_term9, out, 0.5f, 36, 1);
  run("term9 serial", k_term9_serial, out, 0.5f, 36, 1);
  run("term9 x2", k_term9_x2, out, 0.5f, 36, 1);
  run("term9 wide64px", k_term9_wide, out, 0.5f, 36, 1);
  run("term9 body64", k_term9_big<64>, out, 0.5f, 36, 1);
  run("term9 body256", k_term9_big<256>, out, 0.5f, 36, 1);
  run("term9 body1024", k_term9_big<1024>, out, 0.5f, 36, 1);
  run("sub|abs|+max", k_sub_abs_max, out, 1.0001f, 16, 1);
  run("mul_lo_u32", k_mul_lo, (int*)out, 3, 16, 1);
  run("dot2c_i32_i16", k_dot2c, (int*)out, 3, 16, 1);
  run("mul/mad_i24", k_mul_i24, (int*)out, 3, 16, 1);
  run("dep_add", k_dep_add, out, 1.0001f, 16, 1);
  return 0;
}
