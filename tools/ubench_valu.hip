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

template <class K, class T>
static void run(const char* name, K kern, T* out, T arg, double lane_ops_per_thread_iter, int pk) {
  for (int wps = 1; wps <= 8; wps *= 2) {
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

int main() {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 8 * 256);
  run("mul+add", k_mul_add, out, 1.0001f, 16, 1);
  run("pk_mul+pk_add", k_pk_mul_add, out, 1.0001f, 16, 2);
  run("fma", k_fma, out, 1.0001f, 16, 1);
  run("fma_const", k_fma_c, out, 1.0001f, 16, 1);
  run("fma_mix", k_fma_mix, out, 1.0001f, 16, 1);
  run("term9 x4", k_term9, out, 0.5f, 36, 1);
  run("sub|abs|+max", k_sub_abs_max, out, 1.0001f, 16, 1);
  run("mul_lo_u32", k_mul_lo, (int*)out, 3, 16, 1);
  run("mul/mad_i24", k_mul_i24, (int*)out, 3, 16, 1);
  run("dep_add", k_dep_add, out, 1.0001f, 16, 1);
  return 0;
}
