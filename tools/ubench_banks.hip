// ubench_banks.hip -- does the VGPR bank of the two source operands matter on gfx950?
// Each kernel issues the same number of v_sub_f32 / v_mul_f32 with explicitly numbered
// registers: sources 1 apart, 4 apart, 8 apart (same bank if banks = index mod 4), the
// same register twice, and an SGPR operand.  Results go to v40..v47 (not read back in
// the loop), so there are no dependences.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_banks.hip -o /tmp/ubench_banks
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v40","v41","v42","v43","v44","v45","v46","v47"
#define INIT asm volatile("v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n" \
  "v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n" :: "v"(a) : CLOB)
#define FIN float r; asm volatile("v_add_f32 %0, v40, v41\n v_add_f32 %0, %0, v42\n v_add_f32 %0, %0, v43\n v_add_f32 %0, %0, v44\n v_add_f32 %0, %0, v45\n v_add_f32 %0, %0, v46\n v_add_f32 %0, %0, v47\n" : "=v"(r) :: CLOB); out[blockIdx.x * blockDim.x + threadIdx.x] = r
// OP dst, a, b for eight (a, b) pairs with b = a + D
#define EIGHT(OP, D0,D1,D2,D3,D4,D5,D6,D7) \
  OP " v40, v16, v" #D0 "\n" OP " v41, v17, v" #D1 "\n" OP " v42, v18, v" #D2 "\n" OP " v43, v19, v" #D3 "\n" \
  OP " v44, v20, v" #D4 "\n" OP " v45, v21, v" #D5 "\n" OP " v46, v22, v" #D6 "\n" OP " v47, v23, v" #D7 "\n"
#define KERNEL(NAME, BODY) __global__ void NAME(float* out, float a) { INIT; for (int i = 0; i < ITER; ++i) asm volatile(BODY BODY ::: CLOB); FIN; }
KERNEL(k_sub_d1, EIGHT("v_sub_f32", 17,18,19,20,21,22,23,24))
KERNEL(k_sub_d4, EIGHT("v_sub_f32", 20,21,22,23,24,25,26,27))
KERNEL(k_sub_d8, EIGHT("v_sub_f32", 24,25,26,27,28,29,30,31))
KERNEL(k_sub_d9, EIGHT("v_sub_f32", 25,26,27,28,29,30,31,32))
KERNEL(k_sub_d2, EIGHT("v_sub_f32", 18,19,20,21,22,23,24,25))
KERNEL(k_mul_d1, EIGHT("v_mul_f32", 17,18,19,20,21,22,23,24))
KERNEL(k_mul_d4, EIGHT("v_mul_f32", 20,21,22,23,24,25,26,27))
KERNEL(k_mul_d8, EIGHT("v_mul_f32", 24,25,26,27,28,29,30,31))
KERNEL(k_mul_same, EIGHT("v_mul_f32", 16,17,18,19,20,21,22,23))
__global__ void k_mul_sgpr(float* out, float a) { INIT; for (int i = 0; i < ITER; ++i) asm volatile(
  "v_mul_f32 v40, %0, v16\n v_mul_f32 v41, %0, v17\n v_mul_f32 v42, %0, v18\n v_mul_f32 v43, %0, v19\n v_mul_f32 v44, %0, v20\n v_mul_f32 v45, %0, v21\n v_mul_f32 v46, %0, v22\n v_mul_f32 v47, %0, v23\n"
  "v_mul_f32 v40, %0, v16\n v_mul_f32 v41, %0, v17\n v_mul_f32 v42, %0, v18\n v_mul_f32 v43, %0, v19\n v_mul_f32 v44, %0, v20\n v_mul_f32 v45, %0, v21\n v_mul_f32 v46, %0, v22\n v_mul_f32 v47, %0, v23\n" :: "s"(a) : CLOB); FIN; }
// destination bank = a source bank?
KERNEL(k_mul_dst_same_bank, "v_mul_f32 v40, v16, v17\n v_mul_f32 v41, v17, v18\n v_mul_f32 v42, v18, v19\n v_mul_f32 v43, v19, v20\n v_mul_f32 v44, v20, v21\n v_mul_f32 v45, v21, v22\n v_mul_f32 v46, v22, v23\n v_mul_f32 v47, v23, v24\n")
KERNEL(k_mul_dst_other_bank, "v_mul_f32 v42, v16, v17\n v_mul_f32 v43, v17, v18\n v_mul_f32 v44, v18, v19\n v_mul_f32 v45, v19, v20\n v_mul_f32 v46, v20, v21\n v_mul_f32 v47, v21, v22\n v_mul_f32 v40, v22, v23\n v_mul_f32 v41, v23, v24\n")

template <class K> static void run(const char* name, K kern, float* out) {
  for (int wps = 2; wps <= 8; wps *= 2) {
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 1.0001f); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-22s waves/SIMD=%d  %.3f ms  %.2f T lane-instr/s\n", name, wps, ms, (double)grid.x * 256 * ITER * 16 / ms * 1e-9);
  }
}
int main() {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 8 * 256);
  run("sub  src +1", k_sub_d1, out); run("sub  src +2", k_sub_d2, out); run("sub  src +4", k_sub_d4, out);
  run("sub  src +8", k_sub_d8, out); run("sub  src +9", k_sub_d9, out);
  run("mul  src +1", k_mul_d1, out); run("mul  src +4", k_mul_d4, out); run("mul  src +8", k_mul_d8, out);
  run("mul  same reg", k_mul_same, out); run("mul  sgpr x vgpr", k_mul_sgpr, out);
  run("mul  dst bank = src", k_mul_dst_same_bank, out); run("mul  dst other bank", k_mul_dst_other_bank, out);
  return 0;
}
