#!/usr/bin/env python3
"""Writes tools/ubench_lut.hip: issue-rate probes for a table-driven form of the recovery term
(LABNOTES.md 4.2, QS_LUT).  In  t = max(R-|d|,0)^2; x = d*t; y = w*t; num += x*y; den += y*y  the pair
(t, x) depends only on the integer pixel difference d and the coefficient's range R, so it can come
from a 511-entry table in LDS: one v_sub_u32_sdwa (address) + one ds_read_b64 replace three VALU
operations.  Whether that pays depends on what limits the kernel: the two VALU pipes of a SIMD (then
6 VALU per term beat 9) or instruction issue as such (then 8 instructions do not beat 9 by much).

Kernels (14 terms per loop trip, 4-wave workgroups, LDS sized so that exactly W workgroups fit a CU):
  base    the 9-operation term as the kernel has it today
  lutw    14 x (address, ds_read_b64), then per term: s_waitcnt lgkmcnt(n), 5 VALU
  lut1    the same with ONE s_waitcnt lgkmcnt(0) behind the 14 reads
  lut2    two register sets: the reads of the next 14 terms are in flight during the arithmetic of these
  lutpk   lut1 with y = w*t ; v_pk_mul_f32 {y*y, x*y} ; v_pk_add_f32 {den, num}
"""
from pathlib import Path

NT = 14
SDWA = "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0"


def loads(base_reg):
    s = ""
    for g in range(NT):
        r = base_reg + 2 * g
        s += f'"v_sub_u32_sdwa v{r}, %[p{g % 8}], %[p{(g + 1) % 8}] {SDWA}\\n"\n'
        s += f'"ds_read_b64 v[{r}:{r + 1}], v{r}\\n"\n'
    return s


def arith(base_reg, waits):
    s = ""
    for g in range(NT):
        r = base_reg + 2 * g
        if waits:
            s += f'"s_waitcnt lgkmcnt({NT - 1 - g})\\n"\n'
        s += f'"v_mul_f32 v98, %[w], v{r}\\n" "v_mul_f32 v99, v{r + 1}, v98\\n" "v_add_f32 %[num], %[num], v99\\n"\n'
        s += f'"v_mul_f32 v99, v98, v98\\n" "v_add_f32 %[den], %[den], v99\\n"\n'
    return s


def arith_pk(base_reg):
    s = ""
    for g in range(NT):
        r = base_reg + 2 * g
        s += f'"v_mul_f32 v{r}, %[w], v{r}\\n"\n'                                   # (y, x)
        s += f'"v_pk_mul_f32 v[98:99], v[{r}:{r + 1}], v[{r}:{r + 1}] op_sel:[0,0] op_sel_hi:[1,0]\\n"\n'   # (y*y, x*y)
        s += '"v_pk_add_f32 %[acc], %[acc], v[98:99]\\n"\n'
    return s


def base_terms():
    s = ""
    for g in range(NT):
        s += (f'"v_sub_f32 v98, %[f{g % 8}], %[f{(g + 1) % 8}]\\n" "v_sub_f32 v99, %[r], |v98| clamp\\n" "v_mul_f32 v99, v99, v99\\n"\n'
              '"v_mul_f32 v98, v98, v99\\n" "v_mul_f32 v99, %[w], v99\\n" "v_mul_f32 v98, v98, v99\\n"\n'
              '"v_add_f32 %[num], %[num], v98\\n" "v_mul_f32 v98, v99, v99\\n" "v_add_f32 %[den], %[den], v98\\n"\n')
    return s


CLOB = ", ".join(f'"v{r}"' for r in range(98, 100 + 4 * NT))
P_OPS = ", ".join(f'[p{k}] "v"(p{k})' for k in range(8))
F_OPS = ", ".join(f'[f{k}] "v"(f{k})' for k in range(8))

PRO = """
  extern __shared__ float2 lut[];
  const int lane = threadIdx.x & 63;
  for (int j = threadIdx.x; j < 512; j += blockDim.x) lut[j] = make_float2(1.0f / (1 + j), 0.5f / (1 + j));
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float2*)lut + 2040u;
  unsigned p0, p1, p2, p3, p4, p5, p6, p7;
  { unsigned* pp[8] = {&p0, &p1, &p2, &p3, &p4, &p5, &p6, &p7};
    for (int k = 0; k < 8; ++k) { unsigned lo = 8u * ((lane * 7 + k * 5 + spread * (lane >> 2)) & 31); *pp[k] = lo | ((lo + base) << 16); } }
  float num = 0.f, den = 0.f;
"""

EPI = "  out[blockIdx.x * blockDim.x + threadIdx.x] = num + den;\n}\n"


def kernel(name, body, acc_pair=False):
    s = f"__global__ void __launch_bounds__(256) {name}(float* out, float w_in, int spread) {{\n" + PRO
    s += "  float w = w_in;\n  asm volatile(\"\" : \"+s\"(w));\n"
    if acc_pair:
        s += "  typedef float f2 __attribute__((ext_vector_type(2)));\n  f2 acc = {0.f, 0.f};\n"
        s += "  for (int i = 0; i < ITER; ++i) {\n    asm volatile(\n" + body + f'      : [acc] "+v"(acc) : [w] "s"(w), {P_OPS} : {CLOB}, "memory");\n  }}\n'
        s += "  num = acc.x; den = acc.y;\n"
    else:
        s += "  for (int i = 0; i < ITER; ++i) {\n    asm volatile(\n" + body + f'      : [num] "+v"(num), [den] "+v"(den) : [w] "s"(w), {P_OPS} : {CLOB}, "memory");\n  }}\n'
    return s + EPI


def kernel_base():
    s = "__global__ void __launch_bounds__(256) k_base(float* out, float w_in, int spread) {\n"
    s += "  extern __shared__ float2 lut[];\n  const int lane = threadIdx.x & 63;\n  if (threadIdx.x == 0) lut[0] = make_float2(0.f, 0.f);\n"
    s += "  float f0, f1, f2, f3, f4, f5, f6, f7;\n  { float* pp[8] = {&f0, &f1, &f2, &f3, &f4, &f5, &f6, &f7};\n"
    s += "    for (int k = 0; k < 8; ++k) *pp[k] = 2048.0f + 0.000244140625f * ((lane * 7 + k * 5 + spread) & 31); }\n"
    s += "  float num = 0.f, den = 0.f;\n  float w = w_in, r = 0.01f;\n  asm volatile(\"\" : \"+s\"(w), \"+s\"(r));\n"
    s += "  for (int i = 0; i < ITER; ++i) {\n    asm volatile(\n" + base_terms()
    s += f'      : [num] "+v"(num), [den] "+v"(den) : [w] "s"(w), [r] "s"(r), {F_OPS} : "v98", "v99");\n  }}\n'
    return s + EPI


A, B = 100, 100 + 2 * NT
src = '''// GENERATED by tools/gen_ubench_lut.py -- do not edit.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_lut.hip -o build/ubench_lut
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 8192
'''
src += kernel_base()
src += kernel("k_lutw", loads(A) + arith(A, True))
src += kernel("k_lut1", loads(A) + '"s_waitcnt lgkmcnt(0)\\n"\n' + arith(A, False))
src += kernel("k_lut2", loads(B) + arith(A, False) + '"s_waitcnt lgkmcnt(0)\\n"\n' + loads(A) + arith(B, False) + '"s_waitcnt lgkmcnt(0)\\n"\n')
src += kernel("k_lutpk", loads(A) + '"s_waitcnt lgkmcnt(0)\\n"\n' + arith_pk(A), acc_pair=True)
src += '''
typedef void (*kern_t)(float*, float, int);
static void run(const char* name, kern_t k, int terms_per_trip, int spread) {
  float* out; (void)hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
  for (int W = 2; W <= 4; ++W) {
    // LDS per workgroup chosen so that exactly W workgroups (4 waves each: one per SIMD) fit a CU
    const size_t lds = W == 2 ? 65536 : W == 3 ? 53 * 1024 : 40 * 1024;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256 * W), dim3(256), lds, 0, out, 1.5f, spread);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\\n", name); return; }
    const double terms_per_simd = (double)ITER * terms_per_trip * W;      // W waves per SIMD
    printf("%-8s spread=%d W=%d  %8.3f ms  %6.2f cycles per term per SIMD (2.4 GHz)\\n", name, spread, W, best,
           best * 1e-3 * 2.4e9 / terms_per_simd);
  }
  (void)hipFree(out);
}
int main() {
  run("base", k_base, 14, 0);
  for (int spread = 0; spread <= 3; spread += 3) {      // 0: 32 distinct entries per wave, conflict-free; 3: wider, some bank conflicts
    run("lutw", k_lutw, 14, spread);
    run("lut1", k_lut1, 14, spread);
    run("lut2", k_lut2, 28, spread);
    run("lutpk", k_lutpk, 14, spread);
  }
  return 0;
}
'''
Path(__file__).with_name("ubench_lut.hip").write_text(src)
print("wrote tools/ubench_lut.hip")
