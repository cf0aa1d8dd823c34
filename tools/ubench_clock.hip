// ubench_clock.hip -- what shader clock does an MI355X SUSTAIN under a full FP32-VALU load?
// (The roofline peak of bench.py prices the kernel at 2.4 GHz: 256 CUs x 4 SIMDs x 32 lanes x 2.4e9 =
// 78.6 T non-fused lane-ops/s.  If the board holds a lower clock under this load, that share of the
// "missing" roofline fraction is not the kernel's to win.)
//
// Every workgroup's first lane reads s_memtime (shader-clock counter) and s_memrealtime (constant
// 100 MHz counter) before and after a long stream of the recovery kernel's 9-instruction term
// (v_sub, v_sub|x|clamp, 4 x v_mul, v_mul, 2 x v_add; weights from SGPRs) -- shader MHz = 100 * d(memtime) /
// d(memrealtime).  Also printed: instructions issued per SIMD cycle, from the same counters.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_clock.hip -o build/ubench_clock && build/ubench_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define TERM(P) \
  "v_sub_f32 %[d], %[a" #P "], %[b]\n" \
  "v_sub_f32 %[u], %[r], |%[d]| clamp\n" \
  "v_mul_f32 %[t], %[u], %[u]\n" \
  "v_mul_f32 %[x], %[d], %[t]\n" \
  "v_mul_f32 %[y], %[w], %[t]\n" \
  "v_mul_f32 %[x], %[x], %[y]\n" \
  "v_add_f32 %[n], %[n], %[x]\n" \
  "v_mul_f32 %[y], %[y], %[y]\n" \
  "v_add_f32 %[e], %[e], %[y]\n"

__global__ void __launch_bounds__(256) k_terms(float* out, unsigned long long* stamps, int iters, float r, float w) {
  float a0 = threadIdx.x * 0.001f, a1 = a0 + 0.01f, a2 = a0 + 0.02f, a3 = a0 + 0.03f, b = 0.5f * a0;
  float n = 0, e = 0, d, u, t, x, y;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
  for (int i = 0; i < iters; ++i) {
    asm volatile(TERM(0) TERM(1) TERM(2) TERM(3) TERM(0) TERM(1) TERM(2) TERM(3)
                 TERM(0) TERM(1) TERM(2) TERM(3) TERM(0) TERM(1) TERM(2) TERM(3)
                 : [n] "+v"(n), [e] "+v"(e), [d] "=&v"(d), [u] "=&v"(u), [t] "=&v"(t), [x] "=&v"(x), [y] "=&v"(y)
                 : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [b] "v"(b), [r] "s"(r), [w] "s"(w));
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) : "v"(n), "v"(e));
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = n + e;
}

int main() {
  const int cus = 256;
  float* out; unsigned long long* st;
  hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  hipMalloc(&st, sizeof(unsigned long long) * 2 * cus * 8);
  std::vector<unsigned long long> h(2 * cus * 8);
  printf("# 9-instruction term stream, 144 VALU instructions per loop step; workgroups of 4 waves (one per SIMD)\n");
  printf("# wg/CU = waves per SIMD.  'shader MHz' from s_memtime / s_memrealtime (100 MHz) of every workgroup's first wave.\n");
  for (int rep = 0; rep < 2; ++rep)
    for (int wg_per_cu : {1, 2, 3, 4, 8}) {
      for (int iters : {2000, 40000}) {            // ~0.4 ms and ~8 ms per wave at 2 waves per SIMD
        const int grid = cus * wg_per_cu;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_terms, dim3(grid), dim3(256), 0, 0, out, st, iters, 0.75f, 0.33f);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
        std::vector<double> mhz, cpi;
        for (int g = 0; g < grid; ++g) {
          const double dt = (double)h[2 * g], dr = (double)h[2 * g + 1];
          if (dr > 0) { mhz.push_back(100.0 * dt / dr); cpi.push_back(dt / (144.0 * iters)); }
        }
        std::sort(mhz.begin(), mhz.end()); std::sort(cpi.begin(), cpi.end());
        const double lane_ops = 144.0 * iters * 64.0 * 4.0 * grid;
        const int resident = wg_per_cu > 8 ? 8 : wg_per_cu;
        printf("wg/CU %d  iters %6d  kernel %8.3f ms  %6.2f T lane-instr/s  shader MHz min/median/max %6.0f %6.0f %6.0f  "
               "cycles per wave-instruction (one wave) %5.2f  => per SIMD %5.2f\n",
               wg_per_cu, iters, ms, lane_ops / (ms * 1e-3) / 1e12, mhz.front(), mhz[mhz.size() / 2], mhz.back(),
               cpi[cpi.size() / 2], cpi[cpi.size() / 2] / resident);
      }
    }
  return 0;
}
