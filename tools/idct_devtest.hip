// device-vs-host check of the integer IDCT building blocks (debug aid)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#ifndef USE_MUL24
#define USE_MUL24 1
#endif
__host__ __device__ static inline uint32_t mulc(uint32_t a, int c) {
#if defined(__HIP_DEVICE_COMPILE__) && USE_MUL24
  return (uint32_t)__mul24((int)a, c);
#else
  return a * (uint32_t)c;
#endif
}
__host__ __device__ static inline void idct8(uint32_t (&v)[8]) {
  uint32_t z1, z2, z3, z4, z5, t0, t1, t2, t3, e0, e1, e2, e3;
  z2 = v[2]; z3 = v[6];
  z1 = mulc(z2 + z3, 4433);
  t2 = z1 - mulc(z3, 15137);
  t3 = z1 + mulc(z2, 6270);
  t0 = (v[0] + v[4]) << 13;
  t1 = (v[0] - v[4]) << 13;
  e0 = t0 + t3; e3 = t0 - t3; e1 = t1 + t2; e2 = t1 - t2;
  t0 = v[7]; t1 = v[5]; t2 = v[3]; t3 = v[1];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; z4 = t1 + t3;
  z5 = mulc(z3 + z4, 9633);
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
__host__ __device__ static inline void idct2d(const int16_t* c, int* out) {
  uint32_t ws[64];
  for (int i = 0; i < 64; i++) ws[i] = (uint32_t)(int32_t)c[i];
#pragma unroll
  for (int x = 0; x < 8; x++) { uint32_t col[8];
#pragma unroll
    for (int j = 0; j < 8; j++) col[j] = ws[j*8+x];
    idct8(col);
#pragma unroll
    for (int j = 0; j < 8; j++) ws[j*8+x] = (uint32_t)((int32_t)(col[j] + 1024u) >> 11); }
#pragma unroll
  for (int y = 0; y < 8; y++) { uint32_t row[8];
#pragma unroll
    for (int j = 0; j < 8; j++) row[j] = ws[y*8+j];
    idct8(row);
#pragma unroll
    for (int j = 0; j < 8; j++) { int z = (int32_t)(row[j] + (257u << 17)) >> 18; out[y*8+j] = z < 0 ? 0 : z > 255 ? 255 : z; } }
}
__global__ void k(const int16_t* c, int* out, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x; if (t >= n) return;
  int o[64]; idct2d(c + t * 64, o);
  for (int i = 0; i < 64; i++) out[t * 64 + i] = o[i];
}
int main() {
  const int n = 4096; srand(1);
  int16_t* hc = (int16_t*)malloc(n * 128); int* ho = (int*)malloc(n * 256);
  for (int i = 0; i < n * 64; i++) hc[i] = (rand() % 3 == 0) ? (rand() % 1200 - 600) : 0;
  int16_t* dc; int* dout; hipMalloc(&dc, n * 128); hipMalloc(&dout, n * 256);
  hipMemcpy(dc, hc, n * 128, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, dc, dout, n);
  hipMemcpy(ho, dout, n * 256, hipMemcpyDeviceToHost);
  int bad = 0, badpos[64] = {0};
  for (int t = 0; t < n; t++) { int o[64]; idct2d(hc + t * 64, o);
    for (int i = 0; i < 64; i++) if (o[i] != ho[t*64+i]) { if (bad < 5) printf("blk %d px %d dev %d host %d\n", t, i, ho[t*64+i], o[i]); bad++; badpos[i]++; } }
  printf("USE_MUL24=%d bad=%d\n", USE_MUL24, bad);
  for (int i = 0; i < 64; i++) if (badpos[i]) printf("pos%d:%d ", i, badpos[i]);
  printf("\n");
}
