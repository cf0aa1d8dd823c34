// PCIe staging probe: pageable hipMemcpy vs hipHostRegister+copy vs pinned staging (decides the job layer's transfer path)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = 128u << 20;
  char* h = (char*)malloc(n); memset(h, 1, n);
  char* d; hipMalloc(&d, n);
  char* pin; hipHostMalloc(&pin, n);
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now(); hipMemcpy(d, h, n, hipMemcpyHostToDevice); double t1 = now();
    hipMemcpy(h, d, n, hipMemcpyDeviceToHost); double t2 = now();
    printf("pageable      H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, (t2 - t1) * 1e3, n / (t2 - t1) / 1e9);
    t0 = now(); hipHostRegister(h, n, hipHostRegisterDefault); t1 = now();
    hipMemcpy(d, h, n, hipMemcpyHostToDevice); t2 = now();
    hipMemcpy(h, d, n, hipMemcpyDeviceToHost); double t3 = now();
    hipHostUnregister(h); double t4 = now();
    printf("register      reg %.2f ms  H2D %.2f ms (%.1f GB/s)  D2H %.2f ms  unreg %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, n / (t2 - t1) / 1e9, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
    t0 = now(); hipMemcpy(d, pin, n, hipMemcpyHostToDevice); t1 = now();
    printf("pinned        H2D %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
    for (int nt : {1, 4, 8}) {
      t0 = now();
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { size_t c = n / nt; memcpy(pin + t * c, h + t * c, c); });
      for (auto& x : th) x.join();
      t1 = now();
      printf("memcpy->pinned %d threads: %.2f ms (%.1f GB/s)\n", nt, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
    }
  }
  return 0;
}
