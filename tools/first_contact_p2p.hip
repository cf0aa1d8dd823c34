// first_contact_p2p.hip -- step (a) of tools/first_contact.sh: the transport qs_shard.cpp relies on, alone.
//
// qs_shard.cpp (the multi-GPU route behind the C ABI) moves one pixel row per neighbour and iteration with
// hipMemcpyPeerAsync on the RECEIVING band's stream, ordered by events recorded on the SENDING device's stream
// (cross-device hipStreamWaitEvent), after hipDeviceEnablePeerAccess between neighbouring devices.  None of that
// has ever met two devices (the development box has one).  This program exercises exactly those calls on
// devices 0..N-1 and checks every byte:
//   1. peer-access matrix (hipDeviceCanAccessPeer) and hipDeviceEnablePeerAccess between ring neighbours;
//   2. ROW ring: `iters` rounds; in round r device d fills an 8 KiB row with a (d, r) pattern by a kernel on its own
//      stream, records event A[d]; device (d + 1) % N waits for A[d] on ITS stream, pulls the row with
//      hipMemcpyPeerAsync, records X[d + 1] ("my pull is done"); the sender waits for that X before it overwrites the
//      row in round r + 1 -- the A / X protocol of run_sharded_set.  No host synchronisation inside the loop; all
//      received rows are verified at the end.  Reports the mean time per round (= the exposed latency of one halo
//      exchange);
//   3. bulk ring: 64 MiB per hop, the same way, for the xGMI bandwidth per link.
// N = 1 is the degenerate form (a device copies to itself): it runs on a one-GPU box and checks the protocol only.
// Exit code 0 = every check passed.  Build: hipcc -O2 tools/first_contact_p2p.hip -o tools/first_contact_p2p
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL: %s -> %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

__global__ void fill_row(uint32_t* p, size_t n, uint32_t tag) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = tag ^ (uint32_t)(i * 2654435761u);
}
__global__ void keep_row(uint32_t* dst, const uint32_t* src, size_t n) {   // receiver-side: file the pulled row under its round
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  int ndev = 0;
  CHECK(hipGetDeviceCount(&ndev));
  int N = argc > 1 ? atoi(argv[1]) : ndev;
  if (N < 1 || N > ndev) { printf("FAIL: asked for %d devices, %d visible\n", N, ndev); return 2; }
  const int iters = argc > 2 ? atoi(argv[2]) : 64;
  printf("first_contact_p2p: %d of %d visible device(s), %d rounds\n", N, ndev, iters);

  // ---- 1. peer access
  printf("peer-access matrix (row = device, column = peer; 1 = hipDeviceCanAccessPeer):\n");
  bool ring_ok = true;
  for (int a = 0; a < N; ++a) {
    printf("  dev %d:", a);
    for (int b = 0; b < N; ++b) {
      int can = a == b;
      if (a != b) CHECK(hipDeviceCanAccessPeer(&can, a, b));
      printf(" %d", can);
      if (a != b && (b == (a + 1) % N || a == (b + 1) % N) && !can) ring_ok = false;
    }
    printf("\n");
  }
  if (!ring_ok) printf("note: a ring neighbour is not peer-accessible: hipMemcpyPeerAsync will stage through the host\n");
  for (int a = 0; a < N; ++a) {
    CHECK(hipSetDevice(a));
    for (int b : {(a + 1) % N, (a + N - 1) % N}) {
      if (b == a) continue;
      int can = 0; CHECK(hipDeviceCanAccessPeer(&can, a, b));
      if (!can) continue;
      hipError_t e = hipDeviceEnablePeerAccess(b, 0);
      if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
      else if (e != hipSuccess) { printf("FAIL: hipDeviceEnablePeerAccess(%d -> %d): %s\n", a, b, hipGetErrorString(e)); return 2; }
    }
  }

  // ---- 2. row ring with the A / X event protocol
  const size_t row_words = 2112;                       // 8448 B: the pitch of an 8192-pixel plane row (apron included)
  std::vector<hipStream_t> st(N);
  std::vector<hipEvent_t> A(N), X(N), t0(N), t1(N);
  std::vector<uint32_t*> row(N), halo(N), log(N);      // row: what I send; halo: where I receive; log: every received row, by round
  for (int d = 0; d < N; ++d) {
    CHECK(hipSetDevice(d));
    CHECK(hipStreamCreateWithFlags(&st[d], hipStreamNonBlocking));
    CHECK(hipEventCreateWithFlags(&A[d], hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&X[d], hipEventDisableTiming));
    CHECK(hipEventCreate(&t0[d])); CHECK(hipEventCreate(&t1[d]));
    CHECK(hipMalloc(&row[d], row_words * 4)); CHECK(hipMalloc(&halo[d], row_words * 4));
    CHECK(hipMalloc(&log[d], row_words * 4 * (size_t)iters));
  }
  for (int pass = 0; pass < 2; ++pass) {               // pass 0 warms the queues up
    for (int d = 0; d < N; ++d) { CHECK(hipSetDevice(d)); CHECK(hipStreamSynchronize(st[d])); CHECK(hipEventRecord(t0[d], st[d])); }
    const double w0 = now_ms();
    for (int r = 0; r < iters; ++r) {
      for (int d = 0; d < N; ++d) {                    // "pass A": produce the row; first wait until the neighbour pulled the last one
        CHECK(hipSetDevice(d));
        const int nb = (d + 1) % N;
        if (r > 0) CHECK(hipStreamWaitEvent(st[d], X[nb], 0));
        hipLaunchKernelGGL(fill_row, dim3(4), dim3(256), 0, st[d], row[d], row_words, (uint32_t)(d * 1000003 + r));
        CHECK(hipEventRecord(A[d], st[d]));
      }
      for (int d = 0; d < N; ++d) {                    // the pull, on the receiver's stream, behind the sender's event
        CHECK(hipSetDevice(d));
        const int from = (d + N - 1) % N;
        CHECK(hipStreamWaitEvent(st[d], A[from], 0));
        CHECK(hipMemcpyPeerAsync(halo[d], d, row[from], from, row_words * 4, st[d]));
        CHECK(hipEventRecord(X[d], st[d]));
        hipLaunchKernelGGL(keep_row, dim3(4), dim3(256), 0, st[d], log[d] + (size_t)r * row_words, halo[d], row_words);
      }
    }
    const double w_enq = now_ms();
    for (int d = 0; d < N; ++d) { CHECK(hipSetDevice(d)); CHECK(hipEventRecord(t1[d], st[d])); }
    for (int d = 0; d < N; ++d) { CHECK(hipSetDevice(d)); CHECK(hipStreamSynchronize(st[d])); }
    const double w1 = now_ms();
    if (pass == 1) {
      float worst = 0;
      for (int d = 0; d < N; ++d) { float ms = 0; CHECK(hipSetDevice(d)); CHECK(hipEventElapsedTime(&ms, t0[d], t1[d])); if (ms > worst) worst = ms; }
      printf("row ring: %d rounds of {fill kernel, event, peer pull of %zu B, event, copy kernel}: %.1f us per round on the slowest device "
             "(host: enqueue %.2f ms, total %.2f ms)\n", iters, row_words * 4, worst * 1e3 / iters, w_enq - w0, w1 - w0);
    }
  }
  size_t bad = 0;
  std::vector<uint32_t> h(row_words * (size_t)iters);
  for (int d = 0; d < N; ++d) {
    CHECK(hipSetDevice(d));
    CHECK(hipMemcpy(h.data(), log[d], h.size() * 4, hipMemcpyDeviceToHost));
    const int from = (d + N - 1) % N;
    for (int r = 0; r < iters; ++r)
      for (size_t i = 0; i < row_words; ++i)
        if (h[(size_t)r * row_words + i] != ((uint32_t)(from * 1000003 + r) ^ (uint32_t)(i * 2654435761u))) ++bad;
  }
  printf("row ring: %zu wrong words of %zu\n", bad, (size_t)N * iters * row_words);

  // ---- 3. bulk ring: bandwidth per hop
  const size_t big = (size_t)64 << 20;
  std::vector<uint32_t*> src(N), dst(N);
  for (int d = 0; d < N; ++d) {
    CHECK(hipSetDevice(d));
    CHECK(hipMalloc(&src[d], big)); CHECK(hipMalloc(&dst[d], big));
    hipLaunchKernelGGL(fill_row, dim3(1024), dim3(256), 0, st[d], src[d], big / 4, (uint32_t)(77 + d));
    CHECK(hipEventRecord(A[d], st[d]));
  }
  for (int d = 0; d < N; ++d) { CHECK(hipSetDevice(d)); CHECK(hipStreamSynchronize(st[d])); }
  for (int d = 0; d < N; ++d) {
    CHECK(hipSetDevice(d));
    const int from = (d + N - 1) % N;
    CHECK(hipEventRecord(t0[d], st[d]));
    for (int k = 0; k < 4; ++k) CHECK(hipMemcpyPeerAsync(dst[d], d, src[from], from, big, st[d]));
    CHECK(hipEventRecord(t1[d], st[d]));
  }
  size_t bad2 = 0;
  std::vector<uint32_t> hb(big / 4);
  for (int d = 0; d < N; ++d) {
    CHECK(hipSetDevice(d));
    CHECK(hipStreamSynchronize(st[d]));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, t0[d], t1[d]));
    const int from = (d + N - 1) % N;
    printf("bulk ring: dev %d <- dev %d: 4 x 64 MiB in %.2f ms = %.1f GB/s (all hops concurrently)\n", d, from, ms, 4.0 * big / ms / 1e6);
    CHECK(hipMemcpy(hb.data(), dst[d], big, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < big / 4; i += 97) if (hb[i] != ((uint32_t)(77 + from) ^ (uint32_t)(i * 2654435761u))) ++bad2;
  }
  printf("bulk ring: %zu wrong sampled words\n", bad2);
  const bool ok = bad == 0 && bad2 == 0;
  printf("first_contact_p2p: %s\n", ok ? "PASS" : "FAIL");
  return ok ? 0 : 1;
}
