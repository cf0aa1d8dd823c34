// qs_tables.cpp -- error state, layout queries and the per-component constant block
// (quant-derived values + weight tables) of the flat C ABI (include/jpegqs_hip.h).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (the weight tables are
// float and must be bit-identical to the reference's, so no contraction on the
// host side either).
#include <stdarg.h>
#include <math.h>
#include <mutex>

#include "qs_xfer.h"

// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int qs_fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}


extern "C" const char* qs_hip_last_error(void) { return g_err; }

extern "C" int qs_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

extern "C" size_t qs_hip_consts_bytes(void) { return sizeof(QsConsts); }
extern "C" size_t qs_hip_plane_pitch(int wblk) { return (size_t)qs_plane_pitch(wblk); }
extern "C" size_t qs_hip_plane_bytes(int wblk, int hblk) {
  return (size_t)qs_plane_pitch(wblk) * ((size_t)hblk * 8 + 2) + 64;
}
extern "C" size_t qs_hip_plane_row_offset(int wblk, int y) {
  return (size_t)qs_plane_pitch(wblk) * (size_t)(y + 1);
}
extern "C" void qs_hip_free(void* p) { if (p && !qsx::pinned_return(p)) free(p); }

// ---------------------------------------------------------------------------
// constants

// zigzag position -> natural index (ITU T.81 Figure 5; reference idct.h:24-33)
static const unsigned char kZigzag[64] = {
  0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
  58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};

// 8-point float LL&M inverse DCT; operation order is the reference's
// (reference idct.h:568-591) because the weight tables must match bit for bit.
static void idct8f(const float* in, int is, float* out, int os, bool scale) {
  float z1, z2, z3, z4, z5, t0, t1, t2, t3, t4, t5, t6, t7;
  z2 = in[2 * is]; z3 = in[6 * is];
  z1 = (z2 + z3) * 0.541196100f;
  t2 = z1 - z3 * 1.847759065f;
  t3 = z1 + z2 * 0.765366865f;
  z2 = in[0]; z3 = in[4 * is];
  t0 = z2 + z3; t1 = z2 - z3;
  t4 = t0 + t3; t7 = t0 - t3; t5 = t1 + t2; t6 = t1 - t2;
  t0 = in[7 * is]; t1 = in[5 * is]; t2 = in[3 * is]; t3 = in[is];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; z4 = t1 + t3;
  z5 = (z3 + z4) * 1.175875602f;
  t0 = t0 * 0.298631336f; t1 = t1 * 2.053119869f;
  t2 = t2 * 3.072711026f; t3 = t3 * 1.501321110f;
  z1 = z1 * 0.899976223f; z2 = z2 * 2.562915447f;
  z3 = z3 * 1.961570560f; z4 = z4 * 0.390180644f;
  z3 = z3 - z5; t0 = t0 - (z1 + z3); t2 = t2 - (z2 + z3);
  z4 = z4 - z5; t1 = t1 - (z2 + z4); t3 = t3 - (z1 + z4);
  float r[8] = { t4 + t3, t5 + t2, t6 + t1, t7 + t0, t7 - t0, t6 - t1, t5 - t2, t4 - t3 };
  for (int j = 0; j < 8; ++j) out[j * os] = scale ? r[j] * 0.125f : r[j];
}

static void impulse_response(int i, float T[64]) {
  float in[64], ws[64];
  memset(in, 0, sizeof(in)); in[i] = 1.0f;
  for (int x = 0; x < 8; ++x) idct8f(in + x, 8, ws + x, 8, false);       // columns
  for (int y = 0; y < 8; ++y) idct8f(ws + y * 8, 1, T + y * 8, 1, true); // rows
}

// The weight tables depend only on the DIAGONALS flag: built (and checked) once
// per process, copied into every component's constant block.
struct WeightTables {
  float tab[2][64 * QS_TAB_MAX];
  int status[2];
  char msg[2][160];
};

static void build_tables(WeightTables& W, int diag) {
  const int ts = diag ? 272 : 160;
  const float b = diag ? 4.0f : 2.0f;
  float* tab = W.tab[diag];
  memset(tab, 0, sizeof(W.tab[diag]));
  W.status[diag] = QS_HIP_OK;
  for (int k = 0; k < 64; ++k) {
    const int i = kZigzag[k];
    float T[64], *w = tab + (size_t)k * ts;      // reference :251-301, layout in SURVEY A.4
    impulse_response(i, T);
    for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) {
      const int p = y * 8 + x;
      w[p] = x < 7 ? T[p] - T[p + 1] : 0.0f;
      w[96 + p] = y < 7 ? T[p] - T[p + 8] : 0.0f;
    }
    for (int x = 0; x < 8; ++x) {
      w[64 + x] = T[x] * b; w[72 + x] = T[56 + x] * b;
      w[80 + x] = T[8 * x] * b; w[88 + x] = T[8 * x + 7] * b;
    }
    if (diag)
      for (int y = 0; y < 7; ++y) for (int x = 0; x < 8; ++x) {
        const int p = y * 8 + x;
        w[160 + 16 * y + x] = x < 7 ? T[p] - T[p + 9] : 0.0f;
        w[168 + 16 * y + x] = x < 7 ? T[p + 1] - T[p + 8] : 0.0f;
      }
  }
  // The kernel skips the horizontal / vertical difference terms whose weight is
  // structurally zero ((x+1)*u or (y+1)*v a multiple of 8, see qs_kernels.hip);
  // that is only exact if the float tables really hold 0.0f there.
  for (int k = 1; k < 64; ++k) {
    const int i = kZigzag[k], u = i & 7, v = i >> 3;
    const float* w = tab + (size_t)k * ts;
    for (int y = 0; y < 8; ++y) for (int x = 0; x < 7; ++x)
      if (u && ((x + 1) * u) % 8 == 0 && w[y * 8 + x] != 0.0f) {
        W.status[diag] = QS_HIP_EINVAL;
        snprintf(W.msg[diag], sizeof(W.msg[diag]), "weight table: expected exact zero at k=%d h(%d,%d)", k, y, x);
      }
    for (int y = 0; y < 7; ++y) for (int x = 0; x < 8; ++x)
      if (v && ((y + 1) * v) % 8 == 0 && w[96 + y * 8 + x] != 0.0f) {
        W.status[diag] = QS_HIP_EINVAL;
        snprintf(W.msg[diag], sizeof(W.msg[diag]), "weight table: expected exact zero at k=%d v(%d,%d)", k, y, x);
      }
  }
  // The kernel evaluates the sums in a 2^-k scaled domain (QS_TERM_D); that is
  // exact only while no product underflows, which needs every non-zero weight
  // to be comfortably above 2^-38.
  for (size_t j = 0; j < (size_t)64 * ts; ++j) {
    const float a = tab[j] < 0 ? -tab[j] : tab[j];
    if (a != 0.0f && a < 2.3283064e-10f /* 2^-32 */) {
      W.status[diag] = QS_HIP_EINVAL;
      snprintf(W.msg[diag], sizeof(W.msg[diag]), "weight table entry %g too small for the scaled evaluation", (double)a);
    }
  }
}

static const WeightTables& weight_tables() {
  static WeightTables* W = [] {
    WeightTables* w = new WeightTables;
    build_tables(*w, 0);
    build_tables(*w, 1);
    return w;
  }();
  return *W;
}

extern "C" int qs_hip_consts_build(void* host_out, const uint16_t quant[64], int flags) {
  if (!host_out || !quant) return qs_fail(QS_HIP_EINVAL, "qs_hip_consts_build: null argument");
  QsConsts* c = static_cast<QsConsts*>(host_out);
  const int diag = (flags & QS_DIAGONALS) != 0;
  const int ts = diag ? 272 : 160;
  const WeightTables& W = weight_tables();
  if (W.status[diag] != QS_HIP_OK) return qs_fail(W.status[diag], "%s", W.msg[diag]);
  memset(c, 0, offsetof(QsConsts, tab));
  c->tab_size = ts;
  int qn[64], x1n[64], x2n[64];
  for (int i = 0; i < 64; ++i) {           // reference :2506-2539
    unsigned q = quant[i] ? quant[i] : 1u, n = 0, t = q;
    while (t > 1) { t >>= 1; ++n; }
    unsigned x1 = ((0x10000u << n) + q - 1) / q;
    if (n) x1 |= x1 >> 16;
    // the reference's second table entry is x2 = -0x8000 >> n, used as (-a * x2 + 0x4000) >> 15:
    // -a * x2 == a << (15 - n), so the device table carries the shift (an integer multiply costs
    // about four adds on gfx950, a shift one)
    const int sh = 15 - (int)n;
    qn[i] = (int)q; x1n[i] = (int16_t)(uint16_t)x1; x2n[i] = sh;
    c->qraw[i] = quant[i];
    c->qn[i] = qn[i]; c->x1n[i] = x1n[i]; c->x2n[i] = x2n[i];
  }
  for (int k = 0; k < 64; ++k) {
    const int i = kZigzag[k];
    c->nat[k] = i; c->q[k] = qn[i]; c->x1[k] = x1n[i]; c->x2[k] = x2n[i];
    c->range[k] = (float)(qn[i] * 2) * 0.000244140625f;  // R * 2^-12, see QS_TERM_D
  }
  for (int k = 0; k < 64; ++k) {
    const int kn = k > 1 ? k - 1 : 1;
    c->rec[k][0] = c->nat[k] | (c->nat[kn] << 8) | (c->x2[k] << 16);
    c->rec[k][1] = (int32_t)(((uint32_t)c->q[k] & 0xffffu) | ((uint32_t)c->x1[k] << 16));
    memcpy(&c->rec[k][2], &c->range[k], sizeof(float));
    {   // the scalars of the recovery loop that depend on k alone (qs_device.h: QS_REC_*)
      const int i = c->nat[k], u = i & 7, v = i >> 3, i_nxt = c->nat[kn];
      int m = 0;
      if (u) m |= QS_REC_H_ANY;
      if (u == 4) m |= QS_REC_H_SKIP4;
      if (!(u & 1)) m |= QS_REC_H_EVEN;
      if (v) m |= QS_REC_V_ANY;
      if (v == 4) m |= QS_REC_V_SKIP4;
      if (!(v & 1)) m |= QS_REC_V_EVEN;
      m |= ((i >> 1) * QS_LDS_PITCH * 4 + (i & 1) * 2) << QS_REC_LDS_SHIFT;
      if (!(i_nxt & 7)) m |= 0x100 << QS_REC_NXT_SHIFT;
      if (c->q[k] == 1) m |= QS_REC_Q1;
      c->rec[k][3] = m;
    }
  }
  memcpy(c->tab, W.tab[diag], sizeof(c->tab));
  return QS_HIP_OK;
}

