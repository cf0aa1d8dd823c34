// qs_host.cpp -- host side of the flat C ABI (include/jpegqs_hip.h).
//
// Plane layer: thin argument checking + kernel launches.
// Job layer  : the host-side semantics of the reference's plane driver
//              (reference quantsmooth.h:2404-2878): validation, early-outs,
//              iteration loop with progress/cancel between launches, final
//              clamp, quant tables := 1 -- with the per-plane passes running
//              as gfx950 kernels.  No CPU compute fallback exists.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (the weight tables
// are float and must be bit-identical to the reference's, so no contraction
// on the host side either).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <stddef.h>
#include <math.h>
#include <new>
#include <mutex>
#include <vector>
#include <deque>
#include <list>
#include <atomic>
#include <thread>
#include <condition_variable>
#include <functional>
#include <memory>
#include <algorithm>
#include <chrono>

#include "../../include/jpegqs_hip.h"
#include "qs_device.h"
#include "qs_launch.h"

extern "C" void qs_hip_release_cache(void);

// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
  return fail(e_ == hipErrorOutOfMemory ? QS_HIP_ENOMEM : QS_HIP_ENODEV, \
              "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

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
extern "C" void qs_hip_free(void* p) { free(p); }

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
  if (!host_out || !quant) return fail(QS_HIP_EINVAL, "qs_hip_consts_build: null argument");
  QsConsts* c = static_cast<QsConsts*>(host_out);
  const int diag = (flags & QS_DIAGONALS) != 0;
  const int ts = diag ? 272 : 160;
  const WeightTables& W = weight_tables();
  if (W.status[diag] != QS_HIP_OK) return fail(W.status[diag], "%s", W.msg[diag]);
  memset(c, 0, offsetof(QsConsts, tab));
  c->tab_size = ts;
  int qn[64], x1n[64], x2n[64];
  for (int i = 0; i < 64; ++i) {           // reference :2506-2539
    unsigned q = quant[i] ? quant[i] : 1u, n = 0, t = q;
    while (t > 1) { t >>= 1; ++n; }
    unsigned x1 = ((0x10000u << n) + q - 1) / q;
    if (n) x1 |= x1 >> 16;
    int x2 = -0x8000 >> n;
    qn[i] = (int)q; x1n[i] = (int16_t)(uint16_t)x1; x2n[i] = (int16_t)(uint16_t)x2;
    c->qraw[i] = quant[i];
    c->qn[i] = qn[i]; c->x1n[i] = x1n[i]; c->x2n[i] = x2n[i];
  }
  for (int k = 0; k < 64; ++k) {
    const int i = kZigzag[k];
    c->nat[k] = i; c->q[k] = qn[i]; c->x1[k] = x1n[i]; c->x2[k] = x2n[i];
    c->range[k] = (float)(qn[i] * 2) * 0.000244140625f;  // R * 2^-12, see QS_TERM_D
  }
  memcpy(c->tab, W.tab[diag], sizeof(c->tab));
  return QS_HIP_OK;
}

// ---------------------------------------------------------------------------
// plane layer

static int check_plane_args(const void* a, const void* b, int wblk, int hblk, const char* who) {
  if (!a || !b) return fail(QS_HIP_EINVAL, "%s: null device pointer", who);
  if (wblk <= 0 || hblk <= 0 || (long long)wblk * hblk > (1ll << 28))
    return fail(QS_HIP_EINVAL, "%s: bad plane size %dx%d blocks", who, wblk, hblk);
  return 0;
}

static int launch_status(const char* who) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QS_HIP_ENODEV, "%s: launch failed: %s", who, hipGetErrorString(e));
  return QS_HIP_OK;
}

extern "C" int qs_hip_idct_plane(const void* d_consts, int16_t* d_coef, uint8_t* d_plane,
                                 int wblk, int hblk, int first, int rep_top, int rep_bot,
                                 int32_t* d_status, void* stream) {
  if (int r = check_plane_args(d_coef, d_plane, wblk, hblk, "qs_hip_idct_plane")) return r;
  if (!d_consts || (first && !d_status)) return fail(QS_HIP_EINVAL, "qs_hip_idct_plane: null consts/status");
  qs_launch_idct_plane(static_cast<const QsConsts*>(d_consts), d_coef, d_plane, wblk, hblk,
                       first, rep_top, rep_bot, d_status, static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_idct_plane");
}

static int smooth_rows(const void* d_consts, int16_t* d_coef, const uint8_t* d_plane, int wblk, int hblk,
                       int row0, int row1, int flags, int luma, int final_clamp, void* stream, const char* who) {
  if (int r = check_plane_args(d_coef, d_plane, wblk, hblk, who)) return r;
  if (!d_consts) return fail(QS_HIP_EINVAL, "%s: null consts", who);
  if (row0 < 0 || row1 > hblk || row0 > row1) return fail(QS_HIP_EINVAL, "%s: bad row range %d..%d", who, row0, row1);
  if (flags & (QS_JOINT_YUV | QS_UPSAMPLE_UV | QS_LOW_QUALITY))
    return fail(QS_HIP_ENOTSUP, "%s: flags 0x%x are handled by qs_hip_joint_plane / qs_hip_lowq_plane", who, flags);
  int rebalance = !(flags & QS_NO_REBALANCE) && (luma || !(flags & QS_NO_REBALANCE_UV)); // reference :1567-1568
  qs_launch_smooth_plane(static_cast<const QsConsts*>(d_consts), d_coef, d_plane, wblk, hblk,
                         (flags & QS_DIAGONALS) != 0, rebalance, final_clamp, row0 * wblk, row1 * wblk,
                         static_cast<hipStream_t>(stream));
  return launch_status(who);
}

extern "C" int qs_hip_smooth_plane(const void* d_consts, int16_t* d_coef, const uint8_t* d_plane,
                                   int wblk, int hblk, int flags, int luma, int final_clamp, void* stream) {
  return smooth_rows(d_consts, d_coef, d_plane, wblk, hblk, 0, hblk, flags, luma, final_clamp, stream, "qs_hip_smooth_plane");
}

extern "C" int qs_hip_smooth_rows(const void* d_consts, int16_t* d_coef, const uint8_t* d_plane,
                                  int wblk, int hblk, int row0, int row1, int flags, int luma, int final_clamp, void* stream) {
  return smooth_rows(d_consts, d_coef, d_plane, wblk, hblk, row0, row1, flags, luma, final_clamp, stream, "qs_hip_smooth_rows");
}

static int build_plane_set(const qs_hip_plane_ref* refs, int n, int flags, QsPlaneSet& set, const char* who) {
  if (!refs || n < 1 || n > QS_MAX_PLANES) return fail(QS_HIP_EINVAL, "%s: 1..%d planes per launch", who, QS_MAX_PLANES);
  memset(&set, 0, sizeof set);
  set.n = n;
  int w = 0;
  for (int i = 0; i < n; ++i) {
    const qs_hip_plane_ref& r = refs[i];
    if (int e = check_plane_args(r.d_coef, r.d_plane, r.wblk, r.hblk, who)) return e;
    if (!r.d_consts) return fail(QS_HIP_EINVAL, "%s: null consts", who);
    set.wave0[i] = w;
    w += (r.wblk * r.hblk + 63) / 64;
    QsPlaneRef& R = set.ref[i];
    R.cst = static_cast<const QsConsts*>(r.d_consts);
    R.coef = r.d_coef; R.plane = r.d_plane; R.status = r.d_status;
    R.wblk = r.wblk; R.hblk = r.hblk; R.pitch = qs_plane_pitch(r.wblk);
    R.rebalance = !(flags & QS_NO_REBALANCE) && (r.luma || !(flags & QS_NO_REBALANCE_UV));
  }
  for (int i = n; i < QS_MAX_PLANES + 2; ++i) set.wave0[i] = w;
  return QS_HIP_OK;
}

extern "C" int qs_hip_idct_planes(const qs_hip_plane_ref* refs, int n, int first, void* stream) {
  QsPlaneSet set;
  if (int r = build_plane_set(refs, n, 0, set, "qs_hip_idct_planes")) return r;
  if (first)
    for (int i = 0; i < n; ++i)
      if (!refs[i].d_status) return fail(QS_HIP_EINVAL, "qs_hip_idct_planes: first pass needs d_status");
  qs_launch_idct_set(set, first, static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_idct_planes");
}

extern "C" int qs_hip_smooth_planes(const qs_hip_plane_ref* refs, int n, int flags, int final_clamp, void* stream) {
  if (flags & (QS_JOINT_YUV | QS_UPSAMPLE_UV | QS_LOW_QUALITY))
    return fail(QS_HIP_ENOTSUP, "qs_hip_smooth_planes: flags 0x%x need the cross-component stages", flags);
  QsPlaneSet set;
  if (int r = build_plane_set(refs, n, flags, set, "qs_hip_smooth_planes")) return r;
  qs_launch_smooth_set(set, (flags & QS_DIAGONALS) != 0, final_clamp, static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_smooth_planes");
}

extern "C" int qs_hip_clamp_plane(int16_t* d_coef, int wblk, int hblk, void* stream) {
  if (int r = check_plane_args(d_coef, d_coef, wblk, hblk, "qs_hip_clamp_plane")) return r;
  qs_launch_clamp(d_coef, (size_t)wblk * hblk, static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_clamp_plane");
}

extern "C" int qs_hip_dequant_plane(const void* d_consts, int16_t* d_coef, int wblk, int hblk, void* stream) {
  if (int r = check_plane_args(d_coef, d_consts, wblk, hblk, "qs_hip_dequant_plane")) return r;
  qs_launch_dequant(static_cast<const QsConsts*>(d_consts), d_coef, (size_t)wblk * hblk, static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_dequant_plane");
}

// ---------------------------------------------------------------------------
// plane layer, cross-component / low-quality stages

extern "C" int qs_hip_joint_plane(const void* d_consts, int16_t* d_coef, const uint8_t* d_plane,
                                  const uint8_t* d_luma_lowres, int wblk, int hblk,
                                  int rebalance, int final_clamp, void* stream) {
  if (int r = check_plane_args(d_coef, d_plane, wblk, hblk, "qs_hip_joint_plane")) return r;
  if (!d_consts || !d_luma_lowres) return fail(QS_HIP_EINVAL, "qs_hip_joint_plane: null consts/luma plane");
  qs_launch_joint(static_cast<const QsConsts*>(d_consts), d_coef, d_plane, d_luma_lowres, wblk, hblk,
                  rebalance, final_clamp, static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_joint_plane");
}

extern "C" int qs_hip_lowq_plane(const void* d_consts, int16_t* d_coef, const uint8_t* d_plane,
                                 int wblk, int hblk, int rebalance, int final_clamp, void* stream) {
  if (int r = check_plane_args(d_coef, d_plane, wblk, hblk, "qs_hip_lowq_plane")) return r;
  if (!d_consts) return fail(QS_HIP_EINVAL, "qs_hip_lowq_plane: null consts");
  const float c1 = 2.0f * sqrtf(0.5f);   // reference quantsmooth.h:926
  qs_launch_lowq(static_cast<const QsConsts*>(d_consts), d_coef, d_plane, wblk, hblk, rebalance, final_clamp, c1,
                 static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_lowq_plane");
}

extern "C" int qs_hip_downsample_plane(const uint8_t* d_luma, int ywblk, int yhblk, uint8_t* d_lowres,
                                       int lwblk, int lhblk, int ws, int hs, void* stream) {
  if (int r = check_plane_args(d_luma, d_lowres, ywblk, yhblk, "qs_hip_downsample_plane")) return r;
  if (lwblk <= 0 || lhblk <= 0 || ws < 1 || hs < 1 || ws > 4 || hs > 4)
    return fail(QS_HIP_EINVAL, "qs_hip_downsample_plane: bad geometry");
  qs_launch_downsample(d_luma, ywblk, yhblk, d_lowres, lwblk, lhblk, ws, hs, static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_downsample_plane");
}

extern "C" size_t qs_hip_upsample_pitch(int image_width, int ws) {
  const int w1 = (image_width + ws - 1) / ws;
  return (size_t)(((w1 + 8) & -8) * ws);              // reference quantsmooth.h:2714
}
extern "C" size_t qs_hip_upsample_bytes(int image_width, int image_height, int ws, int hs) {
  const int h1 = (image_height + hs - 1) / hs;
  return qs_hip_upsample_pitch(image_width, ws) * (size_t)(((h1 + 8) & -8) * hs) + 64;   // reference :2715-2716
}

extern "C" int qs_hip_upsample_rows(const uint8_t* d_chroma, const uint8_t* d_luma_lowres, int cwblk,
                                    const uint8_t* d_luma, int ywblk, int yhblk, uint8_t* d_pixels, size_t pitch,
                                    int w1, int h1, int first_rows, int ws, int hs, void* stream) {
  if (int r = check_plane_args(d_chroma, d_luma, ywblk, yhblk, "qs_hip_upsample_rows")) return r;
  if (!d_luma_lowres || !d_pixels || cwblk <= 0 || w1 <= 0 || h1 < 0 || first_rows < 0 || first_rows > h1)
    return fail(QS_HIP_EINVAL, "qs_hip_upsample_rows: bad argument");
  qs_launch_upsample(d_chroma, d_luma_lowres, cwblk, d_luma, ywblk, d_pixels, (int)pitch,
                     ywblk * 8, yhblk * 8, w1, h1, first_rows, ws, hs, static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_upsample_rows");
}

extern "C" int qs_hip_upsample_plane(const uint8_t* d_chroma, const uint8_t* d_luma_lowres, int cwblk,
                                     const uint8_t* d_luma, int ywblk, int yhblk, uint8_t* d_pixels,
                                     int image_width, int image_height, int ws, int hs, void* stream) {
  const int w1 = (image_width + ws - 1) / ws, h1 = (image_height + hs - 1) / hs;
  return qs_hip_upsample_rows(d_chroma, d_luma_lowres, cwblk, d_luma, ywblk, yhblk, d_pixels,
                              qs_hip_upsample_pitch(image_width, ws), w1, h1, h1 < 8 ? h1 : 8, ws, hs, stream);
}

extern "C" int qs_hip_fdct_plane(const uint8_t* d_pixels, size_t pitch, int16_t* d_coef, int wblk, int hblk, void* stream) {
  if (int r = check_plane_args(d_pixels, d_coef, wblk, hblk, "qs_hip_fdct_plane")) return r;
  qs_launch_fdct_plane(d_pixels, (int)pitch, d_coef, wblk, hblk, static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_fdct_plane");
}

// ---------------------------------------------------------------------------
// job layer
//
// Two execution modes share one component routine:
//  * careful  -- the reference's order: one component after the other, host
//                sync after the first pass A of each (bad-coefficient stop,
//                reference :2610) and after every iteration that reports
//                progress.  Used whenever a progress callback is installed, and
//                as the re-run path below.
//  * eager    -- no callback: every component is enqueued without host syncs,
//                independent components on their own HIP streams ("one
//                component per stream", BASELINE config 1; chroma waits for
//                luma through an event when JOINT_YUV/UPSAMPLE_UV couple them).
//                The range-check flags are read once at the end; nothing is
//                copied back before that.  If any flag is set (crafted or
//                damaged file) the job is simply re-run in careful mode from
//                the untouched host input, which reproduces the reference's
//                stop semantics exactly.
// Device buffers come from a small process-wide cache (hipMalloc/hipFree of
// 100+ MiB cost milliseconds each); qs_hip_release_cache() empties it.

namespace {

struct CacheEntry { void* p; size_t n; };
static std::mutex g_cache_mu;
static std::vector<CacheEntry> g_cache;            // free device blocks
static const size_t kCacheMaxBytes = (size_t)6 << 30;

static size_t round_size(size_t n) {               // size classes: powers of two from 64 KiB
  size_t c = (size_t)64 << 10;
  while (c < n) c <<= 1;
  return c;
}

struct DevBuf {
  void* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  hipError_t alloc(size_t bytes) {
    release();
    const size_t want = round_size(bytes);
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      for (size_t i = 0; i < g_cache.size(); ++i)
        if (g_cache[i].n == want) { p = g_cache[i].p; n = want; g_cache.erase(g_cache.begin() + i); return hipSuccess; }
    }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {                         // make room and retry once
      (void)hipGetLastError();
      qs_hip_release_cache();
      e = hipMalloc(&p, want);
    }
    if (e == hipSuccess) n = want; else p = nullptr;
    return e;
  }
  void release() {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    size_t held = 0;
    for (auto& c : g_cache) held += c.n;
    if (held + n <= kCacheMaxBytes) g_cache.push_back({p, n}); else (void)hipFree(p);
    p = nullptr; n = 0;
  }
  void take(DevBuf& o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// ---- host -> device upload of large pageable buffers -------------------------
// Measured on the MI355X box (tools/ubench_pcie.hip, 128 MiB): pageable
// hipMemcpy H2D 17-19 GB/s, pinned 57 GB/s, hipHostRegister 6 ms + 57 GB/s,
// memcpy into pinned memory 30 GB/s with one thread and 60-90 GB/s with 4-8;
// pageable D2H already runs at 55 GB/s.  So uploads above a few MiB go through a
// pooled pinned staging buffer: four threads copy 8 MiB chunks into it and each
// chunk's DMA is queued as soon as it is complete, overlapping the next copy.
struct PinnedBuf {
  void* p = nullptr;
  size_t n = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { release(); }
  static std::vector<CacheEntry>& pool() { static std::vector<CacheEntry> v; return v; }
  bool alloc(size_t bytes) {
    const size_t want = round_size(bytes);
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      auto& v = pool();
      for (size_t i = 0; i < v.size(); ++i)
        if (v[i].n == want) { p = v[i].p; n = want; v.erase(v.begin() + i); return true; }
    }
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
    n = want;
    return true;
  }
  void release() {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    size_t held = 0;
    for (auto& c : pool()) held += c.n;
    if (held + n <= ((size_t)2 << 30)) pool().push_back({p, n}); else (void)hipHostFree(p);
    p = nullptr; n = 0;
  }
};


static const size_t kStageMin = (size_t)1 << 20, kStageChunk = (size_t)8 << 20;
static const int kStageThreads = 4;   // parts per chunk
static const int kPoolThreads = 8;    // helper threads (several transfers can be in flight)

// Persistent helper threads for the host halves of the transfers (copying between
// caller memory and pinned staging).  Leaked on purpose: the threads sleep on the
// condition variable until the process ends.
class HostPool {
 public:
  struct Task { std::function<void(int)> fn; int n = 0; std::atomic<int> next{0}, done{0}; };
  typedef std::shared_ptr<Task> Handle;
  static HostPool& get() { static HostPool* p = new HostPool(kPoolThreads); return *p; }
  // fn(i) for every i in [0, n) on the helper threads, in index order; returns at once
  Handle submit(int n, std::function<void(int)> fn) {
    auto t = std::make_shared<Task>();
    t->fn = std::move(fn); t->n = n;
    { std::lock_guard<std::mutex> lk(mu_); q_.push_back(t); }
    cv_.notify_all();
    return t;
  }
  static void wait(const Handle& t) {
    while (t->done.load(std::memory_order_acquire) < t->n) std::this_thread::yield();
  }

 private:
  explicit HostPool(int helpers) {
    for (int i = 0; i < helpers; ++i) std::thread([this] { loop(); }).detach();
  }
  void loop() {
    for (;;) {
      Handle t;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !q_.empty(); });
        t = q_.front();
        if (t->next.load(std::memory_order_relaxed) >= t->n) { q_.pop_front(); continue; }
      }
      for (;;) {
        const int i = t->next.fetch_add(1, std::memory_order_relaxed);
        if (i >= t->n) break;
        t->fn(i);
        t->done.fetch_add(1, std::memory_order_release);
      }
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Handle> q_;
};

// One transfer = several pageable pieces that sit back to back (at the given
// offsets) in one device arena.  `stage` must outlive the stream work.
struct Piece { void* host; size_t off, len; };

// bytes [lo, hi) of the arena image <-> the pieces that overlap them
static void copy_range(char* stage, const std::vector<Piece>& pieces, size_t lo, size_t hi, bool to_stage) {
  for (const Piece& pc : pieces) {
    const size_t a = std::max(lo, pc.off), e = std::min(hi, pc.off + pc.len);
    if (e <= a) continue;
    if (to_stage) memcpy(stage + a, static_cast<const char*>(pc.host) + (a - pc.off), e - a);
    else memcpy(static_cast<char*>(pc.host) + (a - pc.off), stage + a, e - a);
  }
}
// item i of a transfer = part (i % kStageThreads) of chunk (i / kStageThreads)
static void copy_item(char* stage, const std::vector<Piece>& pieces, size_t bytes, int i, bool to_stage) {
  const size_t c0 = (size_t)(i / kStageThreads) * kStageChunk, clen = std::min(kStageChunk, bytes - c0);
  const size_t part = (clen / kStageThreads + 63) & ~(size_t)63;
  const size_t o = std::min(clen, (size_t)(i % kStageThreads) * part), e = std::min(clen, o + part);
  if (e > o) copy_range(stage, pieces, c0 + o, c0 + e, to_stage);
}

// host -> device: the helpers gather 8 MiB chunks into pinned memory; this thread
// queues a chunk's DMA as soon as its parts are in, so DMA and gathering overlap
static hipError_t upload_pieces(void* dst, const std::vector<Piece>& pieces, size_t bytes, hipStream_t s, PinnedBuf& stage) {
  if (bytes < kStageMin || !stage.alloc(bytes)) {
    for (const Piece& pc : pieces) {
      hipError_t e = hipMemcpyAsync(static_cast<char*>(dst) + pc.off, pc.host, pc.len, hipMemcpyHostToDevice, s);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  const int nchunks = (int)((bytes + kStageChunk - 1) / kStageChunk);
  auto done = std::make_shared<std::vector<std::atomic<int>>>(nchunks);
  for (auto& d : *done) d.store(0);
  char* stg = static_cast<char*>(stage.p);
  const std::vector<Piece>* pcs = &pieces;
  HostPool::Handle h = HostPool::get().submit(nchunks * kStageThreads, [=](int i) {
    copy_item(stg, *pcs, bytes, i, true);
    (*done)[i / kStageThreads].fetch_add(1, std::memory_order_release);
  });
  hipError_t err = hipSuccess;
  for (int c = 0; c < nchunks; ++c) {
    while ((*done)[c].load(std::memory_order_acquire) < kStageThreads) std::this_thread::yield();
    const size_t c0 = (size_t)c * kStageChunk, clen = std::min(kStageChunk, bytes - c0);
    if (err == hipSuccess)
      err = hipMemcpyAsync(static_cast<char*>(dst) + c0, stg + c0, clen, hipMemcpyHostToDevice, s);
  }
  HostPool::wait(h);
  return err;
}

// device -> host in two steps.  issue(): the copy into pinned memory is queued on the
// stream right behind the kernels that produce the data (8 MiB chunks, one event
// each), no host wait.  finish(): once the caller knows which pieces it wants, the
// helpers scatter each chunk to the caller's arrays as its event fires.  (Pageable
// D2H of a few MiB per call runs at 12-17 GB/s here, this path at the DMA rate; and
// results reach caller memory only after the range-check flags have been seen.)
struct Download {
  PinnedBuf stage;
  std::vector<hipEvent_t> ev;
  size_t bytes = 0;
  bool staged = false;
  Download() = default;
  Download(const Download&) = delete;
  Download& operator=(const Download&) = delete;
  ~Download() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }

  hipError_t issue(const void* src, size_t nbytes, hipStream_t s) {
    bytes = nbytes;
    staged = nbytes >= kStageMin && stage.alloc(nbytes);
    if (!staged) return hipSuccess;
    for (size_t c0 = 0; c0 < bytes; c0 += kStageChunk) {
      const size_t clen = std::min(kStageChunk, bytes - c0);
      hipError_t e = hipMemcpyAsync(static_cast<char*>(stage.p) + c0, static_cast<const char*>(src) + c0, clen,
                                    hipMemcpyDeviceToHost, s);
      hipEvent_t evt = nullptr;
      if (e == hipSuccess) e = hipEventCreateWithFlags(&evt, hipEventDisableTiming);
      if (e != hipSuccess) return e;
      ev.push_back(evt);
      if ((e = hipEventRecord(evt, s)) != hipSuccess) return e;
    }
    return hipSuccess;
  }
  // everything queued before issue() on the stream has completed when this returns
  hipError_t wait_first(hipStream_t s) const { return staged ? hipEventSynchronize(ev[0]) : hipStreamSynchronize(s); }

  hipError_t finish(const void* src, const std::vector<Piece>& pieces, hipStream_t s) {
    if (!staged) {
      for (const Piece& pc : pieces) {
        hipError_t e = hipMemcpyAsync(pc.host, static_cast<const char*>(src) + pc.off, pc.len, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return e;
      }
      return hipStreamSynchronize(s);
    }
    // a chunk is handed to the helpers only once it has arrived: a helper never waits
    // for the GPU, so transfers of other host threads are not held up behind this one
    const int nchunks = (int)ev.size();
    char* stg = static_cast<char*>(stage.p);
    const std::vector<Piece>* pcs = &pieces;
    const size_t nbytes = bytes;
    std::vector<HostPool::Handle> hs;
    hipError_t e = hipSuccess;
    for (int c = 0; c < nchunks && e == hipSuccess; ++c) {
      e = hipEventSynchronize(ev[c]);
      if (e == hipSuccess && !pieces.empty())
        hs.push_back(HostPool::get().submit(kStageThreads, [=](int t) { copy_item(stg, *pcs, nbytes, c * kStageThreads + t, false); }));
    }
    for (auto& h : hs) HostPool::wait(h);
    return e;
  }
};

// copy `bytes` from pageable `src` to device `dst` on `s`
static hipError_t upload(void* dst, const void* src, size_t bytes, hipStream_t s, PinnedBuf& stage) {
  return upload_pieces(dst, std::vector<Piece>{{const_cast<void*>(src), 0, bytes}}, bytes, s, stage);
}

struct Streams {
  hipStream_t s[3] = {nullptr, nullptr, nullptr};
  hipEvent_t luma_done = nullptr;
  ~Streams() {
    for (auto& x : s) if (x) (void)hipStreamDestroy(x);
    if (luma_done) (void)hipEventDestroy(luma_done);
  }
};

static std::vector<Streams*> g_stream_pool;

struct StreamLease {     // borrow a ready-made set of streams, give it back on scope exit
  Streams* p = nullptr;
  StreamLease() {
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      if (!g_stream_pool.empty()) { p = g_stream_pool.back(); g_stream_pool.pop_back(); return; }
    }
    Streams* n = new (std::nothrow) Streams;
    if (!n) return;
    bool ok = true;
    for (int i = 0; i < 3 && ok; ++i) ok = hipStreamCreateWithFlags(&n->s[i], hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&n->luma_done, hipEventDisableTiming) == hipSuccess;
    if (!ok) { delete n; return; }
    p = n;
  }
  ~StreamLease() {
    if (!p) return;
    for (auto& x : p->s) (void)hipStreamSynchronize(x);   // nothing of this job may outlive it
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_stream_pool.push_back(p);
  }
};

struct Comp {            // per-component device state (kept until the job ends)
  DevBuf coef, plane, cst, status, up, px;
  PinnedBuf stage;           // pinned upload staging, held until the job's streams are drained
  PinnedBuf hstatus;         // range-check flag on its way back
  Download down, down_up;    // results on their way back
  bool processed = false, dequant_only = false, have_up = false;
  hipStream_t stream = nullptr;
};

enum { JOB_RERUN_CAREFUL = -1000 };

static double wall_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static bool trace_on() { static const bool on = getenv("QS_HIP_TRACE") != nullptr; return on; }

static int run_job(qs_hip_job* job, int flags, int niter, int progprec,
                   qs_hip_progress_fn progress, void* userdata, bool eager) {
  int need_lowres = 0, stop = 0;
  if ((flags & (QS_JOINT_YUV | QS_UPSAMPLE_UV)) && job->colorspace == 3 && job->ncomp >= 3 &&
      job->hsamp[1] == 1 && job->vsamp[1] == 1 && job->hsamp[2] == 1 && job->vsamp[2] == 1)
    need_lowres = 1;                                     // reference :2447-2453

  // streams/events are pooled too (creating three streams costs ~1 ms)
  StreamLease lease;
  if (!lease.p) return fail(QS_HIP_ENODEV, "could not create HIP streams: %s", hipGetErrorString(hipGetLastError()));
  Streams& st = *lease.p;
  const int nstreams = eager ? 3 : 1;

  int prog_next = 0, prog_max = 0, prog_thr = 0;
  if (progress) {                                        // reference :2474-2482
    for (int ci = 0; ci < job->ncomp; ++ci) prog_max += job->hblk[ci] * job->vsamp[ci] * niter;
    if (progprec == 0) progprec = 20;
    if (progprec < 0) progprec = prog_max;
    prog_thr = (int)((unsigned)(prog_max + progprec - 1) / (unsigned)progprec);
  }

  QsConsts* hc = new (std::nothrow) QsConsts[QS_HIP_MAXC];   // one per component: uploads are async
  if (!hc) return fail(QS_HIP_ENOMEM, "out of host memory");
  struct HcFree { QsConsts* p; ~HcFree() { delete[] p; } } hc_free{hc};

  const double t_start = wall_ms();
  double t_upload = 0;
  Comp comp[QS_HIP_MAXC];
  // planes that outlive their component (reference image1 / image2, :2753-2815)
  DevBuf d_yfull, d_llow;          // full-res luma plane; luma at chroma resolution
  bool have_yfull = false, have_llow = false;
  int16_t* up_host[2] = { nullptr, nullptr };
  struct UpFree { int16_t** p; bool keep; ~UpFree() { if (!keep) { free(p[0]); free(p[1]); } } } up_free{up_host, false};

  for (int ci = 0; ci < job->ncomp; ++ci) {
    Comp& C = comp[ci];
    const int wb = job->wblk[ci], hb = job->hblk[ci];
    const size_t nblk = (size_t)wb * hb, cbytes = nblk * 64 * sizeof(int16_t);
    int iters = niter, extra = 0;
    int prog_cur = prog_next;
    const int prog_inc = job->vsamp[ci];
    const int luma = !ci || job->colorspace != 3;        // reference :2639
    prog_next += hb * prog_inc * niter;
    if (!job->has_quant[ci]) continue;                   // reference :2493
    if (have_yfull || (!ci && need_lowres)) extra = 1;   // reference :2495

    int acc = 0;
    for (int i = 0; i < 64; ++i) acc |= job->quant[ci][i];
    if (acc <= 1) iters = 0;                             // reference :2501
    if (acc >= 0x800) stop = 1;                          // reference :2504
    if (iters + extra == 0) continue;                    // reference :2542

    // stream: luma (and anything coupled to it) on stream 0; independent
    // components round-robin
    hipStream_t s = st.s[eager ? ci % nstreams : 0];
    C.stream = s; C.processed = true;
    HIP_TRY(C.coef.alloc(cbytes));
    HIP_TRY(C.cst.alloc(sizeof(QsConsts)));
    HIP_TRY(C.status.alloc(sizeof(int32_t)));
    if (int r = qs_hip_consts_build(&hc[ci], job->quant[ci], flags)) return r;
    HIP_TRY(hipMemcpyAsync(C.cst.p, &hc[ci], sizeof(QsConsts), hipMemcpyHostToDevice, s));
    { const double t0 = wall_ms(); HIP_TRY(upload(C.coef.p, job->coef[ci], cbytes, s, C.stage)); t_upload += wall_ms() - t0; }
    HIP_TRY(hipMemsetAsync(C.status.p, 0, sizeof(int32_t), s));

    bool have_plane = false;
    if (!stop) {
      // the reference falls back to dequantise-only when the plane cannot be
      // allocated (reference :2551-2566); same here for device memory
      hipError_t e = C.plane.alloc(qs_hip_plane_bytes(wb, hb));
      if (e == hipSuccess) have_plane = true; else (void)hipGetLastError();
    }
    if (!have_plane) {
      C.dequant_only = true;
      if (int r = qs_hip_dequant_plane(C.cst.p, C.coef.as<int16_t>(), wb, hb, s)) return r;
      continue;
    }
    if (eager && ci > 0 && (have_llow || have_yfull))    // chroma reads planes produced on the luma stream
      HIP_TRY(hipStreamWaitEvent(s, st.luma_done, 0));

    const int rebalance = !(flags & QS_NO_REBALANCE) && (luma || !(flags & QS_NO_REBALANCE_UV));  // :1567-1568
    // JOINT_YUV acts through the low-res luma plane only (reference :2636)
    const bool joint = have_llow && (flags & QS_JOINT_YUV);
    const int plane_flags = flags & (QS_DIAGONALS | QS_NO_REBALANCE | QS_NO_REBALANCE_UV);
    bool clamped = false;
    for (int it = 0; it < iters + extra; ++it) {
      if (int r = qs_hip_idct_plane(C.cst.p, C.coef.as<int16_t>(), C.plane.as<uint8_t>(), wb, hb,
                                    it == 0, 1, 1, C.status.as<int32_t>(), s)) return r;
      if (it == 0 && !eager) {                           // reference :2610
        int32_t bad = 0;
        HIP_TRY(hipMemcpyAsync(&bad, C.status.p, sizeof(bad), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (bad) { stop = 1; break; }
      }
      if (it == iters) break;                            // refresh-only pass, reference :2622
      // pass B.  The +-1023 clamp rides on the last launch of the last iteration --
      // unless a refresh-only pass A follows: the reference clamps after its loop
      // (:2668-2689), so that refresh (the planes JOINT_YUV / UPSAMPLE_UV read) is the
      // IDCT of the unclamped coefficients.
      const int last = (it == iters - 1) && !extra;
      if (flags & QS_LOW_QUALITY) {                      // reference :924-938: never reaches the k-loop
        if (joint) {
          if (int r = qs_hip_joint_plane(C.cst.p, C.coef.as<int16_t>(), C.plane.as<uint8_t>(), d_llow.as<uint8_t>(),
                                         wb, hb, rebalance, last, s)) return r;
        } else {
          if (int r = qs_hip_lowq_plane(C.cst.p, C.coef.as<int16_t>(), C.plane.as<uint8_t>(), wb, hb,
                                        rebalance, last, s)) return r;
        }
      } else {
        if (joint)
          if (int r = qs_hip_joint_plane(C.cst.p, C.coef.as<int16_t>(), C.plane.as<uint8_t>(), d_llow.as<uint8_t>(),
                                         wb, hb, 0, 0, s)) return r;
        if (int r = qs_hip_smooth_plane(C.cst.p, C.coef.as<int16_t>(), C.plane.as<uint8_t>(), wb, hb,
                                        plane_flags, luma, last, s)) return r;
      }
      if (last) clamped = true;
      if (progress) {                                    // reference :2656-2664
        int cur = prog_cur += hb * prog_inc;
        if (cur >= prog_thr) {
          cur = (int)((long long)progprec * cur / prog_max);
          prog_thr = (int)(((long long)(cur + 1) * prog_max + progprec - 1) / progprec);
          HIP_TRY(hipStreamSynchronize(s));              // the pass is done when we report it
          stop = progress(userdata, cur, progprec);
        }
        if (stop) break;
      }
    }
    if (!clamped)                                        // reference :2668-2689
      if (int r = qs_hip_clamp_plane(C.coef.as<int16_t>(), wb, hb, s)) return r;

    if (!stop && have_yfull) {
      // UPSAMPLE_UV: chroma -> luma resolution, re-encoded (reference :2691-2752)
      const int ws = job->hsamp[0], hs = job->vsamp[0];
      const int uwb = job->wblk[0], uhb = job->hblk[0];
      const size_t ubytes = (size_t)uwb * uhb * 64 * sizeof(int16_t);
      HIP_TRY(C.px.alloc(qs_hip_upsample_bytes(job->image_width, job->image_height, ws, hs)));
      HIP_TRY(C.up.alloc(ubytes));
      up_host[ci - 1] = static_cast<int16_t*>(malloc(ubytes));
      if (!up_host[ci - 1]) return fail(QS_HIP_ENOMEM, "out of host memory");
      if (int r = qs_hip_upsample_plane(C.plane.as<uint8_t>(), d_llow.as<uint8_t>(), wb, d_yfull.as<uint8_t>(),
                                        uwb, uhb, C.px.as<uint8_t>(), job->image_width, job->image_height,
                                        ws, hs, s)) return r;
      if (int r = qs_hip_fdct_plane(C.px.as<uint8_t>(), qs_hip_upsample_pitch(job->image_width, ws),
                                    C.up.as<int16_t>(), uwb, uhb, s)) return r;
      C.have_up = true;
    } else if (!stop && !ci && need_lowres) {
      // keep luma for the chroma passes (reference :2753-2815)
      const int ws = job->hsamp[0], hs = job->vsamp[0];
      if (ws == 1 && hs == 1) {
        d_llow.take(C.plane); have_llow = true;          // image2 = image
      } else {
        DevBuf d_l;
        HIP_TRY(d_l.alloc(qs_hip_plane_bytes(job->wblk[1], job->hblk[1])));
        if (int r = qs_hip_downsample_plane(C.plane.as<uint8_t>(), wb, hb, d_l.as<uint8_t>(),
                                            job->wblk[1], job->hblk[1], ws, hs, s)) return r;
        d_llow.take(d_l); have_llow = true;
        if (flags & QS_UPSAMPLE_UV) { d_yfull.take(C.plane); have_yfull = true; }   // image1 = image
      }
      HIP_TRY(hipEventRecord(st.luma_done, s));
    }
    if (!eager) HIP_TRY(hipStreamSynchronize(s));
  }

  // ---- behind each component's kernels: range-check flag and results into pinned memory
  const size_t ubytes = (size_t)job->wblk[0] * job->hblk[0] * 64 * sizeof(int16_t);
  for (int ci = 0; ci < job->ncomp; ++ci) {
    Comp& C = comp[ci];
    if (!C.processed) continue;
    const size_t cbytes = (size_t)job->wblk[ci] * job->hblk[ci] * 64 * sizeof(int16_t);
    if (eager && !C.dequant_only) {
      if (!C.hstatus.alloc(sizeof(int32_t))) return fail(QS_HIP_ENOMEM, "out of pinned host memory");
      HIP_TRY(hipMemcpyAsync(C.hstatus.p, C.status.p, sizeof(int32_t), hipMemcpyDeviceToHost, C.stream));
    }
    HIP_TRY(C.down.issue(C.coef.p, cbytes, C.stream));
    if (C.have_up && !stop) HIP_TRY(C.down_up.issue(C.up.p, ubytes, C.stream));
  }

  // ---- everything is enqueued; eager mode reads the range-check flags now
  const double t_enq = wall_ms();
  for (int i = 0; i < nstreams; ++i) HIP_TRY(hipStreamSynchronize(st.s[i]));
  const double t_done = wall_ms();
  if (eager)
    for (int ci = 0; ci < job->ncomp; ++ci)
      if (comp[ci].processed && !comp[ci].dequant_only && *static_cast<const int32_t*>(comp[ci].hstatus.p))
        return JOB_RERUN_CAREFUL;                          // host input is still untouched

  // ---- scatter the results (the only place host memory is written)
  for (int ci = 0; ci < job->ncomp; ++ci) {
    Comp& C = comp[ci];
    if (!C.processed) continue;
    const size_t cbytes = (size_t)job->wblk[ci] * job->hblk[ci] * 64 * sizeof(int16_t);
    HIP_TRY(C.down.finish(C.coef.p, std::vector<Piece>{{job->coef[ci], 0, cbytes}}, C.stream));
    if (C.have_up && !stop)
      HIP_TRY(C.down_up.finish(C.up.p, std::vector<Piece>{{up_host[ci - 1], 0, ubytes}}, C.stream));
  }
  if (trace_on())
    fprintf(stderr, "qs_hip trace: %s  enqueue %.2f ms (host->pinned->device issue %.2f)  drain %.2f ms  scatter %.2f ms\n",
            eager ? "eager" : "careful", t_enq - t_start, t_upload, t_done - t_enq, wall_ms() - t_done);

  if (!stop && have_yfull && up_host[0] && up_host[1]) {  // reference :2836-2849
    job->coef_up[0] = up_host[0]; job->coef_up[1] = up_host[1]; up_free.keep = true;
    job->up_wblk = job->wblk[0]; job->up_hblk = job->hblk[0];
    job->out_hsamp0 = job->out_vsamp0 = 1;
  }
  for (int ci = 0; ci < job->ncomp; ++ci)                // reference :2851-2859
    if (job->has_quant[ci]) for (int i = 0; i < 64; ++i) job->quant[ci][i] = 1;
  return stop;
}


// ---------------------------------------------------------------------------
// fused execution: jobs whose components are independent of each other (no
// JOINT_YUV / UPSAMPLE_UV coupling, no LOW_QUALITY, ordinary quant tables) run
// as plane sets -- ONE pass-A and ONE pass-B launch per iteration for all
// components of all jobs of a group (qs_*_set_kernel), so that small images
// fill the chip together and a job does not occupy three hardware queues.
// Everything else about the job semantics is as in run_job (eager mode): the
// range-check flags are read once at the end, a job with a set flag is re-run in
// the careful order from its untouched host input.

static int comp_rebalance(const qs_hip_job* job, int ci, int flags) {
  const int luma = !ci || job->colorspace != 3;                                     // reference :2639
  return !(flags & QS_NO_REBALANCE) && (luma || !(flags & QS_NO_REBALANCE_UV));       // :1567-1568
}

static bool job_needs_lowres(const qs_hip_job* job, int flags) {                     // reference :2447-2453
  return (flags & (QS_JOINT_YUV | QS_UPSAMPLE_UV)) && job->colorspace == 3 && job->ncomp >= 3 &&
         job->hsamp[1] == 1 && job->vsamp[1] == 1 && job->hsamp[2] == 1 && job->vsamp[2] == 1;
}

static bool job_fusable(const qs_hip_job* job, int flags) {
  static const bool off = getenv("QS_HIP_NO_FUSE") != nullptr;
  if (off || (flags & QS_LOW_QUALITY) || job_needs_lowres(job, flags)) return false;
  for (int ci = 0; ci < job->ncomp; ++ci) {
    if (!job->has_quant[ci]) return false;
    int acc = 0;
    for (int i = 0; i < 64; ++i) acc |= job->quant[ci][i];
    if (acc <= 1 || acc >= 0x800) return false;          // iterations skipped / stop: the general path knows how
  }
  return true;
}

// One device plane of a set: a whole component, or a band of block rows of a very large
// one (rows [src_row0, src_row0 + hb) of the source, of which [keep0, keep1) are results:
// the rest is halo, see split_rows).
struct FPlane { int job, ci, wb, hb, cst; size_t coef_off, px_off, cbytes; int src_row0, keep0, keep1; };
struct FGroup {
  std::vector<FPlane> planes;
  std::vector<int> jobs;                  // indices into the caller's job list
  DevBuf coef, px, cst, status;
  PinnedBuf stage;
  std::vector<QsConsts> hc;               // host copies stay alive until the stream is drained
  PinnedBuf hstatus;                      // range-check flags
  Download down;                          // results on their way back
  hipStream_t s = nullptr;
  size_t blocks = 0, coef_bytes = 0;
};
struct DrainGuard {                       // error paths: nothing may be freed while the streams still run
  Streams* st;
  ~DrainGuard() { for (auto& x : st->s) (void)hipStreamSynchronize(x); }
};

// a group of >= 3 waves per SIMD runs at the streaming rate; smaller groups let the upload of one
// overlap the kernels of the previous and the download of the one before (three streams)
static const size_t kGroupBlocks = (size_t)200 << 10;
// A plane above kSplitBlocks is cut into bands of about kBandBlocks that travel as separate
// groups, so its upload, kernels and download overlap as they do for a batch of small jobs.
// A block's result after n iterations depends only on blocks within n rows of it, so a band
// carries n extra block rows on each cut side (recomputed, not copied back): bit-exact.
// (QS_HIP_SPLIT_BLOCKS / QS_HIP_BAND_BLOCKS override the two sizes: the tests use them to run
// the band logic on small images.)
static size_t env_size(const char* name, size_t dflt) {
  const char* v = getenv(name);
  const long long n = v ? atoll(v) : 0;
  return n > 0 ? (size_t)n : dflt;
}
static const size_t kSplitBlocks = env_size("QS_HIP_SPLIT_BLOCKS", (size_t)512 << 10),
                    kBandBlocks = env_size("QS_HIP_BAND_BLOCKS", (size_t)256 << 10);

static int run_fused(qs_hip_job* const* jobs, const std::vector<int>& which, int flags, int niter, int* results) {
  StreamLease lease;
  if (!lease.p) return fail(QS_HIP_ENODEV, "could not create HIP streams: %s", hipGetErrorString(hipGetLastError()));
  std::list<FGroup> groups;
  DrainGuard drain{lease.p};
  const double t_start = wall_ms();

  // ---- partition into groups (a job never straddles two, unless it is cut into bands)
  int maxj = 0;
  for (int ji : which) maxj = std::max(maxj, ji);
  std::vector<char> split(maxj + 1, 0), bad_job(maxj + 1, 0), scattered(maxj + 1, 0), defer(maxj + 1, 0);
  for (int ji : which) {
    const qs_hip_job* job = jobs[ji];
    size_t jblocks = 0;
    bool big = false;
    for (int ci = 0; ci < job->ncomp; ++ci) {
      const size_t nb = (size_t)job->wblk[ci] * job->hblk[ci];
      jblocks += nb;
      const int bands = (int)((nb + kBandBlocks - 1) / kBandBlocks);
      if (nb > kSplitBlocks && (job->hblk[ci] + bands - 1) / bands >= 8 * niter) big = true;   // halo <= 25 %
    }
    if (big) {
      split[ji] = 1;
      for (int ci = 0; ci < job->ncomp; ++ci) {
        const int wb = job->wblk[ci], hb = job->hblk[ci];
        const int bands = std::max(1, (int)(((size_t)wb * hb + kBandBlocks - 1) / kBandBlocks));
        const int rows = (hb + bands - 1) / bands;
        for (int r0 = 0; r0 < hb; r0 += rows) {
          const int r1 = std::min(hb, r0 + rows), d0 = std::max(0, r0 - niter), d1 = std::min(hb, r1 + niter);
          groups.emplace_back();
          FGroup& G = groups.back();
          G.jobs.push_back(ji);
          G.blocks = (size_t)wb * (d1 - d0);
          G.planes.push_back({ji, ci, wb, d1 - d0, -1, 0, 0, (size_t)wb * (d1 - d0) * 128, d0, r0 - d0, r1 - d0});
        }
      }
      groups.emplace_back();                                 // the next job starts a fresh group
      continue;
    }
    if (groups.empty() || (int)groups.back().planes.size() + job->ncomp > QS_MAX_PLANES ||
        (groups.back().blocks && groups.back().blocks + jblocks > kGroupBlocks))
      groups.emplace_back();
    FGroup& G = groups.back();
    G.jobs.push_back(ji);
    G.blocks += jblocks;
    for (int ci = 0; ci < job->ncomp; ++ci)
      G.planes.push_back({ji, ci, job->wblk[ci], job->hblk[ci], -1, 0, 0, (size_t)job->wblk[ci] * job->hblk[ci] * 128,
                          0, 0, job->hblk[ci]});
  }
  groups.remove_if([](const FGroup& g) { return g.planes.empty(); });   // placeholders left by band jobs

  // ---- enqueue every group: upload, niter x (pass A, pass B), status readback
  const int diag = (flags & QS_DIAGONALS) != 0;
  size_t gi = 0;
  for (FGroup& G : groups) {
    G.s = lease.p->s[gi++ % 3];
    const int np = (int)G.planes.size();
    size_t coef_bytes = 0, px_bytes = 0;
    std::vector<const uint16_t*> qtabs;
    for (FPlane& P : G.planes) {
      P.coef_off = coef_bytes; coef_bytes += P.cbytes;
      P.px_off = px_bytes; px_bytes += (qs_hip_plane_bytes(P.wb, P.hb) + 255) & ~(size_t)255;
      const uint16_t* q = jobs[P.job]->quant[P.ci];
      for (size_t k = 0; k < qtabs.size() && P.cst < 0; ++k)
        if (!memcmp(qtabs[k], q, 64 * sizeof(uint16_t))) P.cst = (int)k;
      if (P.cst < 0) { P.cst = (int)qtabs.size(); qtabs.push_back(q); }
    }
    HIP_TRY(G.coef.alloc(coef_bytes));
    HIP_TRY(G.px.alloc(px_bytes));
    HIP_TRY(G.cst.alloc(qtabs.size() * sizeof(QsConsts)));
    HIP_TRY(G.status.alloc((size_t)np * sizeof(int32_t)));
    G.hc.resize(qtabs.size());
    for (size_t k = 0; k < qtabs.size(); ++k)
      if (int r = qs_hip_consts_build(&G.hc[k], qtabs[k], flags)) return r;
    HIP_TRY(hipMemcpyAsync(G.cst.p, G.hc.data(), qtabs.size() * sizeof(QsConsts), hipMemcpyHostToDevice, G.s));
    std::vector<Piece> pieces;
    for (const FPlane& P : G.planes)
      pieces.push_back({jobs[P.job]->coef[P.ci] + (size_t)P.src_row0 * P.wb * 64, P.coef_off, P.cbytes});
    G.coef_bytes = coef_bytes;
    HIP_TRY(upload_pieces(G.coef.p, pieces, coef_bytes, G.s, G.stage));
    HIP_TRY(hipMemsetAsync(G.status.p, 0, (size_t)np * sizeof(int32_t), G.s));

    QsPlaneSet set;
    memset(&set, 0, sizeof set);
    set.n = np;
    int w = 0;
    for (int i = 0; i < np; ++i) {
      const FPlane& P = G.planes[i];
      set.wave0[i] = w;
      w += (P.wb * P.hb + 63) / 64;
      QsPlaneRef& R = set.ref[i];
      R.cst = G.cst.as<QsConsts>() + P.cst;
      R.coef = reinterpret_cast<int16_t*>(G.coef.as<char>() + P.coef_off);
      R.plane = G.px.as<uint8_t>() + P.px_off;
      R.status = G.status.as<int32_t>() + i;
      R.wblk = P.wb; R.hblk = P.hb; R.pitch = qs_plane_pitch(P.wb);
      R.rebalance = comp_rebalance(jobs[P.job], P.ci, flags);
    }
    for (int i = np; i < QS_MAX_PLANES + 2; ++i) set.wave0[i] = w;
    for (int it = 0; it < niter; ++it) {
      qs_launch_idct_set(set, it == 0, G.s);
      qs_launch_smooth_set(set, diag, it == niter - 1, G.s);
    }
    HIP_TRY(hipGetLastError());
    // pinned: a pageable destination would make this call wait for the whole stream
    if (!G.hstatus.alloc((size_t)np * sizeof(int32_t))) return fail(QS_HIP_ENOMEM, "out of pinned host memory");
    HIP_TRY(hipMemcpyAsync(G.hstatus.p, G.status.p, (size_t)np * sizeof(int32_t), hipMemcpyDeviceToHost, G.s));
    HIP_TRY(G.down.issue(G.coef.p, coef_bytes, G.s));       // to pinned memory, right behind the kernels
  }
  const double t_enq = wall_ms();

  // ---- drain group by group; results go back only for jobs whose range check passed.
  // A job cut into bands is scattered band by band before its later bands have been
  // checked: should one of those trip the range check after all (crafted file), the rows
  // already written are restored from the pinned upload staging, which still holds the
  // original input.  Without that staging copy (pinned memory exhausted) the job's bands
  // are held back until all of them have been checked.
  auto result_piece = [&](const FPlane& P) {
    const size_t row = (size_t)P.wb * 128;
    return Piece{jobs[P.job]->coef[P.ci] + (size_t)(P.src_row0 + P.keep0) * P.wb * 64,
                 P.coef_off + P.keep0 * row, (size_t)(P.keep1 - P.keep0) * row};
  };
  for (FGroup& G : groups)
    if (!G.stage.p) for (int ji : G.jobs) if (split[ji]) defer[ji] = 1;
  std::vector<FGroup*> held;
  for (FGroup& G : groups) {
    HIP_TRY(G.down.wait_first(G.s));
    const int32_t* hst = static_cast<const int32_t*>(G.hstatus.p);
    for (size_t i = 0; i < G.planes.size(); ++i) if (hst[i]) bad_job[G.planes[i].job] = 1;
    bool hold = false;
    for (int ji : G.jobs) hold |= (defer[ji] != 0);
    if (hold) { held.push_back(&G); continue; }
    std::vector<Piece> back;
    for (const FPlane& P : G.planes)
      if (!bad_job[P.job]) { back.push_back(result_piece(P)); scattered[P.job] = 1; }
    HIP_TRY(G.down.finish(G.coef.p, back, G.s));
  }
  for (FGroup* G : held) {
    std::vector<Piece> back;
    for (const FPlane& P : G->planes) if (!bad_job[P.job]) back.push_back(result_piece(P));
    HIP_TRY(G->down.finish(G->coef.p, back, G->s));
  }
  std::vector<int> rerun;
  for (int ji : which) {
    if (!bad_job[ji]) { results[ji] = 0; continue; }
    rerun.push_back(ji);
    if (!scattered[ji]) continue;                            // host input is still untouched
    for (FGroup& G : groups)                                 // put the original rows back
      for (const FPlane& P : G.planes)
        if (P.job == ji && G.stage.p) {
          const Piece pc = result_piece(P);
          memcpy(pc.host, static_cast<const char*>(G.stage.p) + pc.off, pc.len);
        }
  }
  if (trace_on())
    fprintf(stderr, "qs_hip trace: fused  %zu job(s) in %zu group(s)  enqueue %.2f ms  drain+download %.2f ms  (%zu re-run)\n",
            which.size(), groups.size(), t_enq - t_start, wall_ms() - t_enq, rerun.size());
  for (int ji : which) {
    if (bad_job[ji]) continue;
    for (int ci = 0; ci < jobs[ji]->ncomp; ++ci)           // reference :2851-2859
      for (int i = 0; i < 64; ++i) jobs[ji]->quant[ci][i] = 1;
  }
  const double t_clear = wall_ms();
  groups.clear();                                            // give the arenas back before the re-runs allocate
  if (trace_on()) fprintf(stderr, "qs_hip trace: fused  release %.2f ms\n", wall_ms() - t_clear);
  for (int ji : rerun)
    results[ji] = run_job(jobs[ji], flags, niter, 0, nullptr, nullptr, /*eager=*/false);
  return QS_HIP_OK;
}

}  // namespace

extern "C" void qs_hip_release_cache(void) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  for (auto& c : g_cache) (void)hipFree(c.p);
  g_cache.clear();
  for (auto* sp : g_stream_pool) delete sp;
  g_stream_pool.clear();
  for (auto& c : PinnedBuf::pool()) (void)hipHostFree(c.p);
  PinnedBuf::pool().clear();
}

// validation and the reference's early-outs; returns 1 when there is work to do,
// 0 when the job is already finished (result 0), < 0 on a bad job
static int prepare_job(qs_hip_job* job, int flags, int* niter) {
  if (!job || job->ncomp < 1 || job->ncomp > QS_HIP_MAXC)
    return fail(QS_HIP_EINVAL, "qs_hip_do_quantsmooth: bad job");
  for (int ci = 0; ci < job->ncomp; ++ci)
    if (!job->coef[ci] || job->wblk[ci] <= 0 || job->hblk[ci] <= 0)
      return fail(QS_HIP_EINVAL, "qs_hip_do_quantsmooth: component %d has no data", ci);
  job->up_wblk = job->up_hblk = 0; job->coef_up[0] = job->coef_up[1] = nullptr;
  job->out_hsamp0 = job->hsamp[0]; job->out_vsamp0 = job->vsamp[0];
  if (*niter < 0) *niter = 0;
  if (*niter > 100) *niter = 100;                          // reference :2455-2456
  if (*niter <= 0 && !((flags & QS_UPSAMPLE_UV) && job_needs_lowres(job, flags))) return 0;  // reference :2458
  return 1;
}

extern "C" int qs_hip_do_quantsmooth(qs_hip_job* job, int flags, int niter, int progprec,
                                     qs_hip_progress_fn progress, void* userdata) {
  const int todo = prepare_job(job, flags, &niter);
  if (todo <= 0) return todo;
  if (qs_hip_device_count() <= 0)
    return fail(QS_HIP_ENODEV, "no HIP device available (this library has no CPU fallback)");

  if (!progress && job_fusable(job, flags)) {
    int result = QS_HIP_ENODEV;
    qs_hip_job* one[1] = { job };
    if (int r = run_fused(one, std::vector<int>{0}, flags, niter, &result)) return r;
    return result;
  }
  int r = run_job(job, flags, niter, progprec, progress, userdata, /*eager=*/progress == nullptr);
  if (r == JOB_RERUN_CAREFUL)
    r = run_job(job, flags, niter, progprec, progress, userdata, /*eager=*/false);
  return r;
}

extern "C" int qs_hip_do_quantsmooth_batch(qs_hip_job* const* jobs, int njobs, int flags, int niter, int* results) {
  if (!jobs || !results || njobs < 0) return fail(QS_HIP_EINVAL, "qs_hip_do_quantsmooth_batch: null argument");
  std::vector<int> fused, single;
  const int nit = niter < 0 ? 0 : niter > 100 ? 100 : niter;     // reference :2455-2456
  for (int j = 0; j < njobs; ++j) {
    int n1 = niter;
    const int todo = prepare_job(jobs[j], flags, &n1);
    results[j] = todo < 0 ? todo : 0;
    if (todo <= 0) continue;
    (job_fusable(jobs[j], flags) ? fused : single).push_back(j);
  }
  if (fused.empty() && single.empty()) return QS_HIP_OK;
  if (qs_hip_device_count() <= 0)
    return fail(QS_HIP_ENODEV, "no HIP device available (this library has no CPU fallback)");
  if (!fused.empty()) {
    for (int j : fused) results[j] = QS_HIP_ENODEV;
    const double t0 = wall_ms();
    const int r = run_fused(jobs, fused, flags, nit, results);
    if (trace_on()) fprintf(stderr, "qs_hip trace: batch  run_fused total %.2f ms\n", wall_ms() - t0);
    if (r) return r;
  }
  for (int j : single)                                       // coupled / special jobs: the general path, one by one
    results[j] = qs_hip_do_quantsmooth(jobs[j], flags, niter, 0, nullptr, nullptr);
  return QS_HIP_OK;
}
