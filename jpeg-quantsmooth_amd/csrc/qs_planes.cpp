// qs_planes.cpp -- plane layer of the flat C ABI (include/jpegqs_hip.h): thin argument
// checking + one kernel launch per call, on device pointers and an explicit stream.
#include "qs_common.h"

// ---------------------------------------------------------------------------
// plane layer

static int check_plane_args(const void* a, const void* b, int wblk, int hblk, const char* who) {
  if (!a || !b) return qs_fail(QS_HIP_EINVAL, "%s: null device pointer", who);
  if (wblk <= 0 || hblk <= 0 || (long long)wblk * hblk > (1ll << 28))
    return qs_fail(QS_HIP_EINVAL, "%s: bad plane size %dx%d blocks", who, wblk, hblk);
  return 0;
}

// ---------------------------------------------------------------------------
// band arithmetic: ONE definition for the in-process multi-GPU route (qs_shard.cpp: peer copies) and the
// one-process-per-GPU driver (bands.py: RCCL send/recv).  What differs between the two is the transport.

extern "C" int qs_hip_band_rows(int hblk, int nbands, int band, int align, int* row0, int* row1) {
  if (hblk < 0 || nbands < 1 || band < 0 || band >= nbands || align < 1 || !row0 || !row1)
    return qs_fail(QS_HIP_EINVAL, "qs_hip_band_rows: bad argument");
  const long long units = ((long long)hblk + align - 1) / align;          // band edges on multiples of `align` block rows
  const long long a = units * band / nbands * align, b = units * (band + 1) / nbands * align;
  *row0 = (int)(a < hblk ? a : hblk);
  *row1 = (int)(b < hblk ? b : hblk);
  return QS_HIP_OK;
}

extern "C" int qs_hip_colour_band_rows(int hblk_luma, int hblk_chroma, int v_samp, int nbands, int band,
                                       int* y0, int* y1, int* c0, int* c1) {
  if (v_samp < 1 || !y0 || !y1 || !c0 || !c1) return qs_fail(QS_HIP_EINVAL, "qs_hip_colour_band_rows: bad argument");
  if (int r = qs_hip_band_rows(hblk_chroma, nbands, band, 1, c0, c1)) return r;   // cut on chroma block rows
  const long long a = (long long)*c0 * v_samp, b = (long long)*c1 * v_samp;        // luma: the same image rows
  *y0 = (int)(a < hblk_luma ? a : hblk_luma);
  *y1 = band == nbands - 1 ? hblk_luma : (int)(b < hblk_luma ? b : hblk_luma);
  return QS_HIP_OK;
}

extern "C" int qs_hip_band_halo_rows(int wblk, int hblk, size_t* send_top, size_t* send_bot,
                                     size_t* recv_top, size_t* recv_bot, size_t* nbytes) {
  if (wblk <= 0 || hblk <= 0) return qs_fail(QS_HIP_EINVAL, "qs_hip_band_halo_rows: bad plane size");
  if (send_top) *send_top = qs_hip_plane_row_offset(wblk, 0);               // first pixel row -> the band above
  if (send_bot) *send_bot = qs_hip_plane_row_offset(wblk, hblk * 8 - 1);    // last pixel row  -> the band below
  if (recv_top) *recv_top = qs_hip_plane_row_offset(wblk, -1);              // apron rows: what the neighbours sent
  if (recv_bot) *recv_bot = qs_hip_plane_row_offset(wblk, hblk * 8);
  if (nbytes) *nbytes = qs_hip_plane_pitch(wblk);                           // a whole row, apron columns included
  return QS_HIP_OK;
}

static int launch_status(const char* who) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return qs_fail(QS_HIP_ENODEV, "%s: launch failed: %s", who, hipGetErrorString(e));
  return QS_HIP_OK;
}

extern "C" int qs_hip_idct_plane(const void* d_consts, int16_t* d_coef, uint8_t* d_plane,
                                 int wblk, int hblk, int first, int rep_top, int rep_bot,
                                 int32_t* d_status, void* stream) {
  if (int r = check_plane_args(d_coef, d_plane, wblk, hblk, "qs_hip_idct_plane")) return r;
  if (!d_consts || (first && !d_status)) return qs_fail(QS_HIP_EINVAL, "qs_hip_idct_plane: null consts/status");
  qs_launch_idct_plane(static_cast<const QsConsts*>(d_consts), d_coef, d_plane, wblk, hblk,
                       first, rep_top, rep_bot, d_status, static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_idct_plane");
}

static int smooth_rows(const void* d_consts, int16_t* d_coef, const uint8_t* d_plane, int wblk, int hblk,
                       int row0, int row1, int flags, int luma, int final_clamp, void* stream, const char* who,
                       uint8_t* d_next = nullptr, int rep_top = 1, int rep_bot = 1) {
  if (int r = check_plane_args(d_coef, d_plane, wblk, hblk, who)) return r;
  if (d_next == d_plane) return qs_fail(QS_HIP_EINVAL, "%s: the next plane must be a different buffer (other blocks still read the current one)", who);
  if (!d_consts) return qs_fail(QS_HIP_EINVAL, "%s: null consts", who);
  if (row0 < 0 || row1 > hblk || row0 > row1) return qs_fail(QS_HIP_EINVAL, "%s: bad row range %d..%d", who, row0, row1);
  if (flags & (QS_JOINT_YUV | QS_UPSAMPLE_UV | QS_LOW_QUALITY))
    return qs_fail(QS_HIP_ENOTSUP, "%s: flags 0x%x are handled by qs_hip_joint_plane / qs_hip_lowq_plane", who, flags);
  int rebalance = !(flags & QS_NO_REBALANCE) && (luma || !(flags & QS_NO_REBALANCE_UV)); // reference :1567-1568
  qs_launch_smooth_plane(static_cast<const QsConsts*>(d_consts), d_coef, d_plane, d_next, rep_top, rep_bot, wblk, hblk,
                         (flags & QS_DIAGONALS) != 0, rebalance, final_clamp, row0 * wblk, row1 * wblk,
                         static_cast<hipStream_t>(stream));
  return launch_status(who);
}

extern "C" int qs_hip_smooth_plane(const void* d_consts, int16_t* d_coef, const uint8_t* d_plane,
                                   int wblk, int hblk, int flags, int luma, int final_clamp, void* stream) {
  return smooth_rows(d_consts, d_coef, d_plane, wblk, hblk, 0, hblk, flags, luma, final_clamp, stream, "qs_hip_smooth_plane");
}

extern "C" int qs_hip_smooth_plane_next(const void* d_consts, int16_t* d_coef, const uint8_t* d_plane, uint8_t* d_plane_next,
                                        int wblk, int hblk, int flags, int luma, int final_clamp, int rep_top, int rep_bot, void* stream) {
  return smooth_rows(d_consts, d_coef, d_plane, wblk, hblk, 0, hblk, flags, luma, final_clamp, stream, "qs_hip_smooth_plane_next",
                     d_plane_next, rep_top, rep_bot);
}

extern "C" int qs_hip_smooth_rows(const void* d_consts, int16_t* d_coef, const uint8_t* d_plane,
                                  int wblk, int hblk, int row0, int row1, int flags, int luma, int final_clamp, void* stream) {
  return smooth_rows(d_consts, d_coef, d_plane, wblk, hblk, row0, row1, flags, luma, final_clamp, stream, "qs_hip_smooth_rows");
}

static int build_plane_set(const qs_hip_plane_ref* refs, uint8_t* const* next, int n, int flags, QsPlaneSet& set, const char* who) {
  if (!refs || n < 1 || n > QS_MAX_PLANES) return qs_fail(QS_HIP_EINVAL, "%s: 1..%d planes per launch", who, QS_MAX_PLANES);
  memset(&set, 0, sizeof set);
  set.n = n;
  int w = 0;
  for (int i = 0; i < n; ++i) {
    const qs_hip_plane_ref& r = refs[i];
    if (int e = check_plane_args(r.d_coef, r.d_plane, r.wblk, r.hblk, who)) return e;
    if (!r.d_consts) return qs_fail(QS_HIP_EINVAL, "%s: null consts", who);
    set.wave0[i] = w;
    w += (r.wblk * r.hblk + 63) / 64;
    QsPlaneRef& R = set.ref[i];
    R.cst = static_cast<const QsConsts*>(r.d_consts);
    R.coef = r.d_coef; R.plane = r.d_plane; R.status = r.d_status;
    R.plane_next = next ? next[i] : nullptr;
    if (R.plane_next && R.plane_next == r.d_plane)
      return qs_fail(QS_HIP_EINVAL, "%s: d_plane_next must be a different buffer (other blocks still read d_plane)", who);
    R.wblk = r.wblk; R.hblk = r.hblk; R.pitch = qs_plane_pitch(r.wblk);
    R.mode = ((!(flags & QS_NO_REBALANCE) && (r.luma || !(flags & QS_NO_REBALANCE_UV))) ? QS_PLANE_REBALANCE : 0) |
             ((r.band & 1) ? 0 : QS_PLANE_REP_TOP) | ((r.band & 2) ? 0 : QS_PLANE_REP_BOT);
  }
  for (int i = n; i < QS_MAX_PLANES + 2; ++i) set.wave0[i] = w;
  return QS_HIP_OK;
}

extern "C" int qs_hip_idct_planes(const qs_hip_plane_ref* refs, int n, int first, void* stream) {
  QsPlaneSet set;
  if (int r = build_plane_set(refs, nullptr, n, 0, set, "qs_hip_idct_planes")) return r;
  if (first)
    for (int i = 0; i < n; ++i)
      if (!refs[i].d_status) return qs_fail(QS_HIP_EINVAL, "qs_hip_idct_planes: first pass needs d_status");
  qs_launch_idct_set(set, first, static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_idct_planes");
}

static int smooth_planes(const qs_hip_plane_ref* refs, uint8_t* const* next, int n, int flags, int final_clamp, void* stream, const char* who) {
  if (flags & (QS_JOINT_YUV | QS_UPSAMPLE_UV | QS_LOW_QUALITY))
    return qs_fail(QS_HIP_ENOTSUP, "%s: flags 0x%x need the cross-component stages", who, flags);
  QsPlaneSet set;
  if (int r = build_plane_set(refs, next, n, flags, set, who)) return r;
  qs_launch_smooth_set(set, (flags & QS_DIAGONALS) != 0, final_clamp, static_cast<hipStream_t>(stream));
  return launch_status(who);
}

extern "C" int qs_hip_smooth_planes(const qs_hip_plane_ref* refs, int n, int flags, int final_clamp, void* stream) {
  return smooth_planes(refs, nullptr, n, flags, final_clamp, stream, "qs_hip_smooth_planes");
}

// the second planes come in a parallel array: qs_hip_plane_ref keeps its size and stride
extern "C" int qs_hip_smooth_planes_next(const qs_hip_plane_ref* refs, uint8_t* const* d_plane_next, int n, int flags,
                                         int final_clamp, void* stream) {
  return smooth_planes(refs, d_plane_next, n, flags, final_clamp, stream, "qs_hip_smooth_planes_next");
}

extern "C" int qs_hip_abi_version(void) { return QS_HIP_ABI_VERSION; }

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
  if (!d_consts || !d_luma_lowres) return qs_fail(QS_HIP_EINVAL, "qs_hip_joint_plane: null consts/luma plane");
  qs_launch_joint(static_cast<const QsConsts*>(d_consts), d_coef, d_plane, d_luma_lowres, wblk, hblk,
                  rebalance, final_clamp, static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_joint_plane");
}

extern "C" int qs_hip_lowq_plane(const void* d_consts, int16_t* d_coef, const uint8_t* d_plane,
                                 int wblk, int hblk, int rebalance, int final_clamp, void* stream) {
  if (int r = check_plane_args(d_coef, d_plane, wblk, hblk, "qs_hip_lowq_plane")) return r;
  if (!d_consts) return qs_fail(QS_HIP_EINVAL, "qs_hip_lowq_plane: null consts");
  const float c1 = 2.0f * sqrtf(0.5f);   // reference quantsmooth.h:926
  qs_launch_lowq(static_cast<const QsConsts*>(d_consts), d_coef, d_plane, wblk, hblk, rebalance, final_clamp, c1,
                 static_cast<hipStream_t>(stream));
  return launch_status("qs_hip_lowq_plane");
}

extern "C" int qs_hip_downsample_plane(const uint8_t* d_luma, int ywblk, int yhblk, uint8_t* d_lowres,
                                       int lwblk, int lhblk, int ws, int hs, void* stream) {
  if (int r = check_plane_args(d_luma, d_lowres, ywblk, yhblk, "qs_hip_downsample_plane")) return r;
  if (lwblk <= 0 || lhblk <= 0 || ws < 1 || hs < 1 || ws > 4 || hs > 4)
    return qs_fail(QS_HIP_EINVAL, "qs_hip_downsample_plane: bad geometry");
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
    return qs_fail(QS_HIP_EINVAL, "qs_hip_upsample_rows: bad argument");
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

