// qs_device.h -- shared host/device definitions for the gfx950 kernels.
//
// Data layout in HBM (per image component):
//   coef   int16 [hblk][wblk][64]   JPEG blocks, natural (row-major) order --
//                                   the packing of libjpeg's JBLOCKROWs.
//   plane  uint8, pitch bytes/row, (hblk*8 + 2) rows.  Pixel (x, y) lives at
//          plane[(y + 1) * pitch + QS_APRON_X + x]; x = -1, x = w, y = -1 and
//          y = h form a clamp-to-edge apron (reference quantsmooth.h:2612-2620
//          replicates borders into its scratch image; we let the IDCT kernel
//          write the apron directly).  QS_APRON_X = 16 keeps every block row
//          16-byte aligned.
//   consts QsConsts, one per component (quant derived values + weight tables).
#pragma once
#include <stdint.h>

#define QS_APRON_X 16
#define QS_LDS_PITCH 65 /* recovery kernels: dwords per coefficient-pair row of a wave's LDS slice, 64 lanes + 1 pad */
#define QS_TAB_MAX 272 /* floats per coefficient with DIAGONALS, 160 without */

// algorithm flags, numerically identical to reference libjpegqs.h:16-23
enum {
  QS_DIAGONALS = 1, QS_JOINT_YUV = 2, QS_UPSAMPLE_UV = 4, QS_LOW_QUALITY = 8,
  QS_NO_REBALANCE = 16, QS_NO_REBALANCE_UV = 32, QS_TRANSCODE = 64
};

// Per-component constants, built on the host (qs_tables.cpp) and read through
// the scalar cache: every lane of a wave works on the same coefficient index,
// so all of this is wave-uniform.
struct QsConsts {
  // indexed by zigzag position k (the order the recovery loop walks, 63..1)
  int32_t nat[64];     // natural index of zigzag position k
  int32_t q[64];       // effective quantiser (0 -> 1), reference :2506-2511
  int32_t qraw[64];    // quantiser as stored in the file (natural index!)
  int32_t x1[64];      // reciprocal (as signed 16-bit), reference :2514-2539
  int32_t x2[64];      // 15 - floor(log2 q): the reference's x2 = -0x8000 >> n enters only as -a * x2 = a << (15 - n)
  float   range[64];   // (float)(2 * q) * 2^-12 (the kernel works on pixels * 2^-12)
  // natural-index copies for the rebalance pass (walks k = 1..63 natural)
  int32_t qn[64], x1n[64], x2n[64];
  int32_t tab_size;    // 160 or 272
  int32_t pad[15];
  float   tab[64 * QS_TAB_MAX]; // weight tables, row k = zigzag position
  // Everything the recovery loop needs about zigzag position k in ONE 16-byte record, fetched
  // with a single s_load_dwordx4 one coefficient ahead of its use:
  //   [0] nat[k] | nat[max(k-1,1)] << 8 | x2[k] << 16    [1] q[k] (low 16, unsigned) | x1[k] << 16
  //   [2] range[k] (float bits)                            [3] what is a function of k alone, see QS_REC_* below
  int32_t rec[64][4];
};
// QsConsts::rec[k][3] (round 5): the per-coefficient scalars of the recovery loop that depend on k only, precomputed --
// i = nat[k], u = i & 7, v = i >> 3, i_nxt = nat[max(k - 1, 1)]
enum {
  QS_REC_H_ANY = 1,        // u != 0: the horizontal section runs
  QS_REC_H_SKIP4 = 2,      // u == 4: its terms with x = 1, 3, 5 have zero weight
  QS_REC_H_EVEN = 4,       // u even: x = 3 has
  QS_REC_V_ANY = 8,        // v != 0: the vertical section runs
  QS_REC_V_SKIP4 = 16,     // v == 4
  QS_REC_V_EVEN = 32,      // v even
  QS_REC_LDS_SHIFT = 6,    // bits 6..19: byte offset of coefficient i in a lane's LDS column, (i >> 1) * 65 * 4 + (i & 1) * 2
  QS_REC_NXT_SHIFT = 12,   // bit 20 (value 0x100 after the shift): the NEXT coefficient has no horizontal section, its
                           // first weight chunk is the border chunk, 256 bytes into its row
  QS_REC_Q1 = 1 << 21      // the coefficient's quantiser is 1 (or a damaged file's 0, treated as 1): its interval is the single
                           // point it already holds (reference quantsmooth.h:1552-1557: d0 = d1 = 0, dl = dh = a0 = coef), so
                           // its whole term stream cannot change anything -- the recovery kernels skip it (exact; the
                           // reference has no such shortcut).  High-quality JPEGs: 8 of 63 luma entries at IJG quality 95,
                           // 28 at 98.
};

static inline int qs_plane_pitch(int wblk) {
  return ((wblk * 8 + QS_APRON_X + 1) + 63) & ~63;
}

// A set of planes processed by ONE launch (qs_*_set_kernel): the components of a
// job, or of many small jobs, share the grid so that small images still fill the
// 256 CUs and a job costs 2 launches per iteration instead of 2 per component.
// Passed by value in the kernarg segment (read through the scalar cache).
#define QS_MAX_PLANES 56
struct QsPlaneRef {
  const QsConsts* cst;   // this plane's constants (device)
  int16_t* coef;         // [hblk][wblk][64]
  uint8_t* plane;        // pixel plane with apron
  int32_t* status;       // range-check flag (pass A, first iteration)
  int32_t wblk, hblk, pitch;
  int32_t mode;          // QS_PLANE_* bits
  // Pass B of iteration n writes the pixel plane of iteration n + 1 itself (the IDCT of the block's final
  // coefficients, which it holds in LDS anyway): a SECOND plane of the same geometry, because the other blocks of
  // the launch still read `plane` (Jacobi iteration).  null: no next plane (last iteration; pass A runs separately).
  uint8_t* plane_next;
};
enum {
  QS_PLANE_REBALANCE = 1,  // pass B: run the rebalance step on this plane
  QS_PLANE_REP_TOP = 2,    // pass A: the y = -1 apron row is a replica of row 0 (image edge) ...
  QS_PLANE_REP_BOT = 4     // ... / the y = h apron row of row h-1; clear = halo row owned by the neighbouring band
};
// one more device pointer per plane of a set, for the stages that read a second plane
// (JOINT_YUV: the low-res luma plane the chroma plane is predicted from)
struct QsPlaneAux {
  const uint8_t* p[QS_MAX_PLANES];
};
struct QsPlaneSet {
  int32_t n, pad;
  // wave0[i] = index of the first 64-block group of plane i in the launch;
  // wave0[n] = total number of groups.  Unused entries repeat wave0[n].
  int32_t wave0[QS_MAX_PLANES + 2];
  QsPlaneRef ref[QS_MAX_PLANES];
};
