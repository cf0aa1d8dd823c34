// qs_launch.h -- kernel launchers (defined in qs_kernels.hip), C++ linkage.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "qs_device.h"

void qs_launch_idct_plane(const QsConsts* cst, int16_t* coef, uint8_t* plane, int wblk, int hblk,
                          int first, int rep_top, int rep_bot, int* status, hipStream_t s);
// plane_next (may be null): the pixel plane of the NEXT iteration, written by pass B itself (fused pass A) -- a second
// plane of the same geometry; rep_top / rep_bot as in qs_launch_idct_plane
void qs_launch_smooth_plane(const QsConsts* cst, int16_t* coef, const uint8_t* plane, uint8_t* plane_next, int rep_top, int rep_bot,
                            int wblk, int hblk, int diag, int rebalance, int final_clamp, int blk_begin, int blk_end, hipStream_t s);
// one launch over a set of whole planes (job / batch layer); QsPlaneRef::plane_next per plane
void qs_launch_idct_set(const QsPlaneSet& set, int first, hipStream_t s);
void qs_launch_smooth_set(const QsPlaneSet& set, int diag, int final_clamp, hipStream_t s);
void qs_launch_clamp(int16_t* coef, size_t nblk, hipStream_t s);
void qs_launch_dequant(const QsConsts* cst, int16_t* coef, size_t nblk, hipStream_t s);

// qs_kernels_aux.hip: cross-component (JOINT_YUV / UPSAMPLE_UV) and LOW_QUALITY stages
void qs_launch_joint(const QsConsts* cst, int16_t* coef, const uint8_t* planeC, const uint8_t* planeL,
                     int wblk, int hblk, int do_rebalance, int final_clamp, hipStream_t s);
// ... over the chroma planes of a set; lowres.p[i] = the low-res luma plane of set.ref[i];
// do_rebalance applies to the planes whose mode carries QS_PLANE_REBALANCE
void qs_launch_joint_set(const QsPlaneSet& set, const QsPlaneAux& lowres, int do_rebalance, int final_clamp, hipStream_t s);
void qs_launch_lowq(const QsConsts* cst, int16_t* coef, const uint8_t* plane, int wblk, int hblk,
                    int do_rebalance, int final_clamp, float c1, hipStream_t s);
void qs_launch_downsample(const uint8_t* Y, int ywblk, int yhblk, uint8_t* L, int lwblk, int lhblk,
                          int ws, int hs, hipStream_t s);
void qs_launch_upsample(const uint8_t* C, const uint8_t* L, int cwblk, const uint8_t* Y, int ywblk,
                        uint8_t* out, int st, int ww, int hh, int w1, int h1, int first_rows, int ws, int hs, hipStream_t s);
void qs_launch_fdct_plane(const uint8_t* px, int st, int16_t* coef, int wblk, int hblk, hipStream_t s);
