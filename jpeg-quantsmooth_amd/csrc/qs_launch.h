// qs_launch.h -- kernel launchers (defined in qs_kernels.hip), C++ linkage.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "qs_device.h"

void qs_launch_idct_plane(const QsConsts* cst, int16_t* coef, uint8_t* plane, int wblk, int hblk,
                          int first, int rep_top, int rep_bot, int* status, hipStream_t s);
void qs_launch_smooth_plane(const QsConsts* cst, int16_t* coef, const uint8_t* plane, int wblk, int hblk,
                            int diag, int rebalance, int final_clamp, hipStream_t s);
void qs_launch_clamp(int16_t* coef, size_t nblk, hipStream_t s);
void qs_launch_dequant(const QsConsts* cst, int16_t* coef, size_t nblk, hipStream_t s);
