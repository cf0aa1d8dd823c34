// qs_common.h -- what the host-side translation units share: error reporting.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/jpegqs_hip.h"
#include "qs_device.h"
#include "qs_launch.h"

// records the message for qs_hip_last_error() (per thread) and returns `code`
int qs_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
  return qs_fail(e_ == hipErrorOutOfMemory ? QS_HIP_ENOMEM : QS_HIP_ENODEV, \
                 "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
