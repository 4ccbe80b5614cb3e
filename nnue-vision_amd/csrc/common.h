// Shared host-side helpers for libnnue_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nnue_hip.h"

// Records a printf-style message retrievable through nnue_hip_last_error().
void nnue_set_error(const char* fmt, ...);

#define NNUE_REQUIRE(cond, code, ...)     \
  do {                                    \
    if (!(cond)) {                        \
      nnue_set_error(__VA_ARGS__);        \
      return (code);                      \
    }                                     \
  } while (0)

static inline int nnue_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    nnue_set_error("%s: %s", what, hipGetErrorString(e));
    return NNUE_E_LAUNCH;
  }
  return NNUE_OK;
}

static inline bool nnue_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t nnue_round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

constexpr int kWave = 64;  // CDNA4 wavefront
