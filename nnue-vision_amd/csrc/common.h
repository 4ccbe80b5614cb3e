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

// Zero fills as ordinary kernels.  hipMemsetAsync captured into a hipGraph ahead of a kernel that scatters or
// accumulates into the same buffer was observed not to be ordered reliably before that kernel on this stack (ROCm 7.0:
// replayed training steps of the id-list path read stale values), so nothing in a capturable path uses memset nodes.
#ifdef __HIPCC__
namespace {
__global__ __launch_bounds__(256) void nnue_zero_floats_kernel(float* __restrict__ dst, size_t count) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) dst[i] = 0.0f;
}
__global__ __launch_bounds__(256) void nnue_zero_counters_kernel(int* __restrict__ n, float* __restrict__ sink, int B) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B) {
    n[i] = 0;
    sink[i] = 0.0f;
  }
}
}  // namespace
__attribute__((unused)) static inline void nnue_zero_floats(float* dst, size_t count, hipStream_t s) {
  if (count) hipLaunchKernelGGL(nnue_zero_floats_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, dst, count);
}
__attribute__((unused)) static inline void nnue_zero_counters(int* n, float* sink, int B, hipStream_t s) {
  hipLaunchKernelGGL(nnue_zero_counters_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, n, sink, B);
}
#endif

