// Microbenchmark: v_mfma_f32_16x16x4_f32 issue rate per SIMD as a function of the number of independent accumulators a
// wave cycles through (1 wave per SIMD, 256 workgroups of 256 threads).  Build: hipcc -O3 --offload-arch=gfx950 -o
// mfma_chain mfma_chain.hip ; prints cycles per MFMA assuming 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int NACC>
__global__ __launch_bounds__(256) void chain(float* out, int iters) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(float* out, int waves_per_simd) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * waves_per_simd;
  hipLaunchKernelGGL(chain<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(chain<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 8 * NACC * waves_per_simd;
  printf("acc=%d waves/SIMD=%d: %.3f ms, %.1f ns per MFMA per SIMD = %.1f cycles @2.4GHz, %.1f TFLOP/s\n", NACC, waves_per_simd, ms,
         ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, mfma_per_simd * 1024 * 2048 / (ms * 1e-3) / 1e12);
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
  for (int w = 1; w <= 2; ++w) { run<1>(out, w); run<2>(out, w); run<4>(out, w); run<8>(out, w); }
  return 0;
}
